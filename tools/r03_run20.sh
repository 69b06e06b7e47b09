#!/bin/bash
# XCD-aware block mapping in chunks of C blocks (development knob CLR_XCD_CHUNK): scene-mesh bench per C
set -o pipefail
for c in "$@"; do
  CLR_XCD_CHUNK=$c timeout -k 10 300 python bench.py --scene scene-mesh --passes-per-step 16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03zzg_mesh_c$c.json 2> gpurun_out/r03zzg_mesh_c$c.err || { tail -5 gpurun_out/r03zzg_mesh_c$c.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r03zzg_mesh_c$c.json').read()); print('chunk $c:', round(d['value']), round(d['ms_per_step'],3), 'lone', round(d['roofline']['kernel_ms_avg'],3), d['config']['steps_in_flight'])"
done
