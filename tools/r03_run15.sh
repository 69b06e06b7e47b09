#!/bin/bash
# scene-mesh bench under code-generation variants (CLR_VARIANTS), one line each
set -o pipefail
tag=$1; shift
for v in "$@"; do
  name=${v:-base}
  CLR_VARIANTS=$v timeout -k 10 300 python bench.py --scene scene-mesh --passes-per-step 16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03${tag}_bench_$name.json 2> gpurun_out/r03${tag}_bench_$name.err || { tail -5 gpurun_out/r03${tag}_bench_$name.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r03${tag}_bench_$name.json').read()); print('$name', round(d['value']), round(d['ms_per_step'],3), d['config']['occupancy_hint'], round(d['roofline'].get('kernel_ms_avg'),3), d['config']['kernel_registers']['vgpr_spill_count'])"
  grep occupancy gpurun_out/r03${tag}_bench_$name.err
done
