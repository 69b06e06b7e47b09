#!/bin/bash
# stability of the occupancy choice: the same bench three times, hint and rate of each run
set -o pipefail
for sc in scene-three-objects scene-dev; do for k in 1 2 3; do
  timeout -k 10 200 python bench.py --scene $sc --no-cpu-baseline > gpurun_out/r03zp_$sc$k.json 2> gpurun_out/r03zp_$sc$k.err || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/r03zp_$sc$k.json').read()); print('$sc', round(d['value']), round(d['ms_per_step'],3), d['config']['occupancy_hint'], round(d['roofline'].get('kernel_ms_avg'),3), d['config']['kernel_registers']['vgpr_count'])"
  grep occupancy gpurun_out/r03zp_$sc$k.err
done; done
