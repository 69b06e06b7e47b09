#!/bin/bash
# LDS stash of per-pass / per-launch lane values (one-lane programs): parity, then bench at hints 7 / 8 / auto
set -o pipefail
export CLR_VARIANTS=lds-stash
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 400 --timeout-method thread -k "headline or speculative or segment_state or accumulated or signed_zero or 1080p" > gpurun_out/r03zs_tests.log 2>&1 || { tail -20 gpurun_out/r03zs_tests.log; exit 1; }
tail -2 gpurun_out/r03zs_tests.log
for cfg in "lds-stash 8" "lds-stash 7" "none 8" "none 7" "lds-stash auto" "none auto"; do
  set -- $cfg
  if [ "$1" = none ]; then unset CLR_VARIANTS; else export CLR_VARIANTS=$1; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --occupancy-hint $2 > gpurun_out/r03zs_dev_$1_$2.json 2> gpurun_out/r03zs_dev_$1_$2.err || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/r03zs_dev_$1_$2.json').read()); print('$1 hint $2:', round(d['value']), round(d['ms_per_step'],4), 'lone', round(d['roofline']['kernel_ms_avg'],4), d['config']['occupancy_hint'], d['config']['kernel_registers'])"
done
