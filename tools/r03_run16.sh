#!/bin/bash
# ball cull of plane-bounded bodies: parity (goldens, fuzz, scene-test, 1080p subsample), then bench of C2 / C1
set -o pipefail
tag=${1:-zd}
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 500 --timeout-method thread -k "golden or fixtures or fuzz or scene_test or 1080p or speculative or light or variant or segment_state" > gpurun_out/r03${tag}_tests.log 2>&1 || { tail -30 gpurun_out/r03${tag}_tests.log; exit 1; }
tail -2 gpurun_out/r03${tag}_tests.log
for sc in scene-three-objects; do
  timeout -k 10 200 python bench.py --scene $sc --no-cpu-baseline > gpurun_out/r03${tag}_bench_$sc.json 2> gpurun_out/r03${tag}_bench_$sc.err || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/r03${tag}_bench_$sc.json').read()); print('$sc', round(d['value']), d['ms_per_step'], d['config']['occupancy_hint'], d['roofline'].get('kernel_ms_avg'), d['config']['kernel_registers'])"
  grep occupancy gpurun_out/r03${tag}_bench_$sc.err
done
timeout -k 10 120 python bench.py --scene scene-test --size 320x240 --passes-per-step 16 --no-cpu-baseline > gpurun_out/r03${tag}_test.json 2>gpurun_out/r03${tag}_test.err || exit 1
cut -c1-160 gpurun_out/r03${tag}_test.json
