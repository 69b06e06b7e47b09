#!/bin/bash
# steps in flight chosen by the tuner: stability on scene-test (three runs) and scene-mesh, pass-split parity tests
set -o pipefail
for k in 1 2 3; do
  timeout -k 10 200 python bench.py --scene scene-test --size 320x240 --passes-per-step 16 --no-cpu-baseline > gpurun_out/r03zy_test$k.json 2> gpurun_out/r03zy_test$k.err || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/r03zy_test$k.json').read()); print('scene-test', round(d['value']), round(d['ms_per_step'],3), d['config']['occupancy_hint'], d['config']['steps_in_flight'], round(d['roofline'].get('kernel_ms_avg'),3))"
  grep -o "steps in flight.*" gpurun_out/r03zy_test$k.err
done
timeout -k 10 300 python bench.py --scene scene-mesh --passes-per-step 16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03zy_mesh.json 2> gpurun_out/r03zy_mesh.err || exit 1
python -c "
import json; d=json.loads(open('gpurun_out/r03zy_mesh.json').read()); print('scene-mesh', round(d['value']), round(d['ms_per_step'],3), d['config']['occupancy_hint'], d['config']['steps_in_flight'])"
grep -o "steps in flight.*" gpurun_out/r03zy_mesh.err
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "pass_split or in_flight or occupancy or scene_test or static_log or no_alloc or two_workers" > gpurun_out/r03zy_tests.log 2>&1 || { tail -20 gpurun_out/r03zy_tests.log; exit 1; }
tail -2 gpurun_out/r03zy_tests.log
