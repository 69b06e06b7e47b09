#!/bin/bash
# two-level node records of the pooled octree traversal: parity at two sizes, then A/B on scene-mesh
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 500 --timeout-method thread -k "flattened_octree or one_million_triangles_pooled" > gpurun_out/r03n_tests.log 2>&1 || { tail -30 gpurun_out/r03n_tests.log; exit 1; }
tail -2 gpurun_out/r03n_tests.log
for two in 1 0; do
  timeout -k 10 300 python bench.py --scene scene-mesh --passes-per-step 16 --steps 6 --warmup 2 --no-cpu-baseline --mesh-two-level $two > gpurun_out/r03n_bench_mesh_two$two.json 2> gpurun_out/r03n_bench_mesh_two$two.err || { tail -5 gpurun_out/r03n_bench_mesh_two$two.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r03n_bench_mesh_two$two.json').read()); print('two-level $two', round(d['value']), d['ms_per_step'], d['config']['occupancy_hint'], d['roofline'].get('kernel_ms_avg'), d['config']['kernel_registers'])"
done
