#!/bin/bash
# closed-form ascent + two-level records: parity (all octree users), phases, bench
set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 500 --timeout-method thread -k "octree or mesh or triangles or golden or fixtures or scene_test or variant" > gpurun_out/r03p_tests.log 2>&1 || { tail -30 gpurun_out/r03p_tests.log; exit 1; }
tail -2 gpurun_out/r03p_tests.log
timeout -k 10 200 python tools/mesh_phases.py 1000x500 4 1 > gpurun_out/r03p_phases.log 2>&1 || { tail gpurun_out/r03p_phases.log; exit 1; }
cat gpurun_out/r03p_phases.log
timeout -k 10 300 python bench.py --scene scene-mesh --passes-per-step 16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03p_bench_mesh.json 2> gpurun_out/r03p_bench_mesh.err || { tail -5 gpurun_out/r03p_bench_mesh.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r03p_bench_mesh.json').read()); print('mesh', round(d['value']), d['ms_per_step'], d['config']['occupancy_hint'], d['roofline'].get('kernel_ms_avg'), d['config']['kernel_registers'])"
grep occupancy gpurun_out/r03p_bench_mesh.err
timeout -k 10 120 python bench.py --scene scene-test --size 320x240 --passes-per-step 16 --no-cpu-baseline > gpurun_out/r03p_test.json 2>gpurun_out/r03p_test.err || exit 1
cut -c1-160 gpurun_out/r03p_test.json
