#!/bin/bash
# pooled octree traversal variants: parity of the flattened layouts, phases, bench
set -o pipefail
tag=${1:-q}
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 500 --timeout-method thread -k "flattened_octree or one_million" > gpurun_out/r03${tag}_tests.log 2>&1 || { tail -30 gpurun_out/r03${tag}_tests.log; exit 1; }
tail -2 gpurun_out/r03${tag}_tests.log
timeout -k 10 200 python tools/mesh_phases.py 1000x500 4 > gpurun_out/r03${tag}_phases.log 2>&1 || { tail gpurun_out/r03${tag}_phases.log; exit 1; }
cat gpurun_out/r03${tag}_phases.log
timeout -k 10 300 python bench.py --scene scene-mesh --passes-per-step 16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03${tag}_bench_mesh.json 2> gpurun_out/r03${tag}_bench_mesh.err || { tail -5 gpurun_out/r03${tag}_bench_mesh.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r03${tag}_bench_mesh.json').read()); print('mesh', round(d['value']), d['ms_per_step'], d['config']['occupancy_hint'], d['roofline'].get('kernel_ms_avg'), d['config']['kernel_registers'])"
grep occupancy gpurun_out/r03${tag}_bench_mesh.err
