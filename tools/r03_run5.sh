set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 400 --timeout-method thread -k "octree or triangle or mesh or million" > gpurun_out/r03e_tests.log 2>&1 || { tail -30 gpurun_out/r03e_tests.log; exit 1; }
tail -3 gpurun_out/r03e_tests.log
for ropes in 1 0; do
  timeout -k 10 300 python bench.py --scene scene-mesh --passes-per-step 16 --steps 6 --warmup 2 --no-cpu-baseline --mesh-ropes $ropes > gpurun_out/r03e_bench_mesh_ropes$ropes.json 2> gpurun_out/r03e_bench_mesh_ropes$ropes.err || { tail -5 gpurun_out/r03e_bench_mesh_ropes$ropes.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r03e_bench_mesh_ropes$ropes.json').read()); print('ropes $ropes', round(d['value']), d['ms_per_step'], d['config']['occupancy_hint'], d['roofline'].get('kernel_ms_avg'), d['config']['kernel_registers'])"
  grep occupancy gpurun_out/r03e_bench_mesh_ropes$ropes.err
done
