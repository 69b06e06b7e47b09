set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -k "sorted_paths" > gpurun_out/r03h_tests.log 2>&1 || { tail -40 gpurun_out/r03h_tests.log; exit 1; }
tail -3 gpurun_out/r03h_tests.log
for sc in scene-dev scene-three-objects; do for so in 0 1; do
  timeout -k 10 200 python bench.py --scene $sc --rng philox --no-cpu-baseline --sorted-paths $so > gpurun_out/r03h_bench_${sc}_sorted$so.json 2> gpurun_out/r03h_bench_${sc}_sorted$so.err || { tail -5 gpurun_out/r03h_bench_${sc}_sorted$so.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r03h_bench_${sc}_sorted$so.json').read()); print('$sc sorted $so', round(d['value']), d['ms_per_step'], d['config']['occupancy_hint'], d['roofline'].get('kernel_ms_avg'), d['config']['kernel_registers'])"
  grep occupancy gpurun_out/r03h_bench_${sc}_sorted$so.err
done; done
