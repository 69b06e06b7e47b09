set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -k "signed_zero or accumulated_image or segment_state or speculative or spectral_mode or light or 1080p_bit_exact or per_pixel_random or hip_against_reference_golden or grazing or near_ties" > gpurun_out/r03c_tests.log 2>&1 || { tail -30 gpurun_out/r03c_tests.log; exit 1; }
tail -3 gpurun_out/r03c_tests.log
for sc in scene-dev scene-three-objects scene-dev-spectral; do
  timeout -k 10 200 python bench.py --scene $sc --no-cpu-baseline > gpurun_out/r03c_bench_$sc.json 2> gpurun_out/r03c_bench_$sc.err || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/r03c_bench_$sc.json').read()); print('$sc', round(d['value']), d['ms_per_step'], d['config']['occupancy_hint'], d['roofline'].get('kernel_ms_avg'), d['config']['kernel_registers'])"
  grep occupancy gpurun_out/r03c_bench_$sc.err
done
timeout -k 10 200 python bench.py --rng philox --no-cpu-baseline > gpurun_out/r03c_bench_philox.json 2> gpurun_out/r03c_bench_philox.err || exit 1
python -c "
import json; d=json.loads(open('gpurun_out/r03c_bench_philox.json').read()); print('philox', round(d['value']), d['ms_per_step'], d['config']['occupancy_hint'])"
