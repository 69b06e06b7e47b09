set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -k "accumulated_image or segment_state or speculative or spectral_mode or light or 1080p_bit_exact or interleaved_band or per_pixel_random or hip_against_reference_golden" > gpurun_out/r03b_tests.log 2>&1 || { tail -30 gpurun_out/r03b_tests.log; exit 1; }
tail -3 gpurun_out/r03b_tests.log
for sc in scene-dev scene-three-objects scene-dev-spectral; do
  timeout -k 10 200 python bench.py --scene $sc --no-cpu-baseline > gpurun_out/r03b_bench_$sc.json 2> gpurun_out/r03b_bench_$sc.err || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/r03b_bench_$sc.json').read()); print('$sc', round(d['value']), d['ms_per_step'], d['config']['occupancy_hint'], d['roofline'].get('kernel_ms_avg'))"
done
