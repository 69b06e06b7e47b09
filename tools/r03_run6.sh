set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/bench_variants.py scene-dev philox > gpurun_out/r03f_philox_mul.log 2>&1 || { tail gpurun_out/r03f_philox_mul.log; exit 1; }
cat gpurun_out/r03f_philox_mul.log
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/r03f_tests.log 2>&1 || { tail -40 gpurun_out/r03f_tests.log; exit 1; }
tail -3 gpurun_out/r03f_tests.log
