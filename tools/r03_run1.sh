set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python bench.py > gpurun_out/r03a_bench_dev.json 2> gpurun_out/r03a_bench_dev.err || exit 1
timeout -k 10 300 python tools/bench_variants.py scene-dev fma > gpurun_out/r03a_fma_ceiling.log 2>&1 || exit 1
for o in 0 marchers meshes 1; do
  timeout -k 10 200 python bench.py --scene scene-test --size 320x240 --passes-per-step 16 --no-cpu-baseline --outline $o > gpurun_out/r03a_test_outline_$o.json 2> gpurun_out/r03a_test_outline_$o.err || exit 1
done
