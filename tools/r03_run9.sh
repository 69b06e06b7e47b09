set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err || { tail -5 gpurun_out/r03_bench_default.err; exit 1; }
cut -c1-200 gpurun_out/r03_bench_default.json
rm -f gpurun_out/configs.err
BENCH_BASELINE=1 bash tools/bench_configs.sh || { tail -5 gpurun_out/configs.err; exit 1; }
