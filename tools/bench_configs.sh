set -o pipefail
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>>gpurun_out/configs.err || exit 1; }
{
run --scene scene-three-objects
run --scene scene-test --size 320x240 --passes-per-step 16
run --scene scene-marchers --size 3840x2160 --passes-per-step 16 --steps 5
run --scene scene-mesh --passes-per-step 16 --steps 5
run --scene scene-dev-spectral
run --rng philox
run --rng philox --scene scene-three-objects
} > gpurun_out/bench_configs.jsonl
cut -c1-120 gpurun_out/bench_configs.jsonl
