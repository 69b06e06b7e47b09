# Bench lines of the BASELINE configurations other than the default one (run through gpurun).
# BENCH_BASELINE=1 adds the CPU baseline and with it the roofline fraction of every line.
set -o pipefail
cd $GRAFT_REPO_ROOT
EXTRA="--no-cpu-baseline"
[ -n "$BENCH_BASELINE" ] && EXTRA="--cpu-seconds 6"
run() { timeout -k 10 400 python bench.py $EXTRA "$@" 2>>gpurun_out/configs.err || exit 1; }
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
