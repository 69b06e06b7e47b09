#!/bin/bash
# rocprofv3 evidence for every BASELINE configuration's shipped kernel (run through gpurun):
# one tools/profile.sh pass per configuration, each into its own directory under gpurun_out/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # <dir> <bench args...>
    local out=$1; shift
    PROFILE_OUT=$out PROFILE_ARGS="$*" bash $R/tools/profile.sh > $R/gpurun_out/$out.log 2>&1 || { echo "$out failed"; tail -5 $R/gpurun_out/$out.log; return 1; }
    echo "$out done"
}
for cfg in "$@"; do
    case $cfg in
        dev) run prof_dev --steps 20 --warmup 3 || exit 1;;
        spectral) run prof_spectral --scene scene-dev-spectral --steps 10 --warmup 2 || exit 1;;
        three) run prof_three --scene scene-three-objects --steps 10 --warmup 2 || exit 1;;
        philox) run prof_philox --rng philox --steps 10 --warmup 2 || exit 1;;
        marchers) run prof_marchers --scene scene-marchers --size 3840x2160 --passes-per-step 16 --steps 3 --warmup 1 || exit 1;;
        mesh) run prof_mesh --scene scene-mesh --passes-per-step 16 --steps 3 --warmup 1 || exit 1;;
        test) run prof_test --scene scene-test --size 320x240 --passes-per-step 16 --steps 10 --warmup 2 || exit 1;;
        marcher:*) run prof_${cfg#marcher:} --scene scene-marchers --size 3840x2160 --passes-per-step 16 --steps 2 --warmup 1 --marchers ${cfg#marcher:} || exit 1;;
        *) echo "unknown configuration $cfg"; exit 1;;
    esac
done
