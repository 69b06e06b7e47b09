cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --occupancy-hint none "$@" 2>>gpurun_out/split.err | grep -o "on scene[a-z-]*\", \"value\": [0-9.]*\|pass_groups\": [0-9]*" | tr '\n' ' '; echo " $@"; }
for lanes in 393216 1572864 6291456; do
export CLR_SPLIT_LANES=$lanes; echo lanes $lanes
run --scene scene-test --size 320x240 --passes-per-step 16
run --scene scene-dev --size 320x240 --passes-per-step 32
run --scene scene-dev --size 640x480 --passes-per-step 32
done
unset CLR_SPLIT_LANES
run --scene scene-dev --size 320x240 --passes-per-step 32 --split-passes 0
run --scene scene-dev --size 640x480 --passes-per-step 32 --split-passes 0
run --scene scene-dev --size 800x600 --passes-per-step 32 --split-passes 0
run --scene scene-dev --size 800x600 --passes-per-step 32 --split-passes 1
run --scene scene-dev --size 1920x1080 --passes-per-step 32 --split-passes 1
