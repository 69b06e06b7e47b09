# Pass-split on/off at several frame sizes (run through gpurun; Gsamples/s and groups per line)
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --occupancy-hint none "$@" 2>>gpurun_out/split.err | grep -o "on scene[a-z-]*\", \"value\": [0-9.]*\|pass_groups\": [0-9]*" | tr '\n' ' '; echo " $@"; }
run --scene scene-test --size 320x240 --passes-per-step 16 --split-passes 0
run --scene scene-test --size 320x240 --passes-per-step 16 --split-passes 1
for size in 320x240 640x480 800x600 1024x768 1920x1080; do
run --scene scene-dev --size $size --passes-per-step 32 --split-passes 0
run --scene scene-dev --size $size --passes-per-step 32 --split-passes 1
done
