set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python tools/bench_variants.py scene-dev ab:dofselect,probe_fastroulette,probe_fastgeom,probe_fastroulette+probe_fastgeom > gpurun_out/r03d_ab_dev.log 2>&1 || { tail gpurun_out/r03d_ab_dev.log; exit 1; }
cat gpurun_out/r03d_ab_dev.log
HINT=6 timeout -k 10 400 python tools/bench_variants.py scene-three-objects ab:dofselect,probe_fastroulette,probe_fastgeom > gpurun_out/r03d_ab_three.log 2>&1 || { tail gpurun_out/r03d_ab_three.log; exit 1; }
cat gpurun_out/r03d_ab_three.log
timeout -k 10 300 python tools/tile_balance.py scene-marchers 16 3840x2160 > gpurun_out/r03d_tile_balance.log 2>&1 || { tail gpurun_out/r03d_tile_balance.log; exit 1; }
cat gpurun_out/r03d_tile_balance.log
