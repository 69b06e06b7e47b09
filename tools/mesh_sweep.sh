# development: scene-mesh throughput against mesh size and program form (run through gpurun)
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 240 python bench.py --scene scene-mesh --passes-per-step 16 --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['config'].get('kernel_registers') or {}; print('$*', round(d['value'],1), round(d['ms_per_step'],2), k.get('vgpr_count'), k.get('vgpr_spill_count'), d['config'].get('occupancy_hint'))" || exit 1; }
for grid in ${GRIDS:-4x4 1000x500}; do
  one --mesh-grid $grid "$@"
done
