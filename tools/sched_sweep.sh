#!/bin/bash
# Development: instruction scheduler variants of the render kernel (results do not depend on them).
# Usage (GPU box): SCHED_VARIANTS="a|b" bash tools/sched_sweep.sh [bench args] > gpurun_out/sched.log
BASE="-O3 -ffp-contract=off -fno-slp-vectorize"
IFS='|' read -ra VARIANTS <<< "${SCHED_VARIANTS:-|-mllvm -amdgpu-sched-strategy=max-ilp|-mllvm -misched-prera-direction=bottomup|-mllvm -misched-prera-direction=topdown|-mllvm -amdgpu-sched-strategy=max-memory-clause}"
for extra in "${VARIANTS[@]}"; do
    echo "== $extra"
    timeout -k 10 200 python bench.py --no-cpu-baseline --build-options "$BASE $extra" "$@" 2>gpurun_out/sched.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['config']['occupancy_hint'], d['config']['kernel_registers'].get('vgpr_count'), d['config']['kernel_registers'].get('vgpr_spill_count'), d['image_mean'])" || { tail -3 gpurun_out/sched.err; exit 1; }
done
