#!/bin/bash
# Where do the waves of clr_render spend their cycles?  Two SQ counter passes over bench.py
# (run through gpurun after tools/profile.sh conventions: programs come from an in-tree store).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/stalls
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline $PROFILE_ARGS"
export CLRAY_CACHE_DIR=$R/gpurun_out/progcache
timeout -k 10 200 $BENCH > $OUT/plain.log 2>&1 || { echo plain run failed; tail -5 $OUT/plain.log; exit 1; }
HINT=$(grep -o '"occupancy_hint": [0-9a-z]*' $OUT/plain.log | tail -1 | sed 's/.*: //; s/null/none/')
BENCH="$BENCH --occupancy-hint $HINT"
timeout -k 10 200 $BENCH > $OUT/plain_hint.log 2>&1 || { echo plain run with hint failed; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/a -- $BENCH > $OUT/a.log 2>&1 || { echo pass a failed; tail -5 $OUT/a.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH_LEVEL SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES --output-format csv -d $OUT/b -- $BENCH > $OUT/b.log 2>&1 || { echo pass b failed; tail -5 $OUT/b.log; exit 1; }
python3 - <<PY
import csv, glob, collections, os
for d in ('a', 'b'):
    f = max(glob.glob('$OUT/%s/*/*_counter_collection.csv' % d), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'] == 'clr_render':
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print('%-28s %.4g' % (k, sum(v) / len(v)))
PY
