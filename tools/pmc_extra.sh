#!/bin/bash
# Development: where a wave's cycles go (wait / active by instruction kind) and the instruction mix
# of clr_render, two counter passes of their own (run through gpurun).
#   PROFILE_OUT=dir PROFILE_ARGS="bench args" bash tools/pmc_extra.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${PROFILE_OUT:-pmc_extra}
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CLRAY_CACHE_DIR=$R/gpurun_out/progcache
BENCH="python3 $R/bench.py --no-cpu-baseline --streams 1 --in-flight 1 $PROFILE_ARGS"
timeout -k 10 200 $BENCH > $OUT/plain.log 2>&1 || { echo plain run failed; tail -5 $OUT/plain.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/cycles -- $BENCH > $OUT/cycles.log 2>&1 || { echo cycles pass failed; tail -5 $OUT/cycles.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/mix -- $BENCH > $OUT/mix.log 2>&1 || { echo mix pass failed; tail -5 $OUT/mix.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
for d in ('cycles', 'mix'):
    f = max(glob.glob(sys.argv[1] + '/' + d + '/*/*_counter_collection.csv'))
    tot = collections.Counter(); n = collections.Counter()
    for row in csv.DictReader(open(f)):
        if row['Kernel_Name'].startswith('clr_render') and 'debug' not in row['Kernel_Name']:
            tot[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
    for k in sorted(tot):
        print('%-26s %.4g per launch (%d launches)' % (k, tot[k] / n[k], n[k]))
PY
