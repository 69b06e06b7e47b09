#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's kernel on the GPU box (run through gpurun):
#   1. kernel trace + stats (per-kernel time),
#   2. SQ counters (VALU utilisation, occupancy) in one pass,
#   3. FETCH_SIZE and WRITE_SIZE (HBM traffic) in passes of their own.
# Results land in gpurun_out/prof/; tools/summarize_profile.py condenses them into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${PROFILE_OUT:-prof}
rm -rf $OUT/trace $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_grbm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline $PROFILE_ARGS"
# bench.py compiles its programs in child processes (program_cache.py); under the profiler no
# child may start, so an un-profiled run first fills an opt-in program store inside the tree and
# the profiled runs load the same code objects from it
export CLRAY_CACHE_DIR=$R/gpurun_out/progcache
timeout -k 10 200 $BENCH > $OUT/plain.log 2>&1 || { echo plain run failed; tail -5 $OUT/plain.log; exit 1; }
# the profiled runs use the occupancy hint that run picked, without the tuning launches
HINT=$(grep -o '"occupancy_hint": [0-9a-z]*' $OUT/plain.log | tail -1 | sed 's/.*: //; s/null/none/')
CHAINS="$BENCH --occupancy-hint $HINT"
# counters are collected per launch with the kernels serialised: one launch per step there
# (--streams 1); the launch chains of the default run get a kernel trace of their own
BENCH="$BENCH --occupancy-hint $HINT --streams 1 --in-flight 1"
timeout -k 10 200 $BENCH > $OUT/plain_hint.log 2>&1 || { echo plain run with hint failed; tail -5 $OUT/plain_hint.log; exit 1; }
echo store filled: $(ls $CLRAY_CACHE_DIR | wc -l) programs, occupancy hint $HINT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
echo trace done
rm -rf $OUT/trace_chains
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_chains -- $CHAINS > $OUT/trace_chains.log 2>&1 || { echo trace of the launch chains failed; tail -5 $OUT/trace_chains.log; exit 1; }
echo trace of the launch chains done
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1 || { echo pmc_sq failed; tail -5 $OUT/pmc_sq.log; exit 1; }
echo sq done
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || { echo pmc_fetch failed; tail -5 $OUT/pmc_fetch.log; exit 1; }
echo fetch done
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || { echo pmc_write failed; tail -5 $OUT/pmc_write.log; exit 1; }
echo write done
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- $BENCH > $OUT/pmc_grbm.log 2>&1 || { echo pmc_grbm failed; tail -5 $OUT/pmc_grbm.log; exit 1; }
echo grbm done
find $OUT -name "*.csv" | head -50
