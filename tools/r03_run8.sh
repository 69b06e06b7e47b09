set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export CLRAY_CACHE_DIR=$R/gpurun_out/progcache
for so in 0 1; do
  B="python3 $R/bench.py --rng philox --no-cpu-baseline --sorted-paths $so --occupancy-hint 6 --streams 1 --in-flight 1 --steps 4 --warmup 1"
  timeout -k 10 200 $B > $R/gpurun_out/r03i_plain$so.log 2>&1 || { tail -5 $R/gpurun_out/r03i_plain$so.log; exit 1; }
  rm -rf $R/gpurun_out/r03i_pmc$so
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/r03i_pmc$so -- $B > $R/gpurun_out/r03i_pmc$so.log 2>&1 || { tail -5 $R/gpurun_out/r03i_pmc$so.log; exit 1; }
  python3 - $R/gpurun_out/r03i_pmc$so $so <<'PY'
import csv, glob, sys, collections
f = max(glob.glob(sys.argv[1] + '/*/*_counter_collection.csv'))
tot = collections.Counter(); n = collections.Counter()
for row in csv.DictReader(open(f)):
    if row['Kernel_Name'] == 'clr_render':
        tot[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
c = {k: tot[k] / n[k] for k in tot}
print('sorted', sys.argv[2], {k: '%.4g' % v for k, v in c.items()})
print('  lanes %.3f, VALU insts per sample %.0f, LDS insts per sample %.0f' % (
    c['SQ_THREAD_CYCLES_VALU'] / (c['SQ_ACTIVE_INST_VALU'] * 64), c['SQ_INSTS_VALU'] / (1920 * 1080 * 32 / 64), c['SQ_INSTS_LDS'] / (1920 * 1080 * 32 / 64)))
PY
done
