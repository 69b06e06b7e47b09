#!/usr/bin/env python3
"""Condense gpurun_out/prof/* (tools/profile.sh) into profiles/<tag>_* files.

Writes the kernel stats CSV as is, a per-launch counter table for clr_render, and
profiles/pmc_traffic.json with the HBM bytes per launch that bench.py reports as
roofline.traffic.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte
requests as 64 bytes for wide coalesced reads, so it is doubled
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', 'prof')
DST = os.path.join(ROOT, 'profiles')


def newest(pattern):
    """gpurun merges every call's output into the same directories: take the latest run's file."""
    return max(glob.glob(pattern), key=os.path.getmtime)


def busy_span(intervals):
    """Total time covered by at least one of the (start, end) intervals."""
    total, cur_start, cur_end = 0, None, None
    for b, e in intervals:
        if cur_end is None or b > cur_end:
            if cur_end is not None:
                total += cur_end - cur_start
            cur_start, cur_end = b, e
        else:
            cur_end = max(cur_end, e)
    return total + (cur_end - cur_start if cur_end is not None else 0)


def main(tag, key, src=None):
    global SRC
    if src:
        SRC = os.path.join(ROOT, 'gpurun_out', src)
    os.makedirs(DST, exist_ok=True)
    stats = newest(os.path.join(SRC, 'trace', '*', '*_kernel_stats.csv'))
    shutil.copy(stats, os.path.join(DST, tag + '_kernel_stats.csv'))
    trace = newest(os.path.join(SRC, 'trace', '*', '*_kernel_trace.csv'))
    rows = [r for r in csv.DictReader(open(trace)) if r['Kernel_Name'] == 'clr_render']
    durations = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
    counters = collections.OrderedDict()
    for d in ('pmc_sq', 'pmc_fetch', 'pmc_write', 'pmc_grbm'):
        for f in [newest(os.path.join(SRC, d, '*', '*_counter_collection.csv'))]:
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if r['Kernel_Name'] == 'clr_render':
                    agg[r['Counter_Name']].append(float(r['Counter_Value']))
            for k, v in sorted(agg.items()):
                counters[k] = sum(v) / len(v)
    fetch = counters.get('FETCH_SIZE', 0.0) * 1024 * 2      # gfx950 correction, see docstring
    write = counters.get('WRITE_SIZE', 0.0) * 1024
    summary = {
        'kernel': 'clr_render', 'launches': len(rows),
        'avg_ms': sum(durations) / len(durations), 'min_ms': min(durations), 'max_ms': max(durations),
        # rocprofv3's VGPR_Count column is about half the code object's .vgpr_count (which bench.py
        # reports as config.kernel_registers): e.g. 36-40 here for an 72-80 register kernel
        'rocprof_vgpr_column': int(rows[0]['VGPR_Count']), 'rocprof_sgpr_column': int(rows[0]['SGPR_Count']),
        'grid': int(rows[0]['Grid_Size_X']), 'workgroup': int(rows[0]['Workgroup_Size_X']),
        'counters_per_launch': counters,
        'hbm_read_bytes_per_launch': fetch, 'hbm_write_bytes_per_launch': write,
        'hbm_bytes_per_launch': fetch + write,
    }
    # the default run's launch chains (bench.py --streams 2): launches of slices of the frame that
    # overlap; span = first start to last end of all clr_render launches of that run
    chains = glob.glob(os.path.join(SRC, 'trace_chains', '*', '*_kernel_trace.csv'))
    if chains:
        shutil.copy(newest(os.path.join(SRC, 'trace_chains', '*', '*_kernel_stats.csv')),
                    os.path.join(DST, tag + '_chains_kernel_stats.csv'))
        crow = [r for r in csv.DictReader(open(max(chains, key=os.path.getmtime))) if r['Kernel_Name'] == 'clr_render']
        # the launches of the chains are those on slices of the frame: the most frequent grid size
        # (bench.py's roofline leg adds a few whole-frame launches behind the timed region)
        grids = collections.Counter(int(r['Grid_Size_X']) for r in crow)
        chain_grid = max(grids, key=lambda g: (grids[g], -g))
        crow = [r for r in crow if int(r['Grid_Size_X']) == chain_grid]
        starts = [int(r['Start_Timestamp']) for r in crow]
        ends = [int(r['End_Timestamp']) for r in crow]
        per_step = max(1, round(summary['grid'] / chain_grid))
        steps = len(crow) / per_step
        # the longest gap-free stretch is the timed loop; warm-up launches are waited for one by one
        summary['launch_chains'] = {
            'launches': len(crow), 'launches_per_step': per_step,
            'avg_launch_ms': sum(e - b for b, e in zip(starts, ends)) / len(crow) / 1e6,
            'busy_ms_per_step': busy_span(sorted(zip(starts, ends))) / steps / 1e6,
        }
    c = counters
    if 'SQ_INSTS_VALU' in c and 'SQ_WAVES' in c:
        summary['valu_insts_per_wave'] = c['SQ_INSTS_VALU'] / c['SQ_WAVES']
        summary['salu_insts_per_wave'] = c['SQ_INSTS_SALU'] / c['SQ_WAVES']
        # one wave64 VALU instruction occupies a SIMD-32 for 2 cycles; 256 CUs x 4 SIMDs
        if 'GRBM_GUI_ACTIVE' in c:
            cycles = c['GRBM_GUI_ACTIVE'] / 8.0          # summed over the 8 XCDs
            summary['shader_clock_ghz'] = cycles / (summary['avg_ms'] * 1e6)
            summary['valu_issue_utilisation'] = c['SQ_INSTS_VALU'] * 2.0 / (cycles * 1024)
        if 'SQ_THREAD_CYCLES_VALU' in c and 'SQ_ACTIVE_INST_VALU' in c:
            summary['valu_lane_utilisation'] = c['SQ_THREAD_CYCLES_VALU'] / (c['SQ_ACTIVE_INST_VALU'] * 64.0) \
                if c['SQ_ACTIVE_INST_VALU'] else None
    json.dump(summary, open(os.path.join(DST, tag + '_clr_render.json'), 'w'), indent=1)
    path = os.path.join(DST, 'pmc_traffic.json')
    table = json.load(open(path)) if os.path.exists(path) else {}
    entry = {'bytes_per_launch': fetch + write, 'source': 'profiles/%s_clr_render.json' % tag,
             'hbm_read_bytes_per_launch': fetch, 'hbm_write_bytes_per_launch': write, 'avg_ms': summary['avg_ms']}
    # key = '<scene> <W>x<H> p<passes> <rng>': samples of one launch
    try:
        size, passes = key.split()[1], key.split()[2]
        launch_samples = int(size.split('x')[0]) * int(size.split('x')[1]) * int(passes.lstrip('p'))
    except (IndexError, ValueError):
        launch_samples = None
    for name in ('valu_issue_utilisation', 'valu_lane_utilisation'):
        if summary.get(name) is not None:
            entry[name] = summary[name]
    if launch_samples and 'SQ_INSTS_VALU' in c:
        # wave-level instructions per sample of the wave's 64 lanes
        entry['valu_insts_per_sample'] = c['SQ_INSTS_VALU'] / (launch_samples / 64.0)
        entry['salu_insts_per_sample'] = c['SQ_INSTS_SALU'] / (launch_samples / 64.0)
    table[key] = entry
    json.dump(table, open(path, 'w'), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == '__main__':
    if len(sys.argv) < 2 or sys.argv[1].startswith('-'):
        raise SystemExit('usage: summarize_profile.py <tag> [<workload key of profiles/pmc_traffic.json>] [<directory under gpurun_out>]')
    main(sys.argv[1] if len(sys.argv) > 1 else 'r01',
         sys.argv[2] if len(sys.argv) > 2 else 'scene-dev 1920x1080 p32 tape',
         sys.argv[3] if len(sys.argv) > 3 else None)
