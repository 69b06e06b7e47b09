#!/usr/bin/env python3
"""Static instruction mix of a scene program's kernels (development; CPU only).

  python tools/isa_stats.py scene-dev [1920x1080] [--hint 6] [--opt=-ffp-contract=fast] [--rng philox] [--dump out.s]

Compiles the scene-specialised program with hiprtc for gfx950, disassembles it and prints, per
kernel, registers / spills (code object metadata) and the counts of the instruction classes that
matter for the issue-bound analytic scenes."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


def disassemble(binary):
    start = binary.find(b'\x7fELF')
    with tempfile.NamedTemporaryFile(suffix='.co', delete=False) as f:
        f.write(binary[start:])
        path = f.name
    try:
        return subprocess.check_output([OBJDUMP, '-d', '--mcpu=gfx950', path]).decode()
    finally:
        os.unlink(path)


def classify(mnemonic):
    m = mnemonic
    if m.startswith('v_fma') or m.startswith('v_fmac') or m.startswith('v_mad') or m.startswith('v_pk_fma'):
        return 'v_fma'
    if m.startswith('v_mul_f32') or m.startswith('v_pk_mul_f32'):
        return 'v_mul'
    if m.startswith('v_add_f32') or m.startswith('v_sub_f32') or m.startswith('v_subrev_f32') or m.startswith('v_pk_add_f32'):
        return 'v_add'
    if re.match(r'v_(rcp|sqrt|rsq|exp|log|sin|cos)_', m):
        return 'v_trans'
    if m.startswith('v_div_') or m.startswith('v_ldexp') or m.startswith('v_frexp'):
        return 'v_divfix'
    if m.startswith('v_cmp') or m.startswith('v_cmpx'):
        return 'v_cmp'
    if m.startswith('v_cndmask'):
        return 'v_cndmask'
    if m.startswith('v_mov') or m.startswith('v_accvgpr') or m.startswith('v_readlane') or m.startswith('v_writelane') \
            or m.startswith('v_readfirstlane'):
        return 'v_mov'
    if m.startswith('v_'):
        return 'v_other'
    if m.startswith('s_cbranch') or m.startswith('s_branch'):
        return 's_branch'
    if m.startswith('s_waitcnt') or m.startswith('s_nop'):
        return 's_wait/nop'
    if m.startswith('s_load') or m.startswith('s_buffer_load'):
        return 's_load'
    if m.startswith('s_'):
        return 's_other'
    if m.startswith('ds_'):
        return 'lds'
    if m.startswith('global_') or m.startswith('flat_') or m.startswith('buffer_'):
        return 'vmem'
    if m.startswith('scratch_'):
        return 'scratch'
    return 'other'


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('scene')
    ap.add_argument('size', nargs='?', default='1920x1080')
    ap.add_argument('--hint', type=int, default=None)
    ap.add_argument('--opt', action='append', default=[])
    ap.add_argument('--rng', default='tape')
    ap.add_argument('--dump', default=None)
    ap.add_argument('--kernel', default='clr_render')
    ap.add_argument('--fma', type=int, default=None)
    ap.add_argument('--outline', default=None)
    ap.add_argument('--sorted', type=int, default=None)
    args = ap.parse_args()
    from conftest import load_scene
    from raytracer_amd.renderer import Renderer
    from raytracer_amd import program_cache
    w, h = (int(v) for v in args.size.split('x'))
    kw = {}
    if args.fma is not None:
        kw['fma'] = bool(args.fma)
    if args.sorted is not None:
        kw['sorted_paths'] = bool(args.sorted)
        kw['split_passes'] = False
    if args.outline is not None:
        kw['outline'] = {'0': False, '1': True}.get(args.outline, args.outline)
    r = Renderer(load_scene(args.scene, (w, h)), seed=0, device=False, min_waves_per_simd=args.hint,
                 rng_mode=args.rng, **kw)
    options = list(r.build_options) + args.opt
    binary = program_cache.compile_here(r.program_code, options, 'gfx950')
    meta = program_cache.kernel_metadata(binary)
    text = disassemble(binary)
    if args.dump:
        open(args.dump, 'w').write(text)
    kernels = collections.OrderedDict()
    current = None
    for line in text.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
        if m:
            current = kernels.setdefault(m.group(1), collections.Counter())
            continue
        m = re.match(r'^\s+([a-z_0-9]+)\b', line)
        if m and current is not None:
            current[classify(m.group(1))] += 1
    for name, counts in kernels.items():
        if args.kernel and name != args.kernel and not (args.kernel == 'clr_render' and name not in meta):
            continue
        md = meta.get(name, {})
        print('%s: %d instructions, vgpr %s sgpr %s spills v%s s%s scratch %s lds %s' % (
            name, sum(counts.values()), md.get('.vgpr_count'), md.get('.sgpr_count'), md.get('.vgpr_spill_count'),
            md.get('.sgpr_spill_count'), md.get('.private_segment_fixed_size'), md.get('.group_segment_fixed_size')))
        for k, v in sorted(counts.items(), key=lambda kv: -kv[1]):
            print('    %-12s %6d' % (k, v))


if __name__ == '__main__':
    main()
