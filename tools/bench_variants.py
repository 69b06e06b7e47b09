"""A/B timing of build-option variants of the render kernel in ONE process, interleaved
rounds (cdna_hip_programming.md §5.4 rule 24).  Development tool."""
import runpy, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd.renderer import Renderer

VARIANTS = {
    'parity': ['-O3', '-ffp-contract=off'],
    'contract': ['-O3', '-ffp-contract=fast'],
    'fastdiv': ['-O3', '-ffp-contract=fast', '-fno-hip-fp32-correctly-rounded-divide-sqrt'],
    'fastmath': ['-O3', '-ffast-math', '-fno-hip-fp32-correctly-rounded-divide-sqrt'],
}

BASE = ['-O3', '-ffp-contract=off']
SETS = {
    'options': VARIANTS,
    # exact division / square root: generic expansions vs operand-range speculation, and the
    # speculation with a threshold that makes a large share of the segments replay
    'exact': {
        'parity': (BASE, {'speculate': False}),
        'speculative': (BASE, {'speculate': True}),
        'replay-heavy': (BASE + ['-DCLR_GUARD_LO=0x1p-7f'], {'speculate': True}),
    },
    # code generation only (all bit-identical): SLP packing, scheduler strategy, machine LICM
    'codegen': {
        'parity': (BASE, {'min_waves_per_simd': 5}),
        'noslp5': (BASE + ['-fno-slp-vectorize'], {'min_waves_per_simd': 5}),
        'noslp4': (BASE + ['-fno-slp-vectorize'], {'min_waves_per_simd': 4}),
        'noslp6': (BASE + ['-fno-slp-vectorize'], {'min_waves_per_simd': 6}),
        'maxilp5': (BASE + ['-mllvm', '-amdgpu-sched-strategy=max-ilp'], {'min_waves_per_simd': 5}),
        'nomlicm5': (BASE + ['-mllvm', '-disable-machine-licm'], {'min_waves_per_simd': 5}),
        'noslp+ilp5': (BASE + ['-fno-slp-vectorize', '-mllvm', '-amdgpu-sched-strategy=max-ilp'], {'min_waves_per_simd': 5}),
        'noslp+nomlicm5': (BASE + ['-fno-slp-vectorize', '-mllvm', '-disable-machine-licm'], {'min_waves_per_simd': 5}),
    },
    'codegen2': {
        'parity': (BASE, {}),
        'noslp': (BASE + ['-fno-slp-vectorize'], {}),
        'noslp+nomlicm': (BASE + ['-fno-slp-vectorize', '-mllvm', '-disable-machine-licm'], {}),
        'noslp+scalarq': (BASE + ['-fno-slp-vectorize', '-DCLR_SCALAR_QUOTIENTS'], {}),
        'noslp+scalarq5': (BASE + ['-fno-slp-vectorize', '-DCLR_SCALAR_QUOTIENTS'], {'min_waves_per_simd': 5}),
        'noslp+scalarq6': (BASE + ['-fno-slp-vectorize', '-DCLR_SCALAR_QUOTIENTS'], {'min_waves_per_simd': 6}),
    },
    # round 3: the ceiling of what FMA contraction can give the shipped kernel (every a*b+c the
    # compiler can find, not a selective mode), at the tuned hint and its neighbours
    'fma': {
        'parity6': (BASE + ['-fno-slp-vectorize'], {'min_waves_per_simd': 6}),
        'parity8': (BASE + ['-fno-slp-vectorize'], {'min_waves_per_simd': 8}),
        'contract6': (['-O3', '-ffp-contract=fast', '-fno-slp-vectorize'], {'min_waves_per_simd': 6}),
        'contract7': (['-O3', '-ffp-contract=fast', '-fno-slp-vectorize'], {'min_waves_per_simd': 7}),
        'contract8': (['-O3', '-ffp-contract=fast', '-fno-slp-vectorize'], {'min_waves_per_simd': 8}),
    },
    'philox': {
        'mad64': (BASE + ['-fno-slp-vectorize'], {'rng_mode': 'philox'}),
        'split': (BASE + ['-fno-slp-vectorize', '-DCLR_PHILOX_SPLIT_MUL'], {'rng_mode': 'philox'}),
    },
    # what the generator costs: fewer rounds (NOT the published generator: measurement only)
    'philox-rounds': {
        'rounds10': (BASE + ['-fno-slp-vectorize'], {'rng_mode': 'philox'}),
        'rounds7': (BASE + ['-fno-slp-vectorize', '-DCLR_PHILOX_ROUNDS=7'], {'rng_mode': 'philox'}),
        'rounds1': (BASE + ['-fno-slp-vectorize', '-DCLR_PHILOX_ROUNDS=1'], {'rng_mode': 'philox'}),
    },
    'codegen3': {
        'parity': (BASE + ['-fno-slp-vectorize'], {}),
        'nopostsched': (BASE + ['-fno-slp-vectorize', '-mllvm', '-enable-post-misched=0'], {}),
        'nomisched': (BASE + ['-fno-slp-vectorize', '-mllvm', '-enable-misched=0'], {}),
        'minreg': (BASE + ['-fno-slp-vectorize', '-mllvm', '-amdgpu-sched-strategy=iterative-minreg'], {}),
        'iter-ilp': (BASE + ['-fno-slp-vectorize', '-mllvm', '-amdgpu-sched-strategy=iterative-ilp'], {}),
        'max-ilp': (BASE + ['-fno-slp-vectorize', '-mllvm', '-amdgpu-sched-strategy=max-ilp'], {}),
    },
}


def main(scene_name='scene-dev', size=(1920, 1080), passes=32, rounds=9):
    passes = int(os.environ.get('PASSES', passes))
    if os.environ.get('SIZE'):
        size = tuple(int(v) for v in os.environ['SIZE'].split('x'))
    rs = {}
    tape = None
    for name, opts in VARIANTS.items():
        sys.path.insert(0, 'tests')
        from conftest import load_scene
        scene = load_scene(scene_name, size)
        kwargs = {}
        if isinstance(opts, tuple):
            opts, kwargs = opts
        kwargs.setdefault('split_passes', False)
        if os.environ.get('HINT'):
            kwargs = dict(kwargs, min_waves_per_simd=int(os.environ['HINT']))
        try:
            r = Renderer(scene, seed=0, build_options=opts, **kwargs)
        except Exception as err:
            print(name, 'does not build:', str(err)[:150])
            continue
        if tape is None:
            tape = r.make_tape(0, passes)
        if os.environ.get('TUNE'):
            r.tune_occupancy(passes=8)
        r.upload_tape(tape)
        r.render_resident(0, passes)
        rs[name] = r
    ref = None
    times = {k: [] for k in rs}
    ids0 = t0 = None
    for _ in range(rounds):
        for name, r in rs.items():
            times[name].append(r.render_resident(0, passes))
    for name, r in rs.items():
        r.img.fill(0)
        r.render_resident(0, passes)
        img = r.img.get()[:, :3].astype(np.float64)
        if ref is None:
            ref = img
        ids, t, _ = r.first_hit(tape)
        if ids0 is None:
            ids0, t0 = ids, t
        rmse = np.sqrt(np.mean((img - ref) ** 2)) / np.sqrt(np.mean(ref ** 2))
        ms = np.median(times[name])
        print('%-9s median %.3f ms min %.3f  %.0f Msamples/s  rel-RMSE vs parity %.2e  first-hit id flips %d  max t rel err %.2e'
              % (name, ms, min(times[name]), r.n_pixels * passes / ms / 1e3, rmse, int((ids != ids0).any(axis=1).sum()),
                 float(np.max(np.abs(t - t0) / np.abs(t0)))))

if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[2].startswith('ab:'):
        # ab:name1,name2+name3  -> the shipped program against the same with those code-generation
        # variants switched on (Renderer(variants=...)); every result is checked against the first
        OPTS = BASE + ['-fno-slp-vectorize']
        VARIANTS = {'shipped': (OPTS, {})}
        for name in sys.argv[2][3:].split(','):
            VARIANTS[name] = (OPTS, {'variants': tuple(name.split('+'))})
        if os.environ.get('SPLIT'):
            for k in VARIANTS:
                VARIANTS[k][1]['split_passes'] = bool(int(os.environ['SPLIT']))
    elif len(sys.argv) > 2:
        VARIANTS = SETS[sys.argv[2]]
    main(*(sys.argv[1:2] or ['scene-dev']))
