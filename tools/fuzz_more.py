import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from test_gpu_fuzz import random_scene, bits
from raytracer_amd.renderer import Renderer
from oracle import oracle
bad = 0
for seed in range(int(os.environ.get("FUZZ_FIRST", 300)), int(os.environ.get("FUZZ_LAST", 348))):
    for mode in ('tape', 'philox'):
        for split in (True, False):
            r = Renderer(random_scene(seed), seed=seed, rng_mode=mode, split_passes=split)
            if split:
                osc = oracle.OracleScene(r)
                tape = r.make_tape(0, 6)
                oimg, _, _ = osc.render(tape)
                ref = oimg
            r.regenerate = bool(seed & 1)
            r.render_tape(r.make_tape(0, 6) if not split else tape)
            ok = np.array_equal(bits(r.img.get()), bits(ref))
            if not ok:
                bad += 1
                print('MISMATCH', seed, mode, split, flush=True)
    print(seed, 'done', flush=True)
print('mismatches', bad)
