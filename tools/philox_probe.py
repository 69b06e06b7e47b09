import sys, time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from conftest import load_scene
from raytracer_amd.renderer import Renderer
for light in (True, False):
    for mode in ('tape','philox'):
        s = load_scene('scene-dev',(1920,1080)); s.samples_per_pixel=4096
        s.get_object('light').bidirectional_light = light
        r = Renderer(s, seed=0, rng_mode=mode)
        r.tune_occupancy()
        tape = r.make_tape(0, 64); r.upload_tape(tape)
        r.render_resident(0,32)
        ms = [r.render_resident(32,32) for _ in range(5)]
        print('NEE light', light, mode, 'ms/32 passes %.3f' % np.median(ms), 'hint', r.min_waves_per_simd, flush=True)
