"""Development tool: pooled vs per-lane triangle tests on the 1 M-triangle scene."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd.scenes.library import scene_mesh
from raytracer_amd.renderer import Renderer
import numpy as np
ref = None
for pooled in (False, True):
    scene = scene_mesh(1000, 500, pooled=pooled)
    r = Renderer(scene, seed=0)
    best = r.tune_occupancy(passes=4)
    tape = r.make_tape(0, 4); r.upload_tape(tape); r.render_resident(0, 4)
    ms = min(r.render_resident(0, 4) for _ in range(3))
    r.img.fill(0); r.render_resident(0, 4); img = r.img.get()
    if ref is None: ref = img
    print('pooled', pooled, 'hint', best, {k: round(v, 2) for k, v in r.occupancy_timings.items()}, '%.2f ms' % ms,
          '%.0f Msamples/s' % (r.n_pixels * 4 / ms / 1e3), 'identical', np.array_equal(img, ref), flush=True)
