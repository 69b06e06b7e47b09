"""A/B of __launch_bounds__ occupancy hints for clr_render in one process (development tool)."""
import runpy, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd.renderer import Renderer

def main(scene_name='scene-dev', size=(1920, 1080), passes=32, rounds=5):
    rs, tape = {}, None
    for w in [(spec, hint) for spec in (False, True) for hint in (None, 2, 3, 4, 5)]:
        scene = runpy.run_path('raytracer_amd/scenes/%s.py' % scene_name)['scene']
        scene.image.size = size
        r = Renderer(scene, seed=0, min_waves_per_simd=w[1], speculate=w[0])
        if tape is None:
            tape = r.make_tape(0, passes)
        r.upload_tape(tape)
        r.render_resident(0, passes)
        rs[w] = r
    times = {k: [] for k in rs}
    for _ in range(rounds):
        for w, r in rs.items():
            times[w].append(r.render_resident(0, passes))
    ref = None
    for w, r in rs.items():
        r.img.fill(0); r.render_resident(0, passes)
        img = r.img.get()
        if ref is None: ref = img
        print('(speculate, waves/SIMD hint)', w, 'median %.3f ms' % np.median(times[w]), 'min %.3f' % min(times[w]),
              '%.0f Msamples/s' % (r.n_pixels * passes / np.median(times[w]) / 1e3), 'identical', np.array_equal(img, ref))

if __name__ == '__main__':
    main(*(sys.argv[1:2] or ['scene-dev']))
