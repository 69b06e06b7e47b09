"""Development (run through gpurun): what each shape of a scene costs - the scene is rendered with
one shape left out at a time (same frame, same tape, same occupancy hint) and the launch times are
compared with the full scene's.  Not additive (the shapes share waves), but it ranks them.

  python tools/class_costs.py scene-test 320x240 16 [hint]
"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_scene
from raytracer_amd.renderer import Renderer

name = sys.argv[1] if len(sys.argv) > 1 else 'scene-test'
size = tuple(int(v) for v in sys.argv[2].split('x')) if len(sys.argv) > 2 else (320, 240)
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 16
hint = int(sys.argv[4]) if len(sys.argv) > 4 else 6


def label(index, obj):
    """the object's name, or class and material where the scene gives none"""
    return obj.name or '%s-%s-%d' % (obj.tracer.__class__.__name__, obj.material, index)


def timed(without=None):
    scene = load_scene(name, size)
    if without is not None:
        scene.objects[:] = [o for i, o in enumerate(scene.objects) if label(i, o) != without]
    r = Renderer(scene, seed=0, min_waves_per_simd=hint)
    tape = r.make_tape(0, passes)
    r.upload_tape(tape)
    r.render_resident(0, passes)
    ms = [r.render_resident(0, passes) for _ in range(7)]
    meta = r.acc.last_metadata.get('clr_render', {})
    return float(np.median(ms)), meta.get('.vgpr_count'), meta.get('.vgpr_spill_count')


full, regs, spills = timed()
print('%s %dx%d, %d passes, hint %d: %.3f ms per launch (%s registers, %s spilled)' % (name, size[0], size[1], passes, hint, full, regs, spills), flush=True)
names = [label(i, o) for i, o in enumerate(load_scene(name, size).objects)
         if o.name not in ('wall', 'floor', 'ceiling', 'light') and o.material != 'sky']
rows = []
for shape in names:
    ms, regs, spills = timed(shape)
    rows.append((full - ms, shape, ms, regs, spills))
    print('  without %-28s %.3f ms  (%+.3f)  %s registers, %s spilled' % (shape, ms, ms - full, regs, spills), flush=True)
print('ranked by what leaving them out saves:')
for saved, shape, ms, regs, spills in sorted(rows, reverse=True):
    print('  %-28s %.3f ms = %4.1f %%' % (shape, saved, 100 * saved / full))
