"""Development: where the cycles of the pooled octree traversal go (run through gpurun).

CLR_MESH_PROFILE=1 makes the Octree template keep cycle counters per phase in the module global
clr_mesh_phases (one wave = one count): entry, A locate leaf, B pooled triangle tests, C hop.
usage: python tools/mesh_phases.py [n_u x n_v] [passes] [two_level 0/1]"""
import os
import sys
os.environ['CLR_MESH_PROFILE'] = '1'
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd.renderer import Renderer
from raytracer_amd.scenes.library import scene_mesh

grid = sys.argv[1] if len(sys.argv) > 1 else '1000x500'
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n_u, n_v = (int(v) for v in grid.split('x'))
two_level = bool(int(sys.argv[3])) if len(sys.argv) > 3 else None
scene = scene_mesh(n_u, n_v, two_level=two_level)
r = Renderer(scene, seed=1, split_passes=False)
plain_ms = None
tape = r.make_tape(0, passes)
r.acc.set_global('clr_mesh_phases', np.zeros(64 * 16, np.uint64))
ms, _ = r.render_tape(tape)
c = r.acc.get_global('clr_mesh_phases', (64, 16), np.uint64).astype(float).sum(axis=0)
calls, steps = c[0], c[5]
total = c[1] + c[2] + c[3] + c[4]
print('grid %s: %.2f ms for %d passes; %.3g wave calls, %.1f steps per call, %.1f lanes per call' %
      (grid, ms, passes, calls, steps / calls, c[9] / calls))
print('cycles per wave call %.0f: entry %.0f, A locate %.0f, B tests %.0f, C hop %.0f' %
      (total / calls, c[1] / calls, c[2] / calls, c[3] / calls, c[4] / calls))
print('per step: A %.0f, B %.0f (B ran in %.0f %% of the steps, %.1f items each), C %.0f; lanes still walking %.1f' %
      (c[2] / steps, c[3] / steps, 100 * c[7] / steps, c[8] / max(c[7], 1), c[4] / steps, c[6] / steps))
print('parts per step: A wait + descent %.0f, A exit + hop %.0f; B set-up %.0f, B rounds %.0f (two-level records: %s)' %
      (c[10] / steps, c[11] / steps, c[12] / steps, c[13] / steps, [o.tracer.two_level for o in scene.objects if o.name == 'knot'][0]))
