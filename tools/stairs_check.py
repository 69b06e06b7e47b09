import sys, time, runpy
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from raytracer_amd.renderer import Renderer
from oracle import oracle
t = time.time()
scene = runpy.run_path('raytracer_amd/scenes/scene-stairs.py')['scene']
scene.image.size = (96, 64)
r = Renderer(scene, seed=2)
print('objects', len(scene.objects), 'baked', r.bake_parameters, 'lds', r.materials_in_lds, 'split', r.split_passes, 'build s', round(time.time() - t, 1), flush=True)
osc = oracle.OracleScene(r)
tape = r.make_tape(0, 6)
r.render_tape(tape)
oimg, _, _ = osc.render(tape)
img = r.img.get()
print('equal bits', np.array_equal(img.view(np.uint32), oimg.view(np.uint32)), 'mean', img[:, :3].mean())
