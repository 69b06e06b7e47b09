import sys, time, numpy as np
sys.path.insert(0, '.')
from raytracer_amd.renderer import Renderer
from raytracer_amd.scenes.library import scene_mesh
from raytracer_amd.scenes.default_scenes import BoxScene
from raytracer_amd.objects import Sphere
s = BoxScene()
s.get_object('light').bidirectional_light = True
s.add_object(Sphere((0, 0, 1), 0.8), 'white', 'knot')
s.image.size = (1920, 1080); s.samples_per_pixel = 16; s.max_bounces = 4; s.min_bounces = 2
s.camera_position = (1.5, -4.2, 2.2); s.direct_camera_towards((0, 0, 0.9))
for split in (False, True):
    r = Renderer(s, seed=1, split_passes=split)
    tape = r.make_tape(0, 16)
    r.render_tape(tape)
    ms, _ = r.render_tape(tape)
    print('box + sphere, split', split, 'ms per 16 passes', ms)
