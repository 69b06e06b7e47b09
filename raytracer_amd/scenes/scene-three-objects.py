"""Box room with two triple-cylinder intersections (mirror, wax), a red box and a glass
dodecahedron.  Values: /root/reference/scenes/scene-three-objects.py:7-64 (BASELINE C2)."""
import numpy as np
from raytracer_amd.scenes.default_scenes import BoxScene
from raytracer_amd.objects import (HalfSpace, ConvexIntersection, CylinderComponent,
                                   Parallelepiped, Dodecahedron)
from raytracer_amd.utils import vec_norm


def three_cylinders(origin, axes, radius):
    return ConvexIntersection(origin, [CylinderComponent(a, radius) for a in axes])


def make_scene():
    s = BoxScene()
    m = s.materials
    m['floor'] = {'diffuse': 0.1, 'reflection': 0.15, 'reflection_blur': 0.05}
    m['glass']['transparency'] = (0.4, 0.6, 0.9)
    m['glass']['ior'] = 1.5
    m['wax']['volume_absorption'] = (1.0 - np.array((0.2, 0.8, 0.4))) * 7.0
    m['wax']['diffuse'] = (0.02, 0.03, 0.04)
    m['wax']['reflection'] = (0.01, 0.04, 0.02)

    s.get_object('floor').material = 'floor'
    s.get_object('ceiling').material = 'white'
    s.get_object('light').bidirectional_light = True
    for wall in s.get_objects('wall'):
        wall.material = 'white'

    s.add_object(HalfSpace((-1, -1, -2), 5), 'sky')
    s.add_object(three_cylinders((-0.2, 2.5, 1.0), [(1, 1, 0), (0, 1, 0), (1, 0, 1)], 1), 'mirror')
    cyl_r = .4
    s.add_object(three_cylinders((-0.7, -0.8, cyl_r), [(1, 0, 0), (0, 1, 0), (0, 0, 1)], cyl_r), 'wax')
    s.add_object(Parallelepiped((1.3, -0.5, 0.0), (1, 0, 0), (0, 1.3, 0), (0, 0, 0.6)), 'red')
    dodecahedron_center = (1.8, 0.2, .5 + 0.6)
    s.add_object(Dodecahedron(dodecahedron_center, 0.5), 'glass')

    s.image.size = (1024, 768)
    s.samples_per_pixel = 10000
    s.quasirandom = False
    s.gamma = 1.6
    s.max_bounces = 5
    s.camera_fov = 60
    s.camera_position = (-2, -3, 1.5)
    s.direct_camera_towards((0, 0, 0.5))
    s.camera_sharp_distance = vec_norm(np.array(dodecahedron_center) - np.array(s.camera_position))
    return s


scene = make_scene()
