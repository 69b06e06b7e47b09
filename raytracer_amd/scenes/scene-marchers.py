"""Ray-marching stress scene (BASELINE config C5): an implicit quartic surface, two signed
distance fields and both quaternion Julia set tracers in the box room, 3840 x 2160.
Shape arguments follow /root/reference/scenes/scene-test.py:53-55,74."""
import numpy as np
from raytracer_amd.scenes.default_scenes import BoxScene
from raytracer_amd.objects import (ImplicitSurface, DistanceField, QuaternionJuliaSet,
                                   QuaternionJuliaSet2)

JULIA_C = (-0.2, -0.4, -0.4, -0.4)
R = 0.45

ROUNDED_BOX_TORUS = (
    "float qx = fabs(x) - 0.30f, qy = fabs(y) - 0.30f, qz = fabs(z) - 0.30f; "
    "float ox = fmax(qx, 0.0f), oy = fmax(qy, 0.0f), oz = fmax(qz, 0.0f); "
    "float box = sqrt(ox*ox + oy*oy + oz*oz) + fmin(fmax(qx, fmax(qy, qz)), 0.0f) - 0.08f; "
    "float ring = sqrt(x*x + y*y) - 0.42f; "
    "float torus = sqrt(ring*ring + z*z) - 0.07f; "
    "dist = fmin(box, torus)")


def make_scene():
    s = BoxScene()
    s.get_object('light').bidirectional_light = True
    z = R
    s.add_object(ImplicitSurface('x**4 - 5*x**2 + y**4 - 5*y**2 + z**4 - 5*z**2 + 11.8',
                                 (-1.2, 0.8, z), R * 0.5, 4), 'mirror', 'implicit')
    s.add_object(DistanceField("dist = sqrt(x*x + y*y + z*z) - %g" % R, center=(0.0, 1.6, z)),
                 'glass', 'sdf-sphere')
    s.add_object(DistanceField(ROUNDED_BOX_TORUS, center=(1.2, 0.6, z), bndR=0.7),
                 'white', 'sdf-box-torus')
    s.add_object(QuaternionJuliaSet(JULIA_C, 4, center=(-0.4, -0.6, z), scale=R), 'white', 'julia-ia')
    s.add_object(QuaternionJuliaSet2(JULIA_C, 4, center=(0.9, -0.9, z), scale=R), 'wax', 'julia-de')
    for tracer in (o.tracer for o in s.objects[-5:]):
        tracer.rotate(axis='z', deg=-30)
    s.image.size = (3840, 2160)
    s.samples_per_pixel = 16
    s.max_bounces = 4
    s.min_bounces = 2
    s.direct_camera_towards((0, 0.3, 0.3))
    return s


scene = make_scene()
