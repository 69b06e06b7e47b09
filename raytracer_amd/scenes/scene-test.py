"""Every shape class on a grid (BASELINE C1)."""
from raytracer_amd.scenes.library import scene_test as make_scene

scene = make_scene()
