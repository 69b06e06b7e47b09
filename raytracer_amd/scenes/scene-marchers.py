"""Ray-marching stress scene at 3840 x 2160 (BASELINE C5)."""
from raytracer_amd.scenes.library import scene_marchers as make_scene

scene = make_scene()
