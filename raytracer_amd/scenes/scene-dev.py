"""Box room with a stretched mirror sphere (BASELINE C3); make_scene(spectral=True) for the spectral variant."""
from raytracer_amd.scenes.library import scene_dev as make_scene

scene = make_scene()
