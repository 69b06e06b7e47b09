# scene-dev with the spectral material set and shader (one wavelength per sample pass)
from raytracer_amd.scenes.library import scene_dev as make_scene
scene = make_scene(spectral=True)
