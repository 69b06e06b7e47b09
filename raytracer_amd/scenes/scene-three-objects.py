"""Cylinder intersections, box and glass dodecahedron in the box room (BASELINE C2)."""
from raytracer_amd.scenes.library import scene_three_objects as make_scene

scene = make_scene()
