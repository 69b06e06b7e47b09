"""Development tool: print the timings Renderer.tune_occupancy bases its choice on."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import load_scene
from raytracer_amd.renderer import Renderer

for name in sys.argv[1:] or ['scene-dev', 'scene-three-objects']:
    r = Renderer(load_scene(name, (1920, 1080)), seed=0)
    for passes in (4, 16, 32):
        best = r.tune_occupancy(passes=passes)
        print(name, 'passes', passes, 'best', best, {k: round(v, 3) for k, v in r.occupancy_timings.items()})
