"""Development tool: print the timings Renderer.tune_occupancy bases its choice on."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import load_scene
from raytracer_amd.renderer import Renderer
from raytracer_amd.hip_compiler import PARITY_OPTIONS

extra = os.environ.get('EXTRA_OPTS', '').split()

candidates = (None, 4, 5, 6, 7, 8)
for name in sys.argv[1:] or ['scene-dev', 'scene-three-objects']:
    r = Renderer(load_scene(name, (1920, 1080)), seed=0, build_options=PARITY_OPTIONS + extra)
    passes = 4 if 'mesh' in name or 'march' in name else 32
    best = r.tune_occupancy(candidates=candidates, passes=passes)
    print(name, 'passes', passes, 'best', best, {k: round(v, 3) for k, v in r.occupancy_timings.items()}, flush=True)
