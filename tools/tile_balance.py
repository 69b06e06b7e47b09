"""Development (run through gpurun): how evenly the two forms of the tile partition of
multi_gpu.Shard share the work of a frame.  The render kernel records the cycles of every block of
256 rays (CLR_BLOCK_PROFILE=1); a rank's load is the sum over the blocks of its tile.

  python tools/tile_balance.py scene-marchers 16 3840x2160
"""
import os
import sys
os.environ['CLR_BLOCK_PROFILE'] = '1'
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_scene
from raytracer_amd.renderer import Renderer
from raytracer_amd.multi_gpu import Shard

name = sys.argv[1] if len(sys.argv) > 1 else 'scene-marchers'
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
size = tuple(int(v) for v in sys.argv[3].split('x')) if len(sys.argv) > 3 else (3840, 2160)
r = Renderer(load_scene(name, size), seed=0, split_passes=False)
r.tune_occupancy(passes=4)
tape = r.make_tape(0, passes)
r.upload_tape(tape)
n_blocks = (r.n_pixels + 255) // 256
assert n_blocks <= 65536
r.render_resident(0, passes)
r.acc.set_global('clr_block_cycles', np.zeros(65536, np.uint64))
ms = r.render_resident(0, passes)
cycles = r.acc.get_global('clr_block_cycles', (65536,), np.uint64)[:n_blocks].astype(float)
per_ray = np.repeat(cycles, 256)[:r.n_pixels] / 256.0
print('%s %dx%d, %d passes: %.2f ms; %d blocks, cycles mean %.0f max %.0f' % (
    name, size[0], size[1], passes, ms, n_blocks, cycles.mean(), cycles.max()))
print('load of the busiest rank / mean load (1.00 = even), by tile form:')
for world in (2, 4, 8):
    row = []
    for band_blocks in (0, 1, 4, 16, 64):
        loads = []
        for g in range(world):
            shard = Shard('tiles', g, world, r.n_pixels, band_blocks=band_blocks)
            if shard.ray_bands is None:
                first, count = shard.ray_range
                loads.append(per_ray[first:first + count].sum())
            else:
                _, _, band = shard.ray_bands
                loads.append(sum(per_ray[b:b + band].sum() for b in range(g * band, r.n_pixels, world * band)))
        loads = np.array(loads)
        row.append('%s %.3f' % ('contiguous' if band_blocks == 0 else 'bands of %d' % band_blocks, loads.max() / loads.mean()))
    print('  %d ranks: %s' % (world, ', '.join(row)))
