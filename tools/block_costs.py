"""Development: how long the blocks of clr_render run, and what dispatching the long ones first
would buy (run through gpurun).  CLR_BLOCK_PROFILE=1 makes the kernel record the cycles of the
slowest wave of every block; the blocks are then re-ordered on the host (a permutation of the
ray order in units of 256 rays: cam, image_order and img alike), which changes no pixel."""
import os
import sys
os.environ['CLR_BLOCK_PROFILE'] = '1'
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import load_scene
from raytracer_amd.renderer import Renderer

name = sys.argv[1] if len(sys.argv) > 1 else 'scene-dev'
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 32
# optional: frame size WxH and "split" for the pass-split program (then only the distribution of
# the block times is printed: blocks = 256 lanes of one pass group)
size = tuple(int(v) for v in sys.argv[3].split('x')) if len(sys.argv) > 3 else (1920, 1080)
split = len(sys.argv) > 4 and sys.argv[4] == 'split'
r = Renderer(load_scene(name, size), seed=0, split_passes=split)
r.tune_occupancy(passes=8)
tape = r.make_tape(0, passes * 3)
r.upload_tape(tape)
n_blocks = (r.n_pixels + 255) // 256
if split:
    groups = r._split_args(passes)['pass_groups']
    n_blocks = (((r.n_pixels + 63) // 64) * 64 * groups + 255) // 256


def timed(label, rounds=6):
    best = []
    for i in range(rounds):
        r.acc.set_global('clr_block_cycles', np.zeros(65536, np.uint64))
        ms = r.render_resident((i % 3) * passes, passes)
        best.append(ms)
    print('%s: ms per %d passes: min %.4f median %.4f' % (label, passes, min(best), float(np.median(best))))
    return r.acc.get_global('clr_block_cycles', (65536,), np.uint64)[:n_blocks].astype(float)


cycles = timed('natural order')
print('block cycles: mean %.0f, std %.0f, min %.0f, max %.0f; sum / max = %.1f' %
      (cycles.mean(), cycles.std(), cycles.min(), cycles.max(), cycles.sum() / cycles.max()))
if split:
    clock = 2.4e6          # cycles per ms, nominal
    print('pass groups %d, blocks %d; block ms: mean %.3f, median %.3f, 90%% %.3f, 99%% %.3f, max %.3f' % (
        groups, n_blocks, cycles.mean() / clock, np.median(cycles) / clock, np.percentile(cycles, 90) / clock,
        np.percentile(cycles, 99) / clock, cycles.max() / clock))
    sys.exit(0)
# longest first
order = np.argsort(-cycles, kind='stable')
rays = (order[:, None] * 256 + np.arange(256)[None, :]).ravel()
rays = rays[rays < r.n_pixels]
assert rays.size == r.n_pixels
before = r.img.get().copy()
r.cam.set(np.ascontiguousarray(r.host_cam[rays]))
r.img.set(np.ascontiguousarray(before[rays]))
timed('longest blocks first')
order = np.argsort(cycles, kind='stable')
rays2 = (order[:, None] * 256 + np.arange(256)[None, :]).ravel()
rays2 = rays2[rays2 < r.n_pixels]
r.cam.set(np.ascontiguousarray(r.host_cam[rays2]))
timed('shortest blocks first')
