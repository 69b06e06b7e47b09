"""Development: do two (or more) stream-ordered chains of launches, each on its own slice of the
frame, hide each other's drain?  (run through gpurun)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import load_scene
from raytracer_amd.renderer import Renderer
from raytracer_amd.accelerator import RenderArgs

name = sys.argv[1] if len(sys.argv) > 1 else 'scene-dev'
P, steps = 32, 20
r = Renderer(load_scene(name, (1920, 1080)), seed=0, split_passes=False)
r.tune_occupancy(passes=8)
tape = r.make_tape(0, P * steps)
r.upload_tape(tape)
n = r.n_pixels


def launch(first, count, lo, hi, stream):
    args = RenderArgs(
        img=r.img.ptr + 16 * lo, cam=r.cam.ptr + 16 * lo,
        pass_tape=r.dev_pass_tape[first:first + count].ptr,
        bounce_tape=r.dev_bounce_tape[first:first + count].ptr,
        material_rows=None, vector_data=r.dev_vector_data.ptr, integer_data=r.dev_integer_data.ptr,
        n_pixels=hi - lo, n_passes=count, regenerate=int(r.regenerate), debug_dump=None,
        **r._rng_args(first))
    r.acc.render_passes(args, stream=stream, timed=False)


reference = None
for parts in (1, 2, 3, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    bounds = [((n * p // parts + 255) // 256) * 256 for p in range(parts)] + [n]
    bounds[0] = 0
    for rep in range(3):
        r.img.fill(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            for p in range(parts):
                launch(k * P, P, bounds[p], bounds[p + 1], streams[p].cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
    image = r.img.get()
    if reference is None:
        reference = image
    if parts == 2:
        # the same with the second chain half a launch behind the first from the start
        for rep in range(3):
            r.img.fill(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            launch(0, P // 2, bounds[1], bounds[2], streams[1].cuda_stream)
            for k in range(steps):
                launch(k * P, P, bounds[0], bounds[1], streams[0].cuda_stream)
                if k + 1 < steps:
                    launch(k * P + P // 2, P, bounds[1], bounds[2], streams[1].cuda_stream)
            launch(steps * P - P // 2, P // 2, bounds[1], bounds[2], streams[1].cuda_stream)
            torch.cuda.synchronize()
            ms2 = (time.perf_counter() - t0) * 1e3 / steps
        same2 = np.array_equal(r.img.get().view(np.uint32), reference.view(np.uint32))
        print('2 parts, staggered: %.4f ms per %d passes = %.1f Msamples/s, image identical: %s' % (ms2, P, n * P / ms2 / 1e3, same2))
    same = np.array_equal(image.view(np.uint32), reference.view(np.uint32))
    print('%d part(s): %.4f ms per %d passes = %.1f Msamples/s, image identical: %s' % (parts, ms, P, n * P / ms / 1e3, same))
