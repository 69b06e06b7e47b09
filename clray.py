#!/usr/bin/env python3
"""Command line entry point, same usage as the reference's clray.py (/root/reference/clray.py:15-101):

    python clray.py [-a] [-n] [-itr N] [-o "hiprtc options"] scene.py

The scene file defines a module-level `scene`.  Renders `scene.samples_per_pixel` passes in
batches (one kernel launch per batch), writes out.png and out.raw.npy (the float sum over all
passes, also the `--append` resume format) at every refresh and prints rays/s.
Extra flags: --size WxH, --spp N, --seed S, --device D, --batch B.
"""
import argparse
import os
import runpy
import sys
import time

import numpy as np

PNG_OUTPUT_FILE = 'out.png'
RAW_OUTPUT_FILE = 'out.raw.npy'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-a', '--append', action='store_true')
    ap.add_argument('-n', '--no_window', action='store_true')
    ap.add_argument('-itr', '--itr_per_refresh', type=int, default=100)
    ap.add_argument('-o', '--cl_build_options', default='')
    ap.add_argument('-c', '--choose_opencl_context', action='store_true')
    ap.add_argument('--size', default=None)
    ap.add_argument('--spp', type=int, default=None)
    ap.add_argument('--seed', type=int, default=None)
    ap.add_argument('--device', type=int, default=0)
    ap.add_argument('--batch', type=int, default=None, help='passes per kernel launch')
    ap.add_argument('--rng', choices=['tape', 'philox'], default='tape',
                    help="tape: the reference's random numbers, shared by all pixels of a pass; "
                         "philox: every pixel draws its own (less correlated noise)")
    ap.add_argument('scene')
    args = ap.parse_args()

    startup_time = time.time()
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import raytracer_amd
    from raytracer_amd.imgutils import Image
    raytracer_amd.install_flat_aliases()          # `from objects import *` in reference scene files
    sys.path.append(os.path.dirname(os.path.abspath(args.scene)))
    scene = runpy.run_path(args.scene)['scene']
    if args.size:
        scene.image.size = tuple(int(v) for v in args.size.lower().split('x'))
    if args.spp:
        scene.samples_per_pixel = args.spp

    image = Image(RAW_OUTPUT_FILE if args.append else None, settings=scene.image)
    renderer = raytracer_amd.Renderer(scene, args, seed=args.seed, device_index=args.device, rng_mode=args.rng)
    if scene.samples_per_pixel >= 64:
        renderer.tune_occupancy(passes=8)       # pays off for long renders only
    acc = renderer.acc

    total = scene.samples_per_pixel
    batch = args.batch or max(1, min(args.itr_per_refresh, 64))
    done = 0
    next_refresh = min(max(1, args.itr_per_refresh // 10), total)
    while done < total:
        n = min(batch, total - done, max(next_refresh - done, 1))
        renderer.render_samples(done, n)
        done += n
        elapsed = time.time() - startup_time
        per_second = done / elapsed
        print('%d/%d, depth: %d, %d rays/s elapsed: %.2f s, eta: %.1f min' % (
            done, total, scene.max_bounces - 1, int(renderer.rays_per_sample() * per_second),
            elapsed, (total - done) / per_second / 60.0))
        if done >= next_refresh or done == total:
            imgdata = renderer.get_image()
            print('image mean:', np.mean(np.ravel(imgdata)))
            if not args.no_window:
                image.show(imgdata)
            image.save_raw(RAW_OUTPUT_FILE, imgdata)
            try:
                # 8-bit image straight from the device accumulator (tonemap.py)
                from PIL import Image as PilImage
                PilImage.fromarray(renderer.get_display_image(scene.image, base=image.data), 'RGB').save(PNG_OUTPUT_FILE)
            except NotImplementedError:
                image.save_png(PNG_OUTPUT_FILE, imgdata)        # flares etc.: host pipeline
            acc.output_profiling_info()
            step = args.itr_per_refresh if done >= args.itr_per_refresh else max(1, args.itr_per_refresh // 10)
            next_refresh = min(done + step, total)
    acc.output_profiling_info()


if __name__ == '__main__':
    main()
