#!/usr/bin/env python3
"""Command line entry point, same usage as the reference's clray.py (/root/reference/clray.py:15-101):

    python clray.py [-a] [-n] [-itr N] [-o "hiprtc options"] scene.py

The scene file defines a module-level `scene`.  Renders `scene.samples_per_pixel` passes in
batches (one kernel launch per batch), writes out.png and out.raw.npy (the float sum over all
passes, also the `--append` resume format) at every refresh and prints rays/s.
Extra flags: --size WxH, --spp N, --seed S, --device D, --batch B, --rng, and
--gpus N [--partition passes|tiles]: the render is shared by N GPUs of this node, one process per
GPU under torch.distributed.run, with one RCCL exchange per refresh (raytracer_amd/multi_gpu.py).
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
    ap.add_argument('--light-samples', type=int, default=1,
                    help="points of the segment's light per next-event estimate (1 = the reference's estimator)")
    ap.add_argument('--gpus', type=int, default=1, help='GPUs of this node to share the render')
    ap.add_argument('--partition', choices=['passes', 'tiles'], default='passes',
                    help='passes: GPU g renders passes g, g+N, ... (one all-reduce per refresh); '
                         'tiles: GPU g renders its slice of the frame (one all-gather per refresh)')
    ap.add_argument('scene')
    args = ap.parse_args()

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from raytracer_amd import multi_gpu
    rank, local_rank, world = multi_gpu.worker_identity()
    if args.gpus > 1 and world == 1:
        # not yet a worker: start one process per GPU (before anything here touches a GPU) and leave
        raise SystemExit(multi_gpu.launch_workers(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    if world != max(args.gpus, 1):
        raise SystemExit("--gpus %d, but started with WORLD_SIZE=%d" % (args.gpus, world))

    startup_time = time.time()
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
    n_rays = scene.image.size[0] * scene.image.size[1]
    shard = multi_gpu.Shard(args.partition, rank, world, n_rays)
    # CLRAY_FORCE_DIST=1 runs the several-GPU code path (torch accumulator, RCCL exchange) with
    # whatever number of workers there is, also one (development check on a one-GPU box)
    use_dist = world > 1 or bool(os.environ.get('CLRAY_FORCE_DIST'))
    if use_dist:
        import torch
        dist, args.device = multi_gpu.start_process_group(local_rank)
        args.seed = multi_gpu.common_seed(args.seed, dist, torch.device('cuda', args.device))
        if args.append and rank == 0 and '--seed' in sys.argv:
            print("WARNING: --append with the --seed of the earlier run adds the same samples again; "
                  "leave --seed out (a fresh common seed is drawn) or change it")
    renderer = raytracer_amd.Renderer(scene, args, seed=args.seed, device_index=args.device, rng_mode=args.rng,
                                      light_samples=args.light_samples, overlap_parts='auto', steps_in_flight='auto', **shard.renderer_arguments())
    shard.configure(renderer)
    # Timing the occupancy candidates costs six more program builds (about a minute).  Kernels of
    # analytic primitives run the same at every candidate; ray marchers and meshes differ by up
    # to 20 %, which pays off from a few minutes of rendering on.
    if not renderer.speculate and renderer.n_pixels * scene.samples_per_pixel >= 2e11:
        renderer.tune_occupancy(passes=8)
    acc = renderer.acc
    if use_dist:
        # the accumulator lives in a torch tensor so that RCCL can combine the ranks' results
        frame = torch.zeros((shard.frame_rows, 4), dtype=torch.float32, device='cuda')
        renderer.use_image_buffer(shard.my_rows(frame)[:renderer.n_pixels].data_ptr())
        combined = torch.empty_like(frame)

    def whole_image():
        """(H, W, 3) sums over all passes of all GPUs so far (every rank takes part, rank 0 uses it)."""
        if not use_dist:
            return renderer.get_image()
        combined.copy_(frame)
        shard.combine(combined, dist)
        if world == 1:
            dist.all_reduce(combined)            # one rank: the sum of one accumulator
        from raytracer_amd.renderer import unpermute_rays
        return unpermute_rays(combined[:n_rays, :3].cpu().numpy(), renderer.full_image_order,
                              renderer.img_shape).astype(np.float64)

    total = scene.samples_per_pixel
    batch = args.batch or max(1, min(args.itr_per_refresh, 64))

    def schedule():
        """(first pass, passes, refresh afterwards) of every batch; refreshes come after a tenth of
        -itr passes at first and every -itr passes from then on, as in the reference's loop."""
        at, next_refresh = 0, min(max(1, args.itr_per_refresh // 10), total)
        while at < total:
            n = min(batch, total - at, max(next_refresh - at, 1))
            at += n
            refresh = at >= next_refresh or at == total
            if refresh:
                step = args.itr_per_refresh if at >= args.itr_per_refresh else max(1, args.itr_per_refresh // 10)
                next_refresh = min(at + step, total)
            yield at - n, n, refresh

    # The passes up to the next refresh form one interval: their random numbers are drawn on the
    # host (the reference's generator, in its order) and made resident in one piece, and the
    # interval's batches are launched back to back without a host wait in between - as launch
    # chains on slices of the frame, or with two steps in flight for a pass-split program
    # (Renderer.render_resident).  A helper thread draws interval i + 1 while the GPU renders
    # interval i (the wait at the end of an interval happens outside the interpreter lock).
    from concurrent.futures import ThreadPoolExecutor
    intervals, current = [], []
    for first, n, refresh in schedule():
        current.append((first, n))
        if refresh:
            intervals.append(current)
            current = []
    helper = ThreadPoolExecutor(1)

    def draw(interval):
        return renderer.make_tape(interval[0][0], sum(n for _, n in interval))

    drawn = helper.submit(draw, intervals[0]) if intervals else None
    for i, interval in enumerate(intervals):
        # passes [done, done + n) of the global sequence; with the passes partition this rank
        # renders those whose index is rank modulo the number of GPUs
        done, n = interval[0][0], sum(count for _, count in interval)
        tape = drawn.result()
        if i + 1 < len(intervals):
            drawn = helper.submit(draw, intervals[i + 1])
        mine = shard.local_tape(tape) if shard.pass_stride == 1 else tuple(
            x[(shard.pass_offset - done) % world::world] if x is not None else None for x in tape)
        n_mine = mine[0].shape[0]
        if n_mine:
            renderer.upload_tape(mine, first_pass=done + (shard.pass_offset - done) % shard.pass_stride)
            acc.timer_begin()
            for first in range(0, n_mine, batch):
                renderer.render_resident(first, min(batch, n_mine - first), timed=False)
            renderer.join_parts()
            seconds = acc.timer_end() * 1e-3          # waits for the interval's launches
            acc._record('clr_render', seconds, seconds)
        done += n
        elapsed = time.time() - startup_time
        per_second = done / elapsed
        if rank == 0:
            print('%d/%d, depth: %d, %d rays/s elapsed: %.2f s, eta: %.1f min' % (
                done, total, scene.max_bounces - 1, int(n_rays * per_second),
                elapsed, (total - done) / per_second / 60.0))
        refresh = True
        if refresh:
            imgdata = whole_image()
            if rank != 0:
                continue
            print('image mean:', np.mean(np.ravel(imgdata)))
            if not args.no_window:
                image.show(imgdata)
            image.save_raw(RAW_OUTPUT_FILE, imgdata)
            try:
                if use_dist:
                    raise NotImplementedError       # the device tone mapper reads one GPU's accumulator
                # 8-bit image straight from the device accumulator (tonemap.py)
                from PIL import Image as PilImage
                PilImage.fromarray(renderer.get_display_image(scene.image, base=image.data), 'RGB').save(PNG_OUTPUT_FILE)
            except NotImplementedError:
                image.save_png(PNG_OUTPUT_FILE, imgdata)        # several GPUs, other reductions: host pipeline
            acc.output_profiling_info()
    if rank == 0:
        acc.output_profiling_info()
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
