#!/usr/bin/env python3
"""Throughput benchmark of the hot path: sample passes of scene-dev at 1080p on N MI355X.

  python bench.py --gpus N --steps K --warmup W

A step = one launch of the fused render kernel over `--passes-per-step` sample passes of the
whole frame (inputs - camera rays, random tape, scene tables - already resident in HBM).
With N > 1 (launched by torch.distributed.run, one process per GPU) the passes are striped
over the ranks (pass j belongs to rank j mod N), every rank keeps a private
accumulator and one RCCL all-reduce at the end of the timed region sums them; per-GPU work
is fixed, so scaling is "weak".  Rank 0 prints one JSON line.
"""
import argparse
import json
import math
import os
import runpy
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X peaks (/opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters)
PEAK_FP32_VECTOR_TFLOPS = 157.3        # FP32 vector peak: one v_fma_f32 (2 flop) per lane per clock
PEAK_HBM_GBS = 8000.0                  # HBM3E spec; ~6300 achievable (same guide)
# algorithmic work is counted by the oracle's instrumented build over ALL timed passes, on every
# FLOP_PIXEL_STRIDE-th pixel or sparser (a stride coprime to the 32-pixel blocks of the ray order)
FLOP_COUNT_SAMPLES = 48e6
# passes the random tape always holds, so that qdirs and the draws of the timed passes do not
# depend on --steps / --warmup (the timed passes come first, warm-up passes after them)
TAPE_SAMPLES_PER_PIXEL = 4096
ROOFLINE_LAUNCHES = 10                 # lone launches timed for roofline.kernel_ms_avg


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)     # 0.15 s of launches on the default workload
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--passes-per-step', type=int, default=32)
    ap.add_argument('--scene', default='scene-dev')
    ap.add_argument('--size', default='1920x1080')
    ap.add_argument('--mesh-grid', default='1000x500', help='scene-mesh: quads along / around the tube')
    ap.add_argument('--mesh-overlap', choices=['sat', 'reference'], default='sat',
                    help="scene-mesh: rule that files faces under octree leaves (same hits either way): "
                         "'sat' triangle against box, 'reference' the reference's bounding-sphere rule")
    ap.add_argument('--mesh-pooled', type=int, default=None,
                    help='scene-mesh: 1/0 triangle tests pooled over the wave or per lane (default: the Octree class decides)')
    ap.add_argument('--mesh-two-level', type=int, default=None,
                    help='scene-mesh: 1 = node records that take the descent down two levels per memory access, 0 = plain records')
    ap.add_argument('--mesh-ropes', type=int, default=None,
                    help='scene-mesh: 1/0 next leaf through neighbour links or by the descent from the common ancestor (default: links)')
    ap.add_argument('--mesh-depth', type=int, default=None, help='scene-mesh: octree depth (default: by face count)')
    ap.add_argument('--mesh-leaf', type=int, default=None, help='scene-mesh: faces per leaf before a node splits (default 8)')
    ap.add_argument('--marchers', default=None,
                    help='scene-marchers: comma-separated names of the ray-marched objects to keep (development)')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--regenerate', type=int, default=None, help='path regeneration 1/0 (default: on with --rng philox)')
    ap.add_argument('--rng', choices=['tape', 'philox'], default='tape',
                    help="tape: the reference's per-frame random numbers (parity mode, the default); "
                         "philox: every pixel draws its own on the device")
    ap.add_argument('--build-options', default=None,
                    help='hiprtc options instead of the parity set (-O3 -ffp-contract=off)')
    ap.add_argument('--bake', type=int, default=None, help='1/0: bake scene parameters into the program')
    ap.add_argument('--partition', choices=['passes', 'tiles'], default='passes',
                    help='passes: pass j on rank j mod N (weak scaling); tiles: every rank renders '
                         'all passes of its own slice of the frame (strong scaling)')
    ap.add_argument('--occupancy-hint', default='auto',
                    help="waves per SIMD the render kernel is compiled for: 'auto' (time the candidates "
                         "before the run), 'none' or a number")
    ap.add_argument('--split-passes', type=int, default=None,
                    help='1/0: pass-split program (several lanes per pixel) on/off; default: on for small frames')
    ap.add_argument('--pass-groups', type=int, default=None,
                    help='pass-split program: lanes per pixel at least (default: 4 for mesh scenes, else 1)')
    ap.add_argument('--streams', default='auto',
                    help='launch chains: every step goes out as this many launches on slices of the frame, each '
                         'slice on its own stream, so that one slice fills the GPU while another drains (1: one '
                         'stream; auto: 2, and 1 for a pass-split program)')
    ap.add_argument('--in-flight', default='auto',
                    help='pass-split programs (small frames, meshes): this many steps log their addends at the '
                         'same time, on streams and logs of their own; the sums are added in step order (auto: 2)')
    ap.add_argument('--sorted-paths', type=int, default=None,
                    help='per-pixel random numbers: 1/0 the paths of a block are sorted before every segment (default: the renderer decides)')
    ap.add_argument('--outline', default=None,
                    help="long shape tests (ray marchers, mesh traversals) as functions of their own: 0, 1, "
                         "'marchers' or 'meshes' (default: the renderer's policy)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0,
                    help='target duration of the CPU baseline sample')
    return ap.parse_args()


def stripe(tape, rank, world):
    from raytracer_amd.multi_gpu import stripe as _stripe
    return _stripe(tape, rank, world)


def host_threads():
    """Threads for the CPU baseline: every core this process may run on (its affinity mask, which
    on the GPU box is the 16-CPU share of one GPU; OMP_NUM_THREADS is set to the same there)."""
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    asked = int(os.environ.get('OMP_NUM_THREADS', 0))
    return max(1, min(allowed, asked) if asked else allowed)


def cpu_baseline(renderer, tape, target_seconds, n_timed, first_pass=0):
    """Time the CPU oracle ("port") on a bounded sample of the same workload and count the
    algorithmic FP32 operations per sample with its instrumented build: over the `n_timed`
    passes the GPU was timed on, on a pixel subsample (tape mode: a pixel's work depends on its
    camera ray and the shared tape only) or, with per-pixel random numbers, on whole frames of as
    many of those passes as FLOP_COUNT_SAMPLES allows."""
    from oracle import oracle
    osc = oracle.OracleScene(renderer)
    threads = host_threads()
    one = tuple(x[:1] if x is not None else None for x in tape)
    timed = tuple(x[:n_timed] if x is not None else None for x in tape)
    if renderer.rng_mode == 'tape':
        stride = int(max(1, -(-renderer.n_pixels * n_timed // FLOP_COUNT_SAMPLES))) | 1
        while math.gcd(stride, 32) != 1 or (renderer.n_pixels // 32) % stride == 0 and stride > 1:
            stride += 2
        pixels = np.arange(0, renderer.n_pixels, stride)
        _, _, flops = osc.render(timed, n_threads=threads, count_flops=True, pixels=pixels)
        counted = len(pixels) * n_timed
        counted_text = 'all %d timed passes, every %d-th pixel' % (n_timed, stride)
    else:
        n_count = int(max(1, min(n_timed, FLOP_COUNT_SAMPLES // renderer.n_pixels)))
        some = tuple(x[:n_count] if x is not None else None for x in tape)
        _, _, flops = osc.render(some, n_threads=threads, count_flops=True, first_pass=first_pass)
        counted = renderer.n_pixels * n_count
        counted_text = 'the first %d timed passes, every pixel' % n_count
    flops_per_sample = flops / counted
    mesh_bytes_per_sample = osc.last_mesh_bytes / counted
    t0 = time.time()
    osc.render(one, n_threads=threads)
    t_one = time.time() - t0
    n = int(max(1, min(tape[0].shape[0], target_seconds / max(t_one, 1e-3))))
    sample = tuple(x[:n] if x is not None else None for x in tape)
    t0 = time.time()
    osc.render(sample, n_threads=threads)
    dt = time.time() - t0
    # the same port in the reference's execution structure (one sweep over the live
    # rays per object and step with a barrier behind each, culler, compaction; ray state in arrays)
    structure = None
    if renderer.rng_mode == 'tape':
        n_w = int(max(1, min(n, 0.4 * target_seconds / max(t_one, 1e-3))))
        sample_w = tuple(x[:n_w] if x is not None else None for x in tape)
        t0 = time.time()
        _, sweeps = osc.render_wavefront(sample_w, n_threads=threads)
        dt_w = time.time() - t0
        structure = {'value': renderer.n_pixels * n_w / dt_w / 1e6, 'unit': 'Msamples/s',
                     'sweeps_per_pass': sweeps / n_w, 'culler_group': 1,
                     'sample': '%d sample passes (%.1f s)' % (n_w, dt_w)}
    return {
        'reference_structure': structure,
        'value': renderer.n_pixels * n / dt / 1e6, 'unit': 'Msamples/s', 'cores': threads,
        'kind': 'port', 'host_cpus': os.cpu_count(),
        'cpus_allowed': len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None,
        'sample': '%d sample passes of the same %dx%d frame, same tape (%.1f s of CPU work on %d threads = '
                  'every CPU this process may use)'
                  % (n, renderer.img_shape[1], renderer.img_shape[0], dt, threads),
    }, flops_per_sample, mesh_bytes_per_sample, counted_text


def main():
    args = parse_args()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run "
                         "--nproc-per-node %d" % (args.gpus, world, args.gpus))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the HIP back end has no CPU fallback")
    # CLRAY_FORCE_DIST=1 exercises the RCCL code path with a single rank (development check)
    use_dist = world > 1 or bool(os.environ.get('CLRAY_FORCE_DIST'))
    if use_dist:
        from raytracer_amd.multi_gpu import start_process_group
        dist, local_rank = start_process_group(local_rank)      # (rehearsals may put several workers on GPU 0)
    else:
        torch.cuda.set_device(local_rank)

    from raytracer_amd.renderer import Renderer
    width, height = (int(v) for v in args.size.lower().split('x'))

    mesh_cache = {}

    def load_scene(layout='flat'):
        path = os.path.join(ROOT, 'raytracer_amd', 'scenes', args.scene + '.py')
        if args.scene == 'scene-mesh':
            n_u, n_v = (int(v) for v in args.mesh_grid.lower().split('x'))
            tree = {}
            if args.mesh_depth is not None:
                tree['max_depth'] = args.mesh_depth
            if args.mesh_leaf is not None:
                tree['max_faces_per_leaf'] = args.mesh_leaf
            if args.mesh_pooled is not None:
                tree['pooled'] = bool(args.mesh_pooled)
            if args.mesh_ropes is not None and layout == 'flat':
                tree['ropes'] = bool(args.mesh_ropes)
            if args.mesh_two_level is not None and layout == 'flat':
                tree['two_level'] = bool(args.mesh_two_level)
            if 'mesh' not in mesh_cache:
                # BASELINE config 4, "~1M tris via mesh_formats.py" (SURVEY section 8d): the generated mesh
                # is written as an OFF file and read back through the native reader (csrc/mesh_io.cpp)
                import tempfile
                from raytracer_amd.scenes.library import torus_knot_mesh, mesh_through_off_file
                with tempfile.TemporaryDirectory() as tmp:
                    v, f, seconds = mesh_through_off_file(*torus_knot_mesh(n_u, n_v), os.path.join(tmp, 'knot.off'))
                mesh_cache.update(mesh=(v, f), parse_seconds=seconds)
            scene = runpy.run_path(path, run_name='lib')['make_scene'](n_u, n_v, layout=layout, mesh=mesh_cache['mesh'],
                                                                       overlap=args.mesh_overlap, **tree)
        elif args.scene == 'scene-marchers' and args.marchers is not None:
            scene = runpy.run_path(path, run_name='lib')['make_scene'](only=args.marchers.split(','))
        else:
            scene = runpy.run_path(path)['scene']
        scene.image.size = (width, height)
        scene.samples_per_pixel = max(TAPE_SAMPLES_PER_PIXEL, (args.steps + args.warmup) * args.passes_per_step * world)
        return scene
    scene = load_scene()

    from raytracer_amd.multi_gpu import Shard
    shard = Shard(args.partition, rank, world, width * height)
    tiles = shard.tiles
    hint = int(args.occupancy_hint) if args.occupancy_hint.isdigit() else None
    r = Renderer(scene, seed=args.seed, device_index=local_rank,
                 min_waves_per_simd=hint, rng_mode=args.rng,
                 build_options=args.build_options.split() if args.build_options else None,
                 bake_parameters=None if args.bake is None else bool(args.bake),
                 split_passes=None if args.split_passes is None else bool(args.split_passes),
                 overlap_parts=args.streams if args.streams == 'auto' else int(args.streams),
                 steps_in_flight=args.in_flight if args.in_flight == 'auto' else int(args.in_flight),
                 sorted_paths=None if args.sorted_paths is None else bool(args.sorted_paths),
                 outline=None if args.outline is None else {'0': False, '1': True}.get(args.outline, args.outline),
                 **shard.renderer_arguments())
    if args.pass_groups:
        r.min_pass_groups = args.pass_groups
    shard.configure(r)                          # per-pixel random numbers are keyed by the GLOBAL pass index
    if args.regenerate is not None:
        r.regenerate = bool(args.regenerate)
    if args.occupancy_hint == 'auto':
        r.tune_occupancy(passes=args.passes_per_step)      # part of the program build, outside the timed region
        if rank == 0:
            print('occupancy tuning (ms per hint):', {str(k): round(v, 3) for k, v in r.occupancy_timings.items()},
                  'spilled vector registers:', r.occupancy_scratch,
                  'launch chains (ms per step):', getattr(r, 'chain_timings', None),
                  'steps in flight (ms per step):', getattr(r, 'flight_timings', None), file=sys.stderr)
    frame = torch.zeros((shard.frame_rows, 4), dtype=torch.float32, device='cuda')
    img = shard.my_rows(frame)[:r.n_pixels]
    r.use_image_buffer(img.data_ptr())
    host_frame = torch.empty((width * height, 4), dtype=torch.float32).pin_memory() if rank == 0 else None

    # One global sequence of passes, drawn from the same seed on every rank: the timed passes first,
    # the warm-up passes after them, so that the timed work does not depend on --warmup.  With the
    # passes partition rank g renders passes g, g + N, g + 2N, ... of it.
    P = args.passes_per_step
    n_timed = args.steps * P
    total = (args.steps + args.warmup) * P * (1 if tiles else world)
    tape = shard.local_tape(r.make_tape(0, total))
    r.upload_tape(tape, first_pass=shard.pass_offset)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    for w in range(args.warmup):
        r.render_resident(n_timed + w * P, P)
    if use_dist:
        warm = frame.clone()
        shard.combine(warm, dist)      # set up the communicator outside the timed region
        if world == 1:
            dist.all_reduce(warm)
    frame.zero_()
    sync()
    t0 = time.perf_counter()
    # the launches go out back to back; one HIP event pair on their stream brackets all of them
    r.acc.timer_begin()
    for k in range(args.steps):
        r.render_resident(k * P, P, timed=False)
    r.join_parts()                     # the default stream waits for the slices' streams (it would anyway)
    kernel_ms_total = r.acc.timer_end(wait=False)
    shard.combine(frame, dist)         # passes: all-reduce of the private sums; tiles: all-gather
    sync()
    elapsed = time.perf_counter() - t0
    # The boundary works on device buffers (include/clray_hip.h): the image is in HBM when the timed
    # region ends.  Its way to host memory over PCIe, once per run, is timed apart and reported as
    # the rate that includes it (never as `value`).
    t_copy = time.perf_counter()
    if rank == 0:
        host_frame.copy_(frame[:width * height], non_blocking=True)     # the image goes back to the host
    torch.cuda.synchronize()
    copy_seconds = time.perf_counter() - t_copy
    kernel_ms = [kernel_ms_total() / args.steps]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Roofline leg, outside the timed region: whole-frame launches, one at a time, each between a HIP
    # event pair of its own on the stream it is launched on; for a pass-split program the time of
    # clr_render alone (clr_last_split_ms).  This is the duration rocprofv3's kernel trace reports.
    launches_per_step, steps_in_flight = len(r.part_bounds()) - 1, (r.steps_in_flight if r.split_passes else 1)
    lone_ms, n_lone = None, 0
    if rank == 0 and world == 1:
        r.overlap_parts, r.steps_in_flight = 1, 1
        r.acc.finish()
        times = []
        for k in range(min(args.steps, ROOFLINE_LAUNCHES)):
            ms = r.render_resident(k * P, P)
            times.append(r.acc.last_split_ms()[0] if r.split_passes else ms)
        lone_ms, n_lone = float(np.mean(times)), len(times)

    samples = float(width * height) * P * args.steps * (1 if tiles else world)
    result = {
        'metric': 'Msamples/sec (px*spp/s) on %s' % args.scene, 'value': samples / elapsed / 1e6,
        'unit': 'Msamples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'strong' if tiles else 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s %dx%d, %d sample passes per step (%s, next-event light, '
                               'max_bounces %d), %s, build options %s'
                               % (args.scene, width, height, P, 'RGB' if r.shader.rgb else 'spectral', scene.max_bounces,
                                  'broadcast random tape' if args.rng == 'tape' else
                                  'per-pixel Philox random numbers' + (', path regeneration' if r.regenerate else '')
                                  + (', paths sorted per block' if r.sorted_paths else ''),
                                  ' '.join(r.build_options)),
                   'passes_per_step': P, 'pixels': r.n_pixels,
                   'speculative_exact_arithmetic': bool(r.speculate),
                   'occupancy_hint': r.min_waves_per_simd,
                   'launches_per_step': launches_per_step,
                   'steps_in_flight': steps_in_flight,
                   'kernel_registers': {key.lstrip('.'): value for key, value in getattr(r, 'kernel_metadata', {}).items()
                                        if key in ('.vgpr_count', '.sgpr_count', '.vgpr_spill_count',
                                                   '.sgpr_spill_count', '.private_segment_fixed_size',
                                                   '.group_segment_fixed_size')},
                   'pass_groups': r._split_args(args.passes_per_step)['pass_groups'] if r.split_passes else 1,
                   'parallelism': ('%s x%d, one all-reduce' % ('tile-partitioned' if tiles else 'pass-striped', world))
                                  if world > 1 else 'single GPU'},
    }
    if mesh_cache:
        result['config']['mesh_input'] = ('%d faces written as OFF text and parsed by the native reader in %.2f s '
                                          '(outside the timed region)' % (len(mesh_cache['mesh'][1]), mesh_cache['parse_seconds']))
    result['config']['timed_samples_per_pixel'] = n_timed * (1 if tiles else world)
    result['config']['timed_region'] = '%d launches back to back and the exchange step' % args.steps
    result['config']['image_to_host'] = {
        'ms': copy_seconds * 1e3, 'bytes': int(width * height * 16),
        'note': 'copy of the image to pinned host memory over PCIe, once per run, after the timed region',
        'value_including_it': samples / (elapsed + copy_seconds) / 1e6}
    if rank == 0:
        final = host_frame.numpy()
        assert np.isfinite(final).all() and final[:, :3].mean() > 0, "bad image"
        result['image_mean'] = float(final[:, :3].mean() / (P * args.steps * (1 if tiles else world)))
        flops_per_sample = mesh_bytes = None
        if world == 1 and not args.no_cpu_baseline:
            oracle_renderer = r
            if args.scene == 'scene-mesh':
                # the oracle walks the reference's octree stream: same tree, other serialisation
                oracle_renderer = Renderer(load_scene('stream'), seed=args.seed, device=False, rng_mode=args.rng)
            base, flops_per_sample, mesh_bytes, counted = cpu_baseline(oracle_renderer, tape, args.cpu_seconds, n_timed)
            result['cpu_baseline'] = base
        launch_samples = float(r.n_pixels) * P
        # counters of this workload's shipped kernel: rocprofv3 PMC passes over this same command
        # (tools/profile.sh + tools/summarize_profile.py; counters cannot be read from inside the
        # run), keyed by scene, launch shape and random number mode
        pmc = {}
        pmc_path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path)).get('%s %dx%d p%d %s' % (args.scene, width, height, P, args.rng)) or {}
        traffic = pmc.get('bytes_per_launch')
        lone_s = lone_ms * 1e-3 if lone_ms else None
        if lone_s is None:
            roof = {'bound': 'hbm' if args.scene == 'scene-mesh' else 'valu', 'kernel': 'clr_render',
                    'note': 'the roofline leg runs at N = 1 only'}
        elif mesh_bytes and args.scene == 'scene-mesh':
            # mesh traversal (SURVEY 8d names HBM as its roofline): `achieved` is what actually
            # crosses the HBM interface (FETCH_SIZE + WRITE_SIZE) per second of a lone launch.  The
            # algorithmic bytes of the flattened layout (8 B per node record visited, 36 B per
            # triangle test) are mostly served by L2 / Infinity Cache, so their
            # rate is reported beside it and is not an HBM fraction.
            achieved = traffic / lone_s / 1e9 if traffic else None
            algorithmic = mesh_bytes * launch_samples
            roof = {'bound': 'hbm', 'kernel': 'clr_render', 'unit': 'GB/s', 'peak': PEAK_HBM_GBS,
                    'achieved': achieved, 'frac': achieved / PEAK_HBM_GBS if achieved else None,
                    'achieved_is': 'HBM bytes per launch by the FETCH_SIZE / WRITE_SIZE counters / duration of a lone launch',
                    'algorithmic': {'bytes_per_sample': mesh_bytes, 'bytes_per_launch': algorithmic,
                                    'rate_gbs': algorithmic / lone_s / 1e9,
                                    'served_by_l2_and_infinity_cache': 1.0 - traffic / algorithmic if traffic else None},
                    'flops_per_sample': flops_per_sample, 'counted_over': counted}
        else:
            roof = {'bound': 'valu', 'kernel': 'clr_render', 'unit': 'TFLOP/s', 'peak': PEAK_FP32_VECTOR_TFLOPS}
            if flops_per_sample is not None:
                achieved = flops_per_sample * launch_samples / lone_s / 1e12
                whole_job = flops_per_sample * samples / elapsed / 1e12
                roof.update({'achieved': achieved, 'frac': achieved / PEAK_FP32_VECTOR_TFLOPS,
                             'flops_per_sample': flops_per_sample, 'counted_over': counted,
                             # the shipped launch pattern (overlapping launch chains), by the wall clock
                             # of the timed region: work of all steps / elapsed
                             'whole_job': {'achieved': whole_job, 'frac': whole_job / PEAK_FP32_VECTOR_TFLOPS,
                                           'ms_per_step': elapsed / args.steps * 1e3,
                                           'event_ms_per_step': float(np.mean(kernel_ms))}})
        # `achieved` divides by this: the average device time of clr_render in whole-frame launches
        # that run one at a time (a HIP event pair around each, on the stream it is launched on),
        # made after the timed region.  profiles/<tag>_kernel_stats.csv is rocprofv3 --kernel-trace
        # --stats of this command with --streams 1 --in-flight 1, where every launch is such a launch.
        roof['kernel_ms_avg'] = lone_ms
        roof['kernel_ms_of'] = '%d lone whole-frame launches of %d passes after the timed region' % (n_lone, P)
        roof['traffic'] = traffic
        if pmc:
            roof['traffic_source'] = pmc.get('source')
            issue, lanes = pmc.get('valu_issue_utilisation'), pmc.get('valu_lane_utilisation')
            if issue and lanes:
                # what the hardware executed (not the reference algorithm's operation count): share of
                # the VALU issue slots used x share of the 64 lanes active in them = executed lane
                # operations / the 78.6 T lane operations per second of the chip
                roof['executed'] = {'valu_issue_utilisation': issue, 'valu_lane_utilisation': lanes,
                                    'lane_ops_frac_of_peak': issue * lanes,
                                    'valu_insts_per_sample': pmc.get('valu_insts_per_sample'),
                                    'salu_insts_per_sample': pmc.get('salu_insts_per_sample'),
                                    'counters_avg_ms': pmc.get('avg_ms')}
        result['roofline'] = roof
        print(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
