"""Multi-GPU path on CPU (gloo, world size 2): the sharding and exchange logic of
raytracer_amd/multi_gpu.py, which bench.py and clray.py --gpus run on RCCL.

The GPU render kernel cannot run here, so each rank renders its shard with the CPU oracle
(tests may use it); under test are the partitions, the global pass numbering of the per-pixel
random numbers and that the combined image equals a single-process render."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_scene
from raytracer_amd.multi_gpu import Shard, stripe


def test_stripes_partition_the_passes():
    import bench
    tape = (np.arange(10)[:, None], np.arange(10)[:, None] * 2, None)
    seen = []
    for rank in range(4):
        a, b, c = bench.stripe(tape, rank, 4)
        assert c is None and np.array_equal(b, a * 2)
        shard = Shard('passes', rank, 4, 1000)
        assert np.array_equal(shard.local_tape(tape)[0], a) and shard.local_passes(10) == len(a)
        assert (shard.pass_offset, shard.pass_stride) == (rank, 4) and shard.ray_range is None
        seen += list(a[:, 0])
    assert sorted(seen) == list(range(10))


@pytest.mark.parametrize('n_pixels,world', [(2073600, 8), (2073600, 3), (76800, 2), (1024, 4), (3840 * 2160, 8)])
def test_contiguous_tiles_partition_the_rays(n_pixels, world):
    covered = np.zeros(n_pixels, int)
    for rank in range(world):
        shard = Shard('tiles', rank, world, n_pixels, band_blocks=0)
        assert shard.renderer_arguments() == {'ray_range': shard.ray_range}
        first, count = shard.ray_range
        assert first % 256 == 0 and count > 0 and shard.frame_rows >= n_pixels
        assert (shard.pass_offset, shard.pass_stride) == (0, 1)
        covered[first:first + count] += 1
        rows = shard.my_rows(np.zeros((shard.frame_rows, 4)))
        assert rows.shape[0] == shard.per_rank >= count
    assert (covered == 1).all()


@pytest.mark.parametrize('n_pixels,world,band_blocks', [(2073600, 8, 4), (2073600, 3, 4), (76800, 2, 1), (1024, 4, 4),
                                                        (3840 * 2160, 8, 8), (700, 2, 4), (30 * 22, 2, 4)])
def test_interleaved_tiles_partition_the_rays_and_find_their_way_back(n_pixels, world, band_blocks):
    """rank g owns every N-th band of the ray order.  The Renderer of a rank walks
    exactly those rays, in order; a gather of the padded tiles followed by gather_index() is the frame."""
    from raytracer_amd.renderer import Renderer
    covered = np.zeros(n_pixels, int)
    gathered = None
    for rank in range(world):
        shard = Shard('tiles', rank, world, n_pixels, band_blocks=band_blocks)
        assert shard.ray_range is None and shard.band % 256 == 0 and shard.band <= band_blocks * 256
        assert (shard.pass_offset, shard.pass_stride) == (0, 1)
        g, n, band = shard.renderer_arguments()['ray_bands']
        assert (g, n, band) == (rank, world, shard.band)
        rays = np.concatenate([np.arange(b, min(b + band, n_pixels)) for b in range(g * band, n_pixels, n * band)])
        covered[rays] += 1
        if gathered is None:
            gathered = np.full(shard.frame_rows, -1, dtype=np.int64)
        rows = shard.my_rows(gathered)
        assert rows.shape[0] == shard.per_rank >= len(rays) and shard.frame_rows >= n_pixels
        rows[:len(rays)] = rays                      # every rank "renders" the index of its rays
    assert (covered == 1).all()
    assert np.array_equal(gathered[shard.gather_index()], np.arange(n_pixels))
    # the renderer's view of one such tile: the same rays, the same kernel arguments
    if n_pixels == 30 * 22:
        shard = Shard('tiles', 1, world, n_pixels, band_blocks=band_blocks)
        r = Renderer(load_scene('scene-dev', (30, 22)), device=False, seed=0, rng_mode='philox',
                     **shard.renderer_arguments())
        assert np.array_equal(r.ray_index, np.arange(256, 512)) and r.n_pixels == 256
        args = r._rng_args(7)
        assert (args['ray_band'], args['ray_band_stride'], args['ray_band_base'], args['first_ray']) == (256, 512, 256, 0)
        assert np.array_equal(r.image_order, r.full_image_order[256:512])


def test_tiles_refuse_more_ranks_than_blocks():
    with pytest.raises(ValueError):
        Shard('tiles', 3, 4, 512)
    with pytest.raises(ValueError):
        Shard('tiles', 3, 4, 512, band_blocks=0)
    assert Shard('tiles', 0, 1, 512).ray_range is None          # one GPU: nothing to partition


def _worker(rank, world, port, out_path, partition, rng_mode):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from raytracer_amd.multi_gpu import common_seed
    assert common_seed(3 if rank == 0 else 99, dist) == 3          # rank 0's seed is everybody's
    drawn = common_seed(None, dist)                                 # unseeded: rank 0 draws, all agree
    both = [None, None]
    dist.all_gather_object(both, drawn)
    assert both[0] == both[1] and 0 <= drawn < 2 ** 31
    from raytracer_amd.renderer import Renderer
    from oracle import oracle
    size, n_passes = (64, 32), 8
    shard = Shard(partition.split('-')[0], rank, world, size[0] * size[1],
                  **({'band_blocks': 0} if partition.endswith('-contiguous') else {'band_blocks': 1}))
    partition = partition.split('-')[0]
    r = Renderer(load_scene('scene-dev', size), device=False, seed=3, rng_mode=rng_mode,
                 **shard.renderer_arguments())
    shard.configure(r)
    tape = r.make_tape(0, n_passes)                # the same global sequence of passes on every rank
    mine = shard.local_tape(tape)
    img, _, _ = oracle.OracleScene(r).render(mine, n_threads=2, first_pass=shard.pass_offset)
    frame = torch.zeros((shard.frame_rows, 4))
    shard.my_rows(frame)[:r.n_pixels] = torch.from_numpy(img)
    shard.combine(frame, dist)
    if rank == 0:
        single = Renderer(load_scene('scene-dev', size), device=False, seed=3, rng_mode=rng_mode)
        osc = oracle.OracleScene(single)
        full = single.make_tape(0, n_passes)          # same seed: the same tape as every rank drew
        whole, _, _ = osc.render(full, n_threads=2)
        # the passes this rank rendered, one by one, as a single process renders them
        mine_single = np.zeros_like(whole)
        if partition == 'passes':
            for j in range(shard.pass_offset, n_passes, shard.pass_stride):
                one = tuple(x[j:j + 1] if x is not None else None for x in full)
                mine_single, _, _ = osc.render(one, img=mine_single, n_threads=2, first_pass=j)
        np.savez(out_path, combined=frame.numpy()[:size[0] * size[1]], whole=whole, mine=img, mine_single=mine_single)
    dist.destroy_process_group()


@pytest.mark.parametrize('partition,rng_mode', [('passes', 'tape'), ('passes', 'philox'),
                                                ('tiles', 'tape'), ('tiles', 'philox'),
                                                ('tiles-contiguous', 'philox')])
def test_two_ranks_equal_a_single_process(tmp_path, partition, rng_mode):
    out = str(tmp_path / 'result.npz')
    port = 29500 + (os.getpid() + hash((partition, rng_mode))) % 2000
    mp.spawn(_worker, args=(2, port, out, partition, rng_mode), nprocs=2, join=True)
    res = np.load(out)
    assert res['whole'][:, :3].mean() > 0
    if partition.startswith('tiles'):
        # disjoint tiles are gathered: every pixel's sum is made on one rank, in pass order
        assert np.array_equal(res['combined'].view(np.uint32), res['whole'].view(np.uint32))
    else:
        # every rank renders exactly the passes a single process renders (same tape entries, or
        # per-pixel random numbers keyed by the GLOBAL pass index): rank 0's partial sum is bit
        # for bit the single-process sum over passes 0, 2, 4, ...
        assert np.array_equal(res['mine'].view(np.uint32), res['mine_single'].view(np.uint32))
        # the all-reduce adds per-rank partial sums: equal up to float32 rounding of the sums
        assert np.allclose(res['combined'], res['whole'], rtol=1e-5, atol=1e-6)
        assert not np.allclose(res['mine'], res['whole'], rtol=1e-3)      # a stripe alone is not the sum
