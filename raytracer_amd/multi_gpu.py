"""A render shared by the GPUs of one node: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI) for the one exchange step at the end.

The reference is a single-device program (/root/reference/accelerator.py:24-25); what makes the
path shard is that samples are independent and the image is a sum over passes
(/root/reference/renderer.py:24-28, the property `--append` already relies on,
/root/reference/imgutils.py:52-58).  Two partitions:

  passes  rank g renders passes g, g + N, g + 2N, ... of ONE global sequence of passes over the
          whole frame into a private accumulator; one all-reduce (sum) at the end.  Every pass is
          the pass a single GPU would have rendered: the tape is drawn once from the common seed
          and striped, per-pixel random numbers are keyed by the global pass index
          (first_pass = g, pass_stride = N).  Only the order of the final additions differs
          from a single-GPU render (per-rank partial sums), so images agree to float32 rounding
          of the sums, not bit for bit.
  tiles   rank g renders every pass of rays [g * per_rank, (g + 1) * per_rank), a set of 4-pixel
          wide column blocks of the frame (the reference's ray order, renderer.py:435-456); the
          tiles are disjoint, so they are gathered, not summed, and the image equals the
          single-GPU image bit for bit (a pixel's additions all happen on one GPU, in order).

bench.py and clray.py --gpus use this module; tests/test_distributed.py runs it on gloo.
"""
import os
import subprocess
import sys

import numpy as np

TILE_GRANULE = 256          # rays per thread block of the render kernel


class Shard:
    """What one rank renders, and how the ranks' results are combined."""

    def __init__(self, partition, rank, world, n_pixels):
        if partition not in ('passes', 'tiles'):
            raise ValueError("partition must be 'passes' or 'tiles'")
        if not 0 <= rank < world:
            raise ValueError("rank outside the world")
        self.partition, self.rank, self.world, self.n_pixels = partition, rank, world, n_pixels
        self.tiles = partition == 'tiles' and world > 1
        if self.tiles:
            # whole blocks of TILE_GRANULE rays; the last ranks may get less (or nothing)
            self.per_rank = -(-n_pixels // world // TILE_GRANULE) * TILE_GRANULE
            first = min(rank * self.per_rank, n_pixels)
            self.ray_range = (first, min(self.per_rank, n_pixels - first))
            if self.ray_range[1] <= 0:
                raise ValueError("tile partition: %d GPUs for %d rays leave rank %d without a block of %d rays"
                                 % (world, n_pixels, rank, TILE_GRANULE))
            self.frame_rows = self.per_rank * world          # room for every rank's full tile
        else:
            self.per_rank = n_pixels
            self.ray_range = None
            self.frame_rows = n_pixels
        # global index of local pass j: pass_offset + j * pass_stride
        self.pass_offset, self.pass_stride = (0, 1) if self.tiles or world == 1 else (rank, world)

    def renderer_arguments(self):
        """Keyword arguments for Renderer(...) on this rank."""
        return {'ray_range': self.ray_range} if self.tiles else {}

    def configure(self, renderer):
        renderer.pass_stride = self.pass_stride

    def local_passes(self, n_global):
        """How many of n_global passes this rank renders."""
        if self.pass_stride == 1:
            return n_global
        return len(range(self.pass_offset, n_global, self.pass_stride))

    def local_tape(self, tape):
        """This rank's passes of a tape drawn for the global sequence of passes."""
        if self.pass_stride == 1:
            return tape
        return tuple(x[self.pass_offset::self.pass_stride] if x is not None else None for x in tape)

    def my_rows(self, frame):
        """The rows of the (frame_rows x 4) frame tensor / array this rank accumulates into."""
        if not self.tiles:
            return frame
        first = self.rank * self.per_rank
        return frame[first:first + self.per_rank]

    def combine(self, frame, dist):
        """The exchange step: after it `frame[:n_pixels]` holds the whole image on every rank.
        frame: torch tensor (frame_rows x 4 float32) whose my_rows() this rank has rendered."""
        if self.world == 1:
            return frame
        if self.tiles:
            dist.all_gather_into_tensor(frame, self.my_rows(frame).clone())
        else:
            dist.all_reduce(frame)
        return frame


def stripe(tape, rank, world):
    """Passes rank, rank + world, ... of a tape (bench.py's name for Shard.local_tape)."""
    return tuple(x[rank::world] if x is not None else None for x in tape)


def launch_workers(n_gpus, script, argv, port=None):
    """Run `script argv` as n_gpus processes, one per GPU, under torch.distributed.run; must be
    called before this process touches a GPU.  Returns the exit code."""
    port = port or 29400 + os.getpid() % 2000
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), script] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.call(cmd, env=env)


def worker_identity():
    """(rank, local_rank, world) of a process started by torch.distributed.run, else (0, 0, 1)."""
    return (int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)),
            int(os.environ.get('WORLD_SIZE', 1)))
