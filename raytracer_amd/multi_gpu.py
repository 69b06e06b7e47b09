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
  tiles   rank g renders every pass of its tile of the frame's rays; the tiles are disjoint, so
          they are gathered, not summed, and the image equals the single-GPU image bit for bit (a
          pixel's additions all happen on one GPU, in order).  A tile is every N-th band of
          `band_blocks` x 256 consecutive rays, starting with band g (interleaved: a ray-marched
          object in the middle of the frame is shared by all ranks instead of landing on one;
          tools/tile_balance.py has the measured imbalance of both forms), or with band_blocks = 0
          the contiguous range [g * per_rank, (g + 1) * per_rank).  In the reference's ray order
          (renderer.py:435-456) a band of 4 blocks is a strip 4 pixels wide and 256 pixels high.
          The gather is one all_gather_into_tensor of equally long (padded) tiles followed by one
          index_select that puts the bands back into ray order.

bench.py and clray.py --gpus use this module; tests/test_distributed.py runs it on gloo.
"""
import os
import subprocess
import sys

import numpy as np

TILE_GRANULE = 256          # rays per thread block of the render kernel
BAND_BLOCKS = 4             # blocks per band of the interleaved tile partition (default)


class Shard:
    """What one rank renders, and how the ranks' results are combined."""

    def __init__(self, partition, rank, world, n_pixels, band_blocks=BAND_BLOCKS):
        if partition not in ('passes', 'tiles'):
            raise ValueError("partition must be 'passes' or 'tiles'")
        if not 0 <= rank < world:
            raise ValueError("rank outside the world")
        self.partition, self.rank, self.world, self.n_pixels = partition, rank, world, n_pixels
        self.tiles = partition == 'tiles' and world > 1
        self.ray_bands = None
        self._unshuffle = None
        if self.tiles and band_blocks:
            # interleaved: bands g, g + N, ... of `band` rays; fewer blocks per band on small frames
            # so that every rank gets one
            band_blocks = max(1, min(int(band_blocks), -(-n_pixels // TILE_GRANULE) // world))
            self.band = band = band_blocks * TILE_GRANULE
            n_bands = -(-n_pixels // band)
            if rank >= n_bands:
                raise ValueError("tile partition: %d GPUs for %d rays leave rank %d without a block of %d rays"
                                 % (world, n_pixels, rank, TILE_GRANULE))
            self.per_rank = -(-n_bands // world) * band      # the longest tile, the gather's chunk
            self.ray_bands = (rank, world, band)
            self.ray_range = None
            self.frame_rows = self.per_rank * world
        elif self.tiles:
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
        if self.ray_bands is not None:
            return {'ray_bands': self.ray_bands}
        return {'ray_range': self.ray_range} if self.tiles else {}

    def gather_index(self):
        """Interleaved tiles: for every ray of the frame, its row in the gathered tensor (rank-major,
        per_rank rows per rank)."""
        ray = np.arange(self.n_pixels, dtype=np.int64)
        band_no = ray // self.band
        return (band_no % self.world) * self.per_rank + (band_no // self.world) * self.band + ray % self.band

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
        if frame.is_cuda and dist.get_backend() != 'nccl':
            # rehearsal without RCCL (start_process_group): the same exchange through host memory
            host = frame.cpu()
            self._unshuffle = None
            self.combine(host, dist)
            self._unshuffle = None
            frame.copy_(host)
            return frame
        if self.ray_bands is not None:
            import torch
            gathered = torch.empty_like(frame)
            dist.all_gather_into_tensor(gathered, self.my_rows(frame).contiguous())
            if self._unshuffle is None or self._unshuffle.device != frame.device:
                self._unshuffle = torch.from_numpy(self.gather_index()).to(frame.device)
            frame[:self.n_pixels] = gathered.index_select(0, self._unshuffle)
        elif self.tiles:
            dist.all_gather_into_tensor(frame, self.my_rows(frame).clone())
        else:
            dist.all_reduce(frame)
        return frame


def start_process_group(local_rank):
    """torch.distributed for this worker: RCCL ("nccl") with the worker's own GPU.  Rehearsal on a box
    with ONE GPU (development, tests/test_gpu_cli.py): CLRAY_DIST_BACKEND=gloo lets several workers
    share GPU 0 (CLRAY_ONE_GPU=1); the exchange step then goes through host memory (Shard.combine).
    Returns (torch.distributed, device index of this worker)."""
    # RCCL shares device memory between the workers through dmabuf handles; the legacy IPC mode of
    # the HSA runtime fails there (hipIpcGetMemHandle: invalid argument).  Read when the runtime
    # starts, i.e. at this worker's first HIP call below; an explicit setting is left alone.
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    import torch.distributed as dist
    backend = os.environ.get('CLRAY_DIST_BACKEND', 'nccl')
    device = 0 if os.environ.get('CLRAY_ONE_GPU') else local_rank
    torch.cuda.set_device(device)
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', device))
    else:
        dist.init_process_group(backend)
    return dist, device


def common_seed(seed, dist, device=None):
    """The seed every rank uses: `seed` if given, else one that rank 0 draws and broadcasts (the
    reference's unseeded global generator, /root/reference/renderer.py:84-85, for a render whose
    ranks must all draw the same global sequence of passes).  31 bits: numpy's RandomState takes it."""
    import torch
    if dist.get_backend() != 'nccl':
        device = None
    value = torch.zeros(1, dtype=torch.int64, device=device)
    if dist.get_rank() == 0:
        value[0] = int(seed) if seed is not None else int.from_bytes(os.urandom(4), 'little') & 0x7fffffff
    dist.broadcast(value, src=0)
    return int(value.item())


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
