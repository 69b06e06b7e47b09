"""The C-ABI exchange step (clr_comm_*, RCCL loaded with dlopen) on the one GPU of the test box:
a communicator of one rank, where the sum is the identity and the gather a copy.  Several GPUs
in one process are exercised by the same code path with n > 1; the multi-GPU logic itself
(partitions, global pass numbering) is covered on CPU by tests/test_distributed.py."""
import numpy as np
import pytest

from raytracer_amd.accelerator import Accelerator, Communicator, ClrError

pytestmark = pytest.mark.gpu


def test_single_rank_allreduce_and_allgather_through_the_c_abi():
    acc = Accelerator(0)
    comm = Communicator([acc])
    assert comm.size() == 1
    host = np.random.RandomState(1).rand(1920 * 8, 4).astype(np.float32)
    buf = acc.to_device(host)
    comm.allreduce_sum([buf])
    assert np.array_equal(buf.get(), host)
    out = acc.new_array(host.shape, np.float32, zeros=True)
    comm.allgather([buf], [out])
    assert np.array_equal(out.get(), host)
    # a render accumulator goes through unchanged
    from raytracer_amd.renderer import Renderer
    from conftest import load_scene
    r = Renderer(load_scene('scene-dev', (64, 48)), seed=0)
    r.render_samples(0, 4)
    before = r.img.get()
    Communicator([r.acc]).allreduce_sum([r.img])
    assert np.array_equal(r.img.get(), before) and before[:, :3].mean() > 0
    comm.close()


def test_two_ranks_on_one_gpu_are_refused():
    a, b = Accelerator(0), Accelerator(0)
    with pytest.raises(ClrError) as err:
        Communicator([a, b])
    assert 'one rank per device' in str(err.value)
