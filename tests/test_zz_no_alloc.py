"""clr_render_passes_static_log, the entry point that allocates nothing: called through the C ABI
and compared bit for bit with clr_render_passes (which takes the addend log as a workspace
argument) on two calls in a row, the second with a smaller input, so that what the first call
left in the library's static array would show if the second one read any of it."""
import ctypes

import numpy as np
import pytest

from raytracer_amd.accelerator import ClrError, RenderArgs
from raytracer_amd.renderer import Renderer
from conftest import load_scene

pytestmark = pytest.mark.gpu

STATIC_LOG_BYTES = 128 << 20


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def both_ways(name, size, n_passes, seed):
    """The same passes through the entry point with a caller-provided log and through the one
    with the static log; returns (image, image)."""
    r = Renderer(load_scene(name, size), seed=seed, split_passes=True)
    tape = r.make_tape(0, n_passes)
    r.render_tape(tape)
    expected = r.img.get()
    r.img.fill(0)
    r.render_tape(tape, static_log=True)
    return expected, r.img.get()


def test_static_log_entry_point_equals_the_allocating_one_twice_in_a_row():
    # call 1: scene-test (all shape classes), 160 x 120 x 6 passes = 11.5 MB of log
    expected, got = both_ways('scene-test', (160, 120), 6, seed=3)
    assert expected[:, :3].mean() > 0
    assert np.array_equal(bits(got), bits(expected))
    # call 2, smaller: scene-dev 64 x 48 x 3 passes; the array still holds call 1's log
    expected, got = both_ways('scene-dev', (64, 48), 3, seed=4)
    assert expected[:, :3].mean() > 0
    assert np.array_equal(bits(got), bits(expected))
    # and per-pixel random numbers with path regeneration, smaller again
    r = Renderer(load_scene('scene-dev', (32, 24)), seed=5, split_passes=True, rng_mode='philox')
    tape = r.make_tape(0, 2)
    r.render_tape(tape)
    expected = r.img.get()
    r.img.fill(0)
    r.render_tape(tape, static_log=True)
    assert np.array_equal(bits(r.img.get()), bits(expected))


def test_static_log_entry_point_refuses_what_it_cannot_do():
    r = Renderer(load_scene('scene-dev', (64, 48)), seed=1, split_passes=True)
    floats = 1 + 6 * r.scene.max_bounces
    too_many = STATIC_LOG_BYTES // (4 * floats * r.n_pixels) + 1
    tape = r.make_tape(0, too_many)
    before = r.img.get()
    with pytest.raises(ClrError) as err:
        r.render_tape(tape, static_log=True)
    assert 'fewer passes' in str(err.value)
    assert np.array_equal(bits(r.img.get()), bits(before))
    # the largest call that fits goes through
    r.render_tape(tuple(x[:too_many - 1] if x is not None else None for x in tape), static_log=True)
    assert r.img.get()[:, :3].mean() > 0
    # a caller-provided log is refused, and so is a program that has no log
    r._split_args = lambda n, debug=False, static_log=False: dict(pass_log=r.img.ptr, pass_groups=2)
    with pytest.raises(ClrError) as err:
        r.render_tape(r.make_tape(0, 1), static_log=True)
    assert 'pass_log must be NULL' in str(err.value)
    plain = Renderer(load_scene('scene-dev', (64, 48)), seed=1, split_passes=False)
    with pytest.raises(ClrError) as err:
        plain.render_tape(plain.make_tape(0, 1), static_log=True)
    assert 'pass-split' in str(err.value)


def test_static_log_refuses_a_call_that_would_overlap_the_previous_one_on_another_stream():
    """One log per GPU: while a call is still running, a call on ANOTHER stream is refused instead
    of writing into the log the first one is reading; after a wait, and on the same stream, calls
    go through."""
    r = Renderer(load_scene('scene-test', (320, 240)), seed=2, split_passes=True)
    floats = 1 + 6 * r.scene.max_bounces
    n = min(16, STATIC_LOG_BYTES // (4 * floats * r.n_pixels))
    tape = r.make_tape(0, n)
    r.upload_tape(tape)
    other = r.acc.new_stream()
    r.render_resident(0, n, timed=False, static_log=True)         # about a millisecond of GPU work, not waited for
    r.render_resident(0, 1, timed=False, static_log=True)         # same stream: ordered, accepted
    with pytest.raises(ClrError) as err:
        r.render_resident(0, n, stream=other, timed=False, static_log=True)
    assert 'another stream' in str(err.value)
    r.acc.finish()
    before = r.img.get()
    r.render_resident(0, 1, stream=other, timed=False, static_log=True)
    r.acc.finish(other)
    assert not np.array_equal(bits(r.img.get()), bits(before))
