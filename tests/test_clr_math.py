"""Accuracy of the platform-independent exp / log / tan (hip_templates/clr_math.inl), measured
on the CPU build that the oracle links; the device compiles the same text and the GPU parity
tests show both sides agree bit for bit."""
import ctypes

import numpy as np
import pytest


@pytest.fixture(scope='module')
def lib():
    from oracle import oracle
    lib = oracle.load()
    for name in ('orc_math_exp', 'orc_math_log', 'orc_math_tan'):
        getattr(lib, name).restype = ctypes.c_float
        getattr(lib, name).argtypes = [ctypes.c_float]
    return lib


def ulp_error(got, exact):
    exact = np.asarray(exact, np.float64)
    ulp = 2.0 ** (np.floor(np.log2(np.abs(exact))) - 23)
    return np.abs(np.asarray(got, np.float64) - exact) / ulp


def sweep(fn, xs):
    return np.array([fn(float(x)) for x in xs], dtype=np.float32)


def test_exp(lib):
    xs = np.float32(np.linspace(-87.0, 88.0, 20001))
    assert ulp_error(sweep(lib.orc_math_exp, xs), np.exp(xs.astype(np.float64))).max() < 1.5
    assert lib.orc_math_exp(0.0) == 1.0 and lib.orc_math_exp(-0.0) == 1.0     # exp(-0*d) must be 1
    assert lib.orc_math_exp(-1000.0) == 0.0 and np.isinf(lib.orc_math_exp(100.0))
    # the special cases behind the one range test, and both of its ends
    assert np.isnan(lib.orc_math_exp(float('nan'))) and lib.orc_math_exp(float('-inf')) == 0.0
    assert lib.orc_math_exp(float('inf')) == float('inf')
    assert lib.orc_math_exp(np.float32(88.7228394)) > 3.0e38 and np.isinf(lib.orc_math_exp(np.float32(88.72284)))
    assert 0.0 < lib.orc_math_exp(np.float32(-103.0)) < 3e-45 and lib.orc_math_exp(np.float32(-103.9721)) == 0.0


def test_log(lib):
    xs = np.float32(np.concatenate([np.linspace(1e-6, 1.0, 10001), np.logspace(-37, 38, 10001)]))
    assert ulp_error(sweep(lib.orc_math_log, xs), np.log(xs.astype(np.float64)))[xs != 1.0].max() < 1.2
    assert lib.orc_math_log(1.0) == 0.0
    assert np.isinf(lib.orc_math_log(0.0)) and np.isnan(lib.orc_math_log(-1.0))
    # the special cases behind the one range test: zeros, negative numbers, infinities, NaN, subnormals
    assert lib.orc_math_log(0.0) < 0 and lib.orc_math_log(-0.0) < 0 and np.isinf(lib.orc_math_log(-0.0))
    assert np.isnan(lib.orc_math_log(float('nan'))) and np.isnan(lib.orc_math_log(float('-inf')))
    assert lib.orc_math_log(float('inf')) == float('inf')
    tiny = np.float32([1e-45, 3e-42, 1.1e-38, 1.17549421e-38, 1.17549435e-38, 3.4028235e38])
    assert ulp_error(sweep(lib.orc_math_log, tiny), np.log(tiny.astype(np.float64))).max() < 1.2


def test_tan(lib):
    xs = np.float32(np.linspace(0.001, 1.5707, 20001))      # pi/2 * (1 - blur), blur in [0, 1)
    assert ulp_error(sweep(lib.orc_math_tan, xs), np.tan(xs.astype(np.float64))).max() < 3.5
    assert abs(lib.orc_math_tan(np.float32(np.pi / 2))) > 1e7


def test_sincos_2pi(lib):
    lib.orc_math_sincos_2pi.argtypes = [ctypes.c_float, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    us = np.float32(np.concatenate([np.linspace(0, 1, 20001)[:-1], [0.25, 0.5, 0.75, 0.99999994]]))
    s, c = ctypes.c_float(), ctypes.c_float()
    got = []
    for u in us:
        lib.orc_math_sincos_2pi(float(u), ctypes.byref(s), ctypes.byref(c))
        got.append((s.value, c.value))
    got = np.array(got)
    angle = 2 * np.pi * us.astype(np.float64)
    assert np.abs(got[:, 0] - np.sin(angle)).max() < 3e-7
    assert np.abs(got[:, 1] - np.cos(angle)).max() < 3e-7


def test_philox_known_answers_and_sampler_statistics(lib):
    """Philox-4x32-10 against the published known-answer vectors (Random123 kat_vectors), and
    the moments of the per-segment samplers of hip_templates/clr_rng.inl."""
    u4 = ctypes.c_uint * 4
    u2 = ctypes.c_uint * 2
    out = u4()
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for counter, key, want in kat:
        lib.orc_philox4x32(u4(*counter), u2(*key), out)
        assert tuple(out) == want
    lib.orc_rng_segment.argtypes = [ctypes.c_uint] * 5 + [ctypes.POINTER(ctypes.c_float)]
    buf = (ctypes.c_float * 15)()
    draws = []
    for pixel in range(4000):
        lib.orc_rng_segment(7, 9, pixel, pixel % 5, pixel % 3, buf)
        draws.append(list(buf))
    d = np.array(draws, dtype=np.float64)      # rvec 0:3, p 3, gauss 4:7, russian 7, light_u 8, light_dir 9:12, three more uniforms 12:15
    for cols in (slice(0, 3), slice(9, 12)):
        assert np.abs(np.linalg.norm(d[:, cols], axis=1) - 1).max() < 1e-6
        assert np.abs(d[:, cols].mean(axis=0)).max() < 0.04
    assert abs(d[:, 3].mean() - 0.5) < 0.02 and abs(d[:, 7].mean() - 0.5) < 0.02 and abs(d[:, 8].mean() - 0.5) < 0.02
    assert (d[:, [3, 7, 8, 12, 13, 14]] >= 0).all() and (d[:, [3, 7, 8, 12, 13, 14]] < 1).all()
    assert np.abs(d[:, 12:15].mean(axis=0) - 0.5).max() < 0.03
    g = d[:, 4:7]
    assert np.abs(g.mean(axis=0)).max() < 0.06 and np.abs(g.std(axis=0) - 1).max() < 0.05
    assert np.abs(np.corrcoef(g.T) - np.eye(3)).max() < 0.06
