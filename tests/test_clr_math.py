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


def test_log(lib):
    xs = np.float32(np.concatenate([np.linspace(1e-6, 1.0, 10001), np.logspace(-37, 38, 10001)]))
    assert ulp_error(sweep(lib.orc_math_log, xs), np.log(xs.astype(np.float64)))[xs != 1.0].max() < 1.2
    assert lib.orc_math_log(1.0) == 0.0
    assert np.isinf(lib.orc_math_log(0.0)) and np.isnan(lib.orc_math_log(-1.0))


def test_tan(lib):
    xs = np.float32(np.linspace(0.001, 1.5707, 20001))      # pi/2 * (1 - blur), blur in [0, 1)
    assert ulp_error(sweep(lib.orc_math_tan, xs), np.tan(xs.astype(np.float64))).max() < 3.5
    assert abs(lib.orc_math_tan(np.float32(np.pi / 2))) > 1e7
