"""The C-ABI library loads and exports every symbol include/clray_hip.h declares; without a
GPU it refuses loudly instead of falling back to anything."""
import ctypes
import os
import re

import pytest

from raytracer_amd import accelerator
from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, 'include', 'clray_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(clr_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported():
    lib = ctypes.CDLL(accelerator.LIBRARY_PATH)
    names = declared_functions()
    assert 'clr_render_passes' in names and 'clr_program_build' in names and len(names) >= 17
    for name in names:
        assert hasattr(lib, name), name


def test_python_prototypes_cover_the_header():
    lib = accelerator.load_library()
    for name in declared_functions():
        assert getattr(lib, name).argtypes is not None, name


def test_struct_layout_matches_header():
    assert ctypes.sizeof(accelerator.RenderArgs) == 7 * 8 + 3 * 4 + 4 + 8   # 7 ptrs, 3 ints, pad, ptr
    assert accelerator.RenderArgs.debug_dump.offset == 72
    assert ctypes.sizeof(accelerator.DeviceInfo) == 128 + 64 + 4 * 4 + 8


def test_compile_errors_carry_the_hiprtc_log():
    with pytest.raises(accelerator.ClrError) as err:
        accelerator.compile_only('extern "C" __global__ void k() { this is not HIP }')
    assert 'error' in str(err.value)


def test_no_silent_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(accelerator.ClrError) as err:
        accelerator.Accelerator()
    assert 'no HIP device' in str(err.value)
