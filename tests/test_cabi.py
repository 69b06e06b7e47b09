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
    # size word, 7 pointers, 3 ints, padding, pointer, 5 ints of the per-pixel random number mode +
    # padding, then the pass-split pair: pointer, int, the pass stride, the stream of the sums, and
    # the three ints of the band-interleaved tile + padding
    assert ctypes.sizeof(accelerator.RenderArgs) == 8 + 7 * 8 + 3 * 4 + 4 + 8 + 5 * 4 + 4 + 8 + 4 + 4 + 8 + 3 * 4 + 4
    assert accelerator.RenderArgs.ray_band.offset == 136 and accelerator.RenderArgs.ray_band_base.offset == 144
    assert accelerator.RenderArgs.reduce_after.offset == 128
    assert accelerator.RenderArgs.struct_size.offset == 0 and accelerator.RenderArgs.img.offset == 8
    assert accelerator.RenderArgs.pass_log.offset == 112 and accelerator.RenderArgs.pass_groups.offset == 120
    assert accelerator.RenderArgs.pass_stride.offset == 124
    assert accelerator.RenderArgs.debug_dump.offset == 80
    assert accelerator.RenderArgs.per_pixel_rng.offset == 88 and accelerator.RenderArgs.first_ray.offset == 104
    assert accelerator.RenderArgs().struct_size == ctypes.sizeof(accelerator.RenderArgs)
    # the ctypes binding in INTEGRATION.md lists the same fields in the same order
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    stub = text[text.index('class RenderArgs'):]
    stub = stub[:stub.index(']')]
    assert re.findall(r"\('([a-z_]+)'", stub) == [name for name, _ in accelerator.RenderArgs._fields_]
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


def test_program_store_and_child_compile(tmp_path, monkeypatch):
    """program_cache: the child-process compile gives a code object like the in-process one; the
    disk store exists only when CLRAY_CACHE_DIR asks for it and rejects altered files."""
    from raytracer_amd import program_cache
    source = 'extern "C" __global__ void k(float *p) { p[threadIdx.x] = 2.0f * p[threadIdx.x]; }\n'
    monkeypatch.delenv('CLRAY_CACHE_DIR', raising=False)
    assert program_cache.cache_dir() is None
    here = program_cache.compile_here(source, ['-O3'], 'gfx950')
    child = program_cache.compile_in_child(source, ['-O3'], 'gfx950')
    assert here[:4] in (b'\x7fELF', b'__CL') and len(child) == len(here)
    binary, how = program_cache.get_binary(source, ['-O3'], 'gfx950')
    assert how in ('here', 'child') and len(binary) == len(here)
    _, how = program_cache.get_binary(source, ['-O3'], 'gfx950')
    assert how != 'cache'                               # nothing is kept unless asked for

    monkeypatch.setenv('CLRAY_CACHE_DIR', str(tmp_path))
    binary, how = program_cache.get_binary(source, ['-O3'], 'gfx950')
    assert how != 'cache'
    again, how = program_cache.get_binary(source, ['-O3'], 'gfx950')
    assert how == 'cache' and again == binary
    _, how = program_cache.get_binary(source, ['-O2'], 'gfx950')
    assert how != 'cache'
    for f in tmp_path.iterdir():                        # a flipped byte invalidates the file
        blob = bytearray(f.read_bytes())
        blob[-1] ^= 1
        f.write_bytes(bytes(blob))
    _, how = program_cache.get_binary(source, ['-O3'], 'gfx950')
    assert how != 'cache'

    with pytest.raises(Exception) as err:
        program_cache.compile_in_child('this is not HIP', [], 'gfx950')
    assert 'error' in str(err.value).lower()


def test_code_object_metadata_is_read_without_a_device():
    """program_cache.kernel_metadata: registers and spills of each kernel from the code object's
    NT_AMDGPU_METADATA note (what the occupancy tuner uses to stay away from builds that spill)."""
    from raytracer_amd import program_cache
    source = ('extern "C" __global__ void a(float *p) { p[threadIdx.x] *= 2.0f; }\n'
              'extern "C" __global__ void __launch_bounds__(64) b(float *p, int n) '
              '{ __shared__ float s[64]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[63 - threadIdx.x] * n; }\n')
    meta = program_cache.kernel_metadata(program_cache.compile_here(source, ['-O3'], 'gfx950'))
    assert sorted(meta) == ['a', 'b']
    assert meta['b']['.group_segment_fixed_size'] == 256 and meta['b']['.max_flat_workgroup_size'] == 64
    assert meta['a']['.vgpr_spill_count'] == 0 and 0 < meta['a']['.vgpr_count'] <= 16
    assert program_cache.kernel_metadata(b'not a code object') == {}
