"""Compiled scene programs: which compiler builds them and, if asked, where they are kept.

The reference rebuilds its OpenCL program on every start (/root/reference/accelerator.py:43-49)
and leaves caching to pyopencl.  Here the scene-specialised HIP source is compiled by hiprtc, and
two things matter for a production process:

* which hiprtc.  A process that has imported PyTorch has the wheel's own libhiprtc / comgr
  mapped, an older compiler than the system's ROCm; the render kernel it generates is about 5 %
  slower (measured on scene-dev).  In that case the source is compiled by the system's hiprtc in
  a short-lived child process (compile_worker.py: CPU only, source in and code object out through
  pipes, no files) and only the code object is loaded.
* start-up time.  OPT-IN: when the environment variable CLRAY_CACHE_DIR names a directory, code
  objects are kept there under a key made of source text, options, target and compiler, each
  file carrying the SHA-256 of its payload, which is verified before the object is used.
  Without that variable nothing is written anywhere.
"""
import ctypes
import hashlib
import os
import subprocess
import sys

from . import accelerator

_HERE = os.path.dirname(os.path.abspath(__file__))
SYSTEM_HIPRTC = '/opt/rocm/lib/libhiprtc.so'
_MAGIC = b'CLRAYCO1'


def cache_dir():
    """The directory the user named in CLRAY_CACHE_DIR (created on demand), or None."""
    path = os.environ.get('CLRAY_CACHE_DIR')
    if not path:
        return None
    try:
        os.makedirs(path, exist_ok=True)
    except OSError:
        return None
    return path if os.access(path, os.W_OK) else None


def mapped_hiprtc():
    """Path of the libhiprtc mapped into this process, or None."""
    try:
        with open('/proc/self/maps') as maps:
            for line in maps:
                if 'libhiprtc' in line:
                    return line.split()[-1]
    except OSError:
        pass
    return None


def system_compiler_id():
    """Identifies the system hiprtc (its resolved file name carries the version)."""
    if os.path.exists(SYSTEM_HIPRTC):
        return os.path.basename(os.path.realpath(SYSTEM_HIPRTC))
    return None


def foreign_compiler_mapped():
    """True if this process would compile with another hiprtc than the system's."""
    accelerator.load_library()
    mapped, system = mapped_hiprtc(), system_compiler_id()
    return bool(mapped and system and os.path.basename(os.path.realpath(mapped)) != system)


def compile_here(source, options, arch):
    lib = accelerator.load_library()
    arr, n = accelerator._options_array(options)
    binary, size = ctypes.c_void_p(), ctypes.c_size_t(0)
    accelerator._check(lib.clr_compile(source.encode('ascii'), n, arr, arch.encode(),
                                       ctypes.byref(binary), ctypes.byref(size)))
    try:
        return ctypes.string_at(binary, size.value)
    finally:
        lib.clr_binary_free(binary)


class ChildUnavailable(OSError):
    """The compile child could not do its job for a reason other than the source text."""


def compile_in_child(source, options, arch):
    """Compile with the system's hiprtc in a child process that never touches the GPU.  Raises
    ClrError with the compiler's log if the source does not compile, ChildUnavailable (an OSError)
    if the child could not run."""
    from .compile_worker import COMPILE_ERROR
    proc = subprocess.run([sys.executable, os.path.join(_HERE, 'compile_worker.py'), arch] + list(options),
                          input=source.encode('ascii'), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if proc.returncode == COMPILE_ERROR:
        raise accelerator.ClrError(proc.stderr.decode('utf-8', 'replace'))
    if proc.returncode != 0 or not proc.stdout:
        raise ChildUnavailable('compile_worker failed with exit status %d: %s'
                               % (proc.returncode, proc.stderr.decode('utf-8', 'replace')[-500:]))
    return proc.stdout


def _read_verified(path):
    """Payload of a store file, or None if it is absent, truncated or altered."""
    try:
        with open(path, 'rb') as f:
            blob = f.read()
    except OSError:
        return None
    head = len(_MAGIC) + 32
    if len(blob) <= head or blob[:len(_MAGIC)] != _MAGIC:
        return None
    payload = blob[head:]
    return payload if hashlib.sha256(payload).digest() == blob[len(_MAGIC):head] else None


def get_binary(source, options, arch):
    """Code object for (source, options, arch).  Returns (binary, how) with how in
    'cache' (verified file of the opt-in store), 'child', 'here'."""
    options = list(options)
    use_child = foreign_compiler_mapped()
    mapped = mapped_hiprtc()
    compiler = system_compiler_id() if use_child or not mapped else os.path.basename(os.path.realpath(mapped))
    store = cache_dir()
    path = None
    if store:
        key = hashlib.sha256('\0'.join([source, '\1'.join(options), arch, compiler or '?']).encode()).hexdigest()
        path = os.path.join(store, key + '.co')
        binary = _read_verified(path)
        if binary is not None:
            return binary, 'cache'
    if use_child:
        try:
            binary, how = compile_in_child(source, options, arch), 'child'
        except OSError:
            # no child processes here (some profilers and sandboxes refuse them): compile with the
            # hiprtc this process has mapped - another compiler than the key above names, so the
            # result is not kept.  (A source that does not compile raises ClrError with the log.)
            binary, how = compile_here(source, options, arch), 'here'
            path = None
    else:
        binary, how = compile_here(source, options, arch), 'here'
    if path:
        tmp = '%s.%d.tmp' % (path, os.getpid())
        try:
            with open(tmp, 'wb') as f:
                f.write(_MAGIC + hashlib.sha256(binary).digest() + binary)
            os.replace(tmp, path)
        except OSError:
            pass
    return binary, how


# ---------------------------------------------------------------- code object metadata
def _msgpack(buf, pos=0):
    """Minimal MessagePack reader (maps, arrays, strings, integers, booleans, nil, floats):
    returns (value, next position).  The AMDGPU code object metadata uses nothing else."""
    import struct
    b = buf[pos]
    pos += 1
    if b <= 0x7f:
        return b, pos
    if 0x80 <= b <= 0x8f or b in (0xde, 0xdf):
        n = b & 0x0f if b <= 0x8f else int.from_bytes(buf[pos:pos + (2 if b == 0xde else 4)], 'big')
        pos += 0 if b <= 0x8f else (2 if b == 0xde else 4)
        out = {}
        for _ in range(n):
            key, pos = _msgpack(buf, pos)
            out[key], pos = _msgpack(buf, pos)
        return out, pos
    if 0x90 <= b <= 0x9f or b in (0xdc, 0xdd):
        n = b & 0x0f if b <= 0x9f else int.from_bytes(buf[pos:pos + (2 if b == 0xdc else 4)], 'big')
        pos += 0 if b <= 0x9f else (2 if b == 0xdc else 4)
        out = []
        for _ in range(n):
            item, pos = _msgpack(buf, pos)
            out.append(item)
        return out, pos
    if 0xa0 <= b <= 0xbf or b in (0xd9, 0xda, 0xdb, 0xc4, 0xc5, 0xc6):
        width = {0xd9: 1, 0xda: 2, 0xdb: 4, 0xc4: 1, 0xc5: 2, 0xc6: 4}.get(b, 0)
        n = b & 0x1f if not width else int.from_bytes(buf[pos:pos + width], 'big')
        pos += width
        raw = bytes(buf[pos:pos + n])
        return (raw if b in (0xc4, 0xc5, 0xc6) else raw.decode('utf-8', 'replace')), pos + n
    if b == 0xc0:
        return None, pos
    if b in (0xc2, 0xc3):
        return b == 0xc3, pos
    if b in (0xcc, 0xcd, 0xce, 0xcf):
        width = 1 << (b - 0xcc)
        return int.from_bytes(buf[pos:pos + width], 'big'), pos + width
    if b in (0xd0, 0xd1, 0xd2, 0xd3):
        width = 1 << (b - 0xd0)
        return int.from_bytes(buf[pos:pos + width], 'big', signed=True), pos + width
    if b == 0xca:
        return struct.unpack('>f', buf[pos:pos + 4])[0], pos + 4
    if b == 0xcb:
        return struct.unpack('>d', buf[pos:pos + 8])[0], pos + 8
    if b >= 0xe0:
        return b - 0x100, pos
    raise ValueError('unsupported MessagePack type 0x%02x' % b)


def kernel_metadata(binary):
    """{kernel name: metadata map} from the NT_AMDGPU_METADATA note of a gfx9 code object (ELF64,
    possibly inside a clang offload bundle): registers, spills, scratch and LDS of every kernel,
    as the compiler states them (`.vgpr_count`, `.vgpr_spill_count`, `.sgpr_spill_count`,
    `.private_segment_fixed_size`, `.group_segment_fixed_size`, ...)."""
    import struct
    start = binary.find(b'\x7fELF')
    if start < 0:
        return {}
    elf = memoryview(binary)[start:]
    shoff, = struct.unpack_from('<Q', elf, 0x28)
    shentsize, shnum = struct.unpack_from('<HH', elf, 0x3a)
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from('<I', elf, sh + 4)
        offset, size = struct.unpack_from('<QQ', elf, sh + 0x18)
        if sh_type != 7:                       # SHT_NOTE
            continue
        pos, end = offset, offset + size
        while pos + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from('<III', elf, pos)
            name_at = pos + 12
            desc_at = name_at + (namesz + 3) // 4 * 4
            if ntype == 32 and bytes(elf[name_at:name_at + 6]) == b'AMDGPU':
                meta, _ = _msgpack(elf[desc_at:desc_at + descsz])
                return {k['.name']: k for k in meta.get('amdhsa.kernels', [])}
            pos = desc_at + (descsz + 3) // 4 * 4
    return {}
