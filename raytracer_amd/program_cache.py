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
