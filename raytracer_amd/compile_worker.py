"""Stand-alone compile step: hiprtc-compiles the HIP source on stdin to a code object on stdout,
in a process of its own.  Run by program_cache.py as `python compile_worker.py <arch> [options]`
when the calling process has another hiprtc mapped than the system's (PyTorch wheels bring their
own).  Uses only ctypes and libclray_hip.so; reads and writes no files; never touches a GPU."""
import ctypes
import os
import sys


COMPILE_ERROR = 3        # exit status: the compiler ran and refused the source (stderr = its log)


def main(argv):
    arch = argv[1]
    options = [o.encode() for o in argv[2:]]
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libclray_hip.so'))
    lib.clr_last_error.restype = ctypes.c_char_p
    lib.clr_compile.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_char_p,
                                ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    lib.clr_binary_free.argtypes = [ctypes.c_void_p]
    source = sys.stdin.buffer.read()
    arr = (ctypes.c_char_p * max(len(options), 1))(*options)
    binary, size = ctypes.c_void_p(), ctypes.c_size_t(0)
    if lib.clr_compile(source, len(options), arr, arch.encode(), ctypes.byref(binary), ctypes.byref(size)):
        sys.stderr.write(lib.clr_last_error().decode('utf-8', 'replace'))
        return COMPILE_ERROR
    sys.stdout.buffer.write(ctypes.string_at(binary, size.value))
    sys.stdout.buffer.flush()
    lib.clr_binary_free(binary)
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv))
