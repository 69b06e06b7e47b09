"""Device abstraction: a thin ctypes layer over the C ABI of ``libclray_hip.so``.

Plays the role of the reference's ``Accelerator`` (/root/reference/accelerator.py:11-229,
a pyopencl wrapper) with the same method names where the Renderer uses them:
``build_program``, ``call``, ``enqueue_copy``, ``to_device``, ``new_array``,
``new_const_buffer``, ``new_vec3_array``, ``make_vec3_array``, ``finish``,
``output_profiling_info``.  There is no CPU fallback: without the library or without a GPU
every constructor raises.
"""
import ctypes
import os
import time
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY_PATH = os.path.join(_HERE, 'libclray_hip.so')


class ClrError(RuntimeError):
    pass


class DeviceInfo(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char * 128), ('arch', ctypes.c_char * 64),
                ('compute_units', ctypes.c_int), ('clock_khz', ctypes.c_int),
                ('wavefront_size', ctypes.c_int), ('lds_bytes_per_block', ctypes.c_int),
                ('global_mem_bytes', ctypes.c_uint64)]


class RenderArgs(ctypes.Structure):
    _fields_ = [('struct_size', ctypes.c_size_t), ('img', ctypes.c_void_p), ('cam', ctypes.c_void_p),
                ('pass_tape', ctypes.c_void_p), ('bounce_tape', ctypes.c_void_p),
                ('material_rows', ctypes.c_void_p), ('vector_data', ctypes.c_void_p),
                ('integer_data', ctypes.c_void_p), ('n_pixels', ctypes.c_int),
                ('n_passes', ctypes.c_int), ('regenerate', ctypes.c_int),
                ('debug_dump', ctypes.c_void_p), ('per_pixel_rng', ctypes.c_int),
                ('rng_seed_lo', ctypes.c_uint), ('rng_seed_hi', ctypes.c_uint),
                ('first_pass', ctypes.c_int), ('first_ray', ctypes.c_int),
                ('pass_log', ctypes.c_void_p), ('pass_groups', ctypes.c_int),
                ('pass_stride', ctypes.c_int), ('reduce_after', ctypes.c_void_p),
                ('ray_band', ctypes.c_int), ('ray_band_stride', ctypes.c_int), ('ray_band_base', ctypes.c_int)]

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.struct_size = ctypes.sizeof(RenderArgs)


class KernelInfo(ctypes.Structure):
    _fields_ = [('vgprs', ctypes.c_int), ('scratch_bytes', ctypes.c_int), ('lds_bytes', ctypes.c_int),
                ('max_threads_per_block', ctypes.c_int)]


class OctreeInfo(ctypes.Structure):
    _fields_ = [('n_nodes', ctypes.c_int), ('n_leaves', ctypes.c_int), ('max_depth', ctypes.c_int),
                ('stream_length', ctypes.c_int), ('root_data_offset', ctypes.c_int),
                ('flat_triangles', ctypes.c_int), ('root_origin', ctypes.c_double * 3),
                ('root_size', ctypes.c_double)]


_lib = None


def load_library():
    """Load libclray_hip.so and declare the prototypes of include/clray_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBRARY_PATH):
        raise ClrError("%s is missing: build it with `python -c 'import __graft_entry__ as g; "
                       "g.build()'` (needs hipcc)" % LIBRARY_PATH)
    lib = ctypes.CDLL(LIBRARY_PATH)
    vp, ci, cp = ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p
    P = ctypes.POINTER
    sigs = {
        'clr_last_error': (cp, []),
        'clr_ctx_create': (ci, [ci, P(vp)]),
        'clr_ctx_destroy': (ci, [vp]),
        'clr_device_info': (ci, [vp, P(DeviceInfo)]),
        'clr_sync': (ci, [vp, vp]),
        'clr_stream_create': (ci, [vp, P(vp)]),
        'clr_stream_destroy': (ci, [vp, vp]),
        'clr_stream_wait': (ci, [vp, vp, vp]),
        'clr_timer_begin': (ci, [vp, vp]),
        'clr_timer_end': (ci, [vp, vp]),
        'clr_timer_elapsed': (ci, [vp, P(ctypes.c_float)]),
        'clr_last_split_ms': (ci, [vp, P(ctypes.c_float), P(ctypes.c_float)]),
        'clr_program_build': (ci, [vp, cp, ci, P(cp), P(vp)]),
        'clr_program_destroy': (ci, [vp]),
        'clr_program_set_global': (ci, [vp, cp, vp, ctypes.c_size_t]),
        'clr_program_get_global': (ci, [vp, cp, vp, ctypes.c_size_t]),
        'clr_compile_only': (ci, [cp, ci, P(cp), cp, P(ctypes.c_size_t)]),
        'clr_compile': (ci, [cp, ci, P(cp), cp, P(vp), P(ctypes.c_size_t)]),
        'clr_binary_free': (None, [vp]),
        'clr_program_load': (ci, [vp, vp, ctypes.c_size_t, P(vp)]),
        'clr_compiler_version': (ci, [P(ci), P(ci)]),
        'clr_kernel_info': (ci, [vp, cp, P(KernelInfo)]),
        'clr_buffer_alloc': (ci, [vp, ctypes.c_size_t, P(vp)]),
        'clr_buffer_free': (ci, [vp, vp]),
        'clr_buffer_upload': (ci, [vp, vp, vp, ctypes.c_size_t, vp]),
        'clr_buffer_download': (ci, [vp, vp, vp, ctypes.c_size_t, vp]),
        'clr_buffer_fill32': (ci, [vp, vp, ctypes.c_uint32, ctypes.c_size_t, vp]),
        'clr_buffer_copy': (ci, [vp, vp, vp, ctypes.c_size_t, vp]),
        'clr_find_non_negative': (ci, [vp, vp, vp, ctypes.c_size_t, P(ci), vp]),
        'clr_launch': (ci, [vp, cp, P(ctypes.c_uint), P(ctypes.c_uint), ci, P(vp), vp,
                            P(ctypes.c_float)]),
        'clr_render_passes': (ci, [vp, P(RenderArgs), vp, P(ctypes.c_float)]),
        'clr_render_passes_static_log': (ci, [vp, P(RenderArgs), vp, P(ctypes.c_float)]),
        'clr_first_hit': (ci, [vp, vp, vp, vp, vp, ci, vp, vp, vp, vp]),
        'clr_comm_create': (ci, [P(vp), ci, P(vp)]),
        'clr_comm_destroy': (ci, [vp]),
        'clr_comm_size': (ci, [vp, P(ci)]),
        'clr_allreduce_sum_f32': (ci, [vp, P(vp), ctypes.c_size_t, P(vp)]),
        'clr_allgather_f32': (ci, [vp, P(vp), P(vp), ctypes.c_size_t, P(vp)]),
        'clr_octree_build': (ci, [vp, ci, vp, ci, ci, ci, P(vp)]),
        'clr_octree_build_rule': (ci, [vp, ci, vp, ci, ci, ci, ci, P(vp)]),
        'clr_octree_info': (ci, [vp, P(OctreeInfo)]),
        'clr_octree_copy': (ci, [vp, vp, vp, vp]),
        'clr_octree_destroy': (ci, [vp]),
        'clr_mesh_read': (ci, [cp, ci, P(vp)]),
        'clr_mesh_info': (ci, [vp, P(ci), P(ci)]),
        'clr_mesh_copy': (ci, [vp, vp, vp]),
        'clr_mesh_destroy': (ci, [vp]),
        'clr_mesh_unique_faces': (ci, [vp, ci, vp, P(ci)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise ClrError(load_library().clr_last_error().decode('utf-8', 'replace'))


def _options_array(options):
    options = [o.encode() for o in options]
    arr = (ctypes.c_char_p * max(len(options), 1))(*options)
    return arr, len(options)


def compile_only(source, options=(), arch='gfx950'):
    """hiprtc-compile `source` without touching a device; returns the code-object size."""
    lib = load_library()
    arr, n = _options_array(options)
    size = ctypes.c_size_t(0)
    _check(lib.clr_compile_only(source.encode('ascii'), n, arr, arch.encode(), ctypes.byref(size)))
    return size.value


MESH_FORMATS = {'off': 0, 'zipper': 1, 'ply': 1}


def read_mesh(path, fmt=None):
    """Native reader for OFF / Stanford zipper files: (vertices float64 (n, 3), triangles int32
    (m, 3)), the arrays the reference's readers return as nested lists."""
    lib = load_library()
    if fmt is None:
        fmt = os.path.splitext(path)[1].lstrip('.').lower()
    if fmt not in MESH_FORMATS:
        raise ClrError("unknown mesh format %r (known: off, zipper/ply)" % fmt)
    handle = ctypes.c_void_p()
    _check(lib.clr_mesh_read(os.fsencode(path), MESH_FORMATS[fmt], ctypes.byref(handle)))
    try:
        nv, nt = ctypes.c_int(0), ctypes.c_int(0)
        _check(lib.clr_mesh_info(handle, ctypes.byref(nv), ctypes.byref(nt)))
        vertices = np.empty((nv.value, 3), dtype=np.float64)
        triangles = np.empty((nt.value, 3), dtype=np.int32)
        _check(lib.clr_mesh_copy(handle, vertices.ctypes.data_as(ctypes.c_void_p),
                                 triangles.ctypes.data_as(ctypes.c_void_p)))
    finally:
        lib.clr_mesh_destroy(handle)
    return vertices, triangles


def unique_faces(faces):
    """Indices of the faces that remove_duplicate_faces keeps (first of each vertex set)."""
    lib = load_library()
    faces = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
    kept = np.empty(faces.shape[0], dtype=np.int32)
    n = ctypes.c_int(0)
    _check(lib.clr_mesh_unique_faces(faces.ctypes.data_as(ctypes.c_void_p), faces.shape[0],
                                     kept.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n)))
    return kept[:n.value]


OVERLAP_RULES = {'reference': 0, 'sat': 1}


def build_octree(vertices, faces, max_depth, max_faces_per_leaf, overlap='reference'):
    """Native (C++) octree construction; host only.  Returns a dict with the reference-format
    node stream, the flattened nodes / triangle records and the root cube.
    overlap: 'reference' = the reference's bounding-sphere rule (its tree, its stream), 'sat' =
    exact triangle-box overlap (same hits, far fewer leaf records)."""
    lib = load_library()
    vertices = np.ascontiguousarray(vertices, dtype=np.float32)
    faces = np.ascontiguousarray(faces, dtype=np.int32)
    handle = ctypes.c_void_p()
    rc = lib.clr_octree_build_rule(vertices.ctypes.data_as(ctypes.c_void_p), vertices.shape[0],
                                   faces.ctypes.data_as(ctypes.c_void_p), faces.shape[0],
                                   int(max_depth), int(max_faces_per_leaf), OVERLAP_RULES[overlap],
                                   ctypes.byref(handle))
    if rc != 0:
        raise ClrError("clr_octree_build: out of memory" if rc == 2 else
                       "clr_octree_build: bad mesh or parameters (face index out of range, "
                       "max_depth outside 0..20)")
    try:
        info = OctreeInfo()
        lib.clr_octree_info(handle, ctypes.byref(info))
        stream = np.empty(info.stream_length, dtype=np.int32)
        nodes = np.empty((info.n_nodes, 2), dtype=np.int32)
        tris = np.empty((info.flat_triangles * 3, 4), dtype=np.float32)
        lib.clr_octree_copy(handle, stream.ctypes.data_as(ctypes.c_void_p),
                            nodes.ctypes.data_as(ctypes.c_void_p), tris.ctypes.data_as(ctypes.c_void_p))
    finally:
        lib.clr_octree_destroy(handle)
    return {'stream': stream, 'root_data_offset': info.root_data_offset, 'flat_nodes': nodes,
            'leaf_records': info.flat_triangles,
            'flat_tris': tris, 'root_origin': np.array(info.root_origin[:]), 'root_size': info.root_size,
            'n_nodes': info.n_nodes, 'n_leaves': info.n_leaves, 'max_depth': info.max_depth}


class DeviceArray:
    """A typed view of device memory (the pyopencl ``Array`` operations the Renderer needs)."""

    def __init__(self, acc, shape, dtype, ptr=None, owner=None):
        self.acc = acc
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._owner = owner
        if ptr is None:
            p = ctypes.c_void_p()
            _check(acc.lib.clr_buffer_alloc(acc.ctx, self.nbytes, ctypes.byref(p)))
            self.ptr = p.value
            self._owned = True
        else:
            self.ptr = ptr
            self._owned = False

    def __del__(self):
        if getattr(self, '_owned', False) and self.acc.ctx:
            self.acc.lib.clr_buffer_free(self.acc.ctx, ctypes.c_void_p(self.ptr))
            self._owned = False

    @property
    def data(self):
        return self.ptr

    def set(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.nbytes == self.nbytes, (host.shape, self.shape)
        _check(self.acc.lib.clr_buffer_upload(self.acc.ctx, ctypes.c_void_p(self.ptr),
                                              host.ctypes.data_as(ctypes.c_void_p), self.nbytes, None))
        return self

    def get(self):
        host = np.empty(self.shape, dtype=self.dtype)
        _check(self.acc.lib.clr_buffer_download(self.acc.ctx, host.ctypes.data_as(ctypes.c_void_p),
                                                ctypes.c_void_p(self.ptr), self.nbytes, None))
        return host

    def fill(self, value):
        assert self.dtype.itemsize == 4
        bits = int(np.array(value, dtype=self.dtype).view(np.uint32))
        _check(self.acc.lib.clr_buffer_fill32(self.acc.ctx, ctypes.c_void_p(self.ptr), bits,
                                              self.nbytes // 4, None))
        return self

    def __getitem__(self, key):
        if not (isinstance(key, slice) and key.step is None):
            raise TypeError("only contiguous slices of the first axis are supported")
        start, stop, _ = key.indices(self.shape[0])
        row = self.nbytes // max(self.shape[0], 1)
        return DeviceArray(self.acc, (max(stop - start, 0),) + self.shape[1:], self.dtype,
                           ptr=self.ptr + start * row, owner=self)


class Communicator:
    """RCCL communicator over the GPUs of several Accelerators of THIS process (C ABI clr_comm_*):
    the exchange step for a host that drives all GPUs from one process."""

    def __init__(self, accelerators):
        self.lib = load_library()
        self.accelerators = list(accelerators)
        ctxs = (ctypes.c_void_p * len(self.accelerators))(*[a.ctx for a in self.accelerators])
        self.comm = ctypes.c_void_p()
        _check(self.lib.clr_comm_create(ctxs, len(self.accelerators), ctypes.byref(self.comm)))

    def size(self):
        n = ctypes.c_int(0)
        _check(self.lib.clr_comm_size(self.comm, ctypes.byref(n)))
        return n.value

    def allreduce_sum(self, arrays):
        """In-place sum of equally shaped float32 DeviceArrays, one per GPU."""
        assert len(arrays) == len(self.accelerators) and all(a.dtype == np.float32 for a in arrays)
        count = arrays[0].nbytes // 4
        assert all(a.nbytes == arrays[0].nbytes for a in arrays)
        ptrs = (ctypes.c_void_p * len(arrays))(*[a.ptr for a in arrays])
        _check(self.lib.clr_allreduce_sum_f32(self.comm, ptrs, count, None))
        for acc in self.accelerators:
            acc.finish()

    def allgather(self, send, recv):
        """recv[i] (n x the size of a send array) = send[0] | send[1] | ... on every GPU."""
        count = send[0].nbytes // 4
        assert all(a.nbytes == send[0].nbytes for a in send) and all(a.nbytes == count * 4 * len(send) for a in recv)
        s = (ctypes.c_void_p * len(send))(*[a.ptr for a in send])
        r = (ctypes.c_void_p * len(recv))(*[a.ptr for a in recv])
        _check(self.lib.clr_allgather_f32(self.comm, s, r, count, None))
        for acc in self.accelerators:
            acc.finish()

    def close(self):
        if self.comm:
            self.lib.clr_comm_destroy(self.comm)
            self.comm = None


class Accelerator:

    warp_size = 64          # wavefront width of gfx950

    def __init__(self, device_index=0):
        self.lib = load_library()
        ctx = ctypes.c_void_p()
        self.ctx = None
        _check(self.lib.clr_ctx_create(int(device_index), ctypes.byref(ctx)))
        self.ctx = ctx
        self.prog = None
        self._streams = []
        self.profiling_info = {}
        self.last_code = None

    def close(self):
        if self.prog is not None:
            self.lib.clr_program_destroy(self.prog)
            self.prog = None
        if self.ctx:
            for stream in getattr(self, '_streams', []):
                self.lib.clr_stream_destroy(self.ctx, stream)
            self._streams = []
            self.lib.clr_ctx_destroy(self.ctx)
            self.ctx = None

    def device_info(self):
        info = DeviceInfo()
        _check(self.lib.clr_device_info(self.ctx, ctypes.byref(info)))
        return {f: (getattr(info, f).decode() if isinstance(getattr(info, f), bytes) else getattr(info, f))
                for f, _ in DeviceInfo._fields_}

    def finish(self, stream=None):
        _check(self.lib.clr_sync(self.ctx, stream))

    def new_stream(self):
        """A further in-order queue of this GPU (ordinary HIP stream; the default stream and it wait
        for each other's earlier work).  Released with the accelerator."""
        stream = ctypes.c_void_p()
        _check(self.lib.clr_stream_create(self.ctx, ctypes.byref(stream)))
        self._streams.append(stream)
        return stream

    def stream_wait(self, waiting, on):
        """`waiting` (None = default stream) waits for what is queued on `on` so far; the host does not."""
        _check(self.lib.clr_stream_wait(self.ctx, waiting, on))

    def last_split_ms(self):
        """(clr_render ms, clr_reduce_log ms) of the last timed launch of a pass-split program."""
        a, b = ctypes.c_float(0), ctypes.c_float(0)
        _check(self.lib.clr_last_split_ms(self.ctx, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def timer_begin(self, stream=None):
        """Start of a timed region of launches on `stream` (one HIP event pair around all of them)."""
        _check(self.lib.clr_timer_begin(self.ctx, stream))

    def timer_end(self, stream=None, wait=True):
        """End of the region: milliseconds, or with wait=False a function that waits and returns them."""
        _check(self.lib.clr_timer_end(self.ctx, stream))

        def elapsed():
            ms = ctypes.c_float(0)
            _check(self.lib.clr_timer_elapsed(self.ctx, ctypes.byref(ms)))
            return ms.value
        return elapsed() if wait else elapsed

    # ---- program
    def build_program(self, prog_code, options=(), keep_previous=False):
        """Compile and load `prog_code` and make it the current program.  keep_previous leaves
        the program that was current loaded (its handle stays valid: use_program / drop_program)."""
        if isinstance(options, str):
            options = options.split()
        self.last_code = prog_code
        if self.prog is not None and not keep_previous:
            self.lib.clr_program_destroy(self.prog)
            self.prog = None
        # compile (system hiprtc, disk store: program_cache.py), then load the code object
        from . import program_cache
        arch = self.device_info()['arch'].split(':')[0]
        binary, self.last_build = program_cache.get_binary(prog_code, list(options), arch)
        # what the compiler says about the kernels (registers, spills, scratch), per program handle
        self.last_metadata = program_cache.kernel_metadata(binary)
        prog = ctypes.c_void_p()
        _check(self.lib.clr_program_load(self.ctx, binary, len(binary), ctypes.byref(prog)))
        self.prog = prog
        return prog

    def use_program(self, prog):
        self.prog = prog

    def drop_program(self, prog):
        if prog is not None and prog is not self.prog:
            self.lib.clr_program_destroy(prog)

    def kernel_info(self, kernel='clr_render', prog=None):
        """Registers, scratch bytes per lane and LDS bytes of a kernel of the program."""
        info = KernelInfo()
        _check(self.lib.clr_kernel_info(prog if prog is not None else self.prog, kernel.encode(), ctypes.byref(info)))
        return {f: getattr(info, f) for f, _ in KernelInfo._fields_}

    def set_global(self, name, host):
        host = np.ascontiguousarray(host)
        _check(self.lib.clr_program_set_global(self.prog, name.encode(),
                                               host.ctypes.data_as(ctypes.c_void_p), host.nbytes))

    def get_global(self, name, shape, dtype):
        """Contents of a module global of the current program (waits for the device)."""
        host = np.zeros(shape, dtype=dtype)
        _check(self.lib.clr_program_get_global(self.prog, name.encode(),
                                               host.ctypes.data_as(ctypes.c_void_p), host.nbytes))
        return host

    def _record(self, name, device_seconds, host_seconds):
        info = self.profiling_info.setdefault(name, {'n': 0, 't': 0.0, 'ta': 0.0})
        info['n'] += 1
        info['t'] += device_seconds
        info['ta'] += host_seconds

    def call(self, kernel_name, ndrange_size, buffer_args, value_args=(), work_group_size=None,
             stream=None, prog=None):
        """Launch a kernel of the program over `ndrange_size` work items (1-D or 2-D)."""
        t0 = time.time()
        if isinstance(ndrange_size, int):
            ndrange_size = (ndrange_size,)
        block = [256, 1, 1] if work_group_size is None else list(work_group_size) + [1] * (3 - len(work_group_size))
        grid = [1, 1, 1]
        for i, n in enumerate(ndrange_size):
            grid[i] = (int(n) + block[i] - 1) // block[i]
        holders = []
        for a in tuple(buffer_args) + tuple(value_args):
            if isinstance(a, DeviceArray):
                holders.append(ctypes.c_void_p(a.ptr))
            elif a is None:
                holders.append(ctypes.c_void_p(0))
            elif isinstance(a, np.int64):
                holders.append(ctypes.c_longlong(int(a)))       # a Python int stays a C int
            elif isinstance(a, (np.integer, int)):
                holders.append(ctypes.c_int(int(a)))
            elif isinstance(a, np.float64):
                holders.append(ctypes.c_double(float(a)))      # a Python float stays a C float
            elif isinstance(a, (np.floating, float)):
                holders.append(ctypes.c_float(float(a)))
            else:
                raise TypeError("unsupported kernel argument %r" % (a,))
        argv = (ctypes.c_void_p * len(holders))(*[ctypes.cast(ctypes.pointer(h), ctypes.c_void_p) for h in holders])
        ms = ctypes.c_float(0)
        _check(self.lib.clr_launch(prog if prog is not None else self.prog, kernel_name.encode(), (ctypes.c_uint * 3)(*grid),
                                   (ctypes.c_uint * 3)(*block), len(holders), argv, stream,
                                   ctypes.byref(ms)))
        self._record(kernel_name, ms.value * 1e-3, time.time() - t0)

    def render_passes(self, args, stream=None, timed=True, static_log=False):
        t0 = time.time()
        ms = ctypes.c_float(0)
        entry = self.lib.clr_render_passes_static_log if static_log else self.lib.clr_render_passes
        _check(entry(self.prog, ctypes.byref(args), stream, ctypes.byref(ms) if timed else None))
        if timed:
            self._record('clr_render', ms.value * 1e-3, time.time() - t0)
        return ms.value

    # ---- buffers
    def enqueue_copy(self, dest, src):
        if isinstance(dest, DeviceArray):
            dest.set(np.asarray(src).reshape(dest.shape) if np.size(src) == int(np.prod(dest.shape)) else src)
        else:
            np.copyto(dest, src.get())

    def device_memcpy(self, dest, src, n=None):
        """Device-to-device copy of `n` 32-bit values, or of all of `src` (reference accelerator.py:224-229)."""
        nbytes = src.nbytes if n is None else int(n) * 4
        if nbytes > dest.nbytes or nbytes > src.nbytes:
            raise ValueError("device_memcpy: %d bytes do not fit" % nbytes)
        _check(self.lib.clr_buffer_copy(self.ctx, ctypes.c_void_p(dest.ptr), ctypes.c_void_p(src.ptr),
                                        ctypes.c_size_t(nbytes), None))
        self.finish()

    def find_non_negative(self, in_array, out_array, n):
        """Stable compaction on the device: the first `n` elements of the int32 array `in_array` that
        are >= 0 go to the front of `out_array` in their order; returns how many (reference
        accelerator.py:196-222, there a pyopencl scan kernel).  The fused render kernel does not need
        it; it is here for step-wise callers of the reference's interface."""
        if in_array.dtype != np.int32 or out_array.dtype != np.int32:
            raise TypeError("find_non_negative works on int32 arrays")
        n = int(n)
        if n * 4 > in_array.nbytes or n * 4 > out_array.nbytes:
            raise ValueError("find_non_negative: n exceeds an array")
        t0 = time.time()
        count = ctypes.c_int(0)
        _check(self.lib.clr_find_non_negative(self.ctx, ctypes.c_void_p(in_array.ptr), ctypes.c_void_p(out_array.ptr),
                                              ctypes.c_size_t(n), ctypes.byref(count), None))
        self._record('find_non_negative', 0.0, time.time() - t0)
        return count.value

    def to_device(self, cpuarray):
        return DeviceArray(self, cpuarray.shape, cpuarray.dtype).set(cpuarray)

    def new_array(self, shape, datatype=np.float32, zeros=False):
        arr = DeviceArray(self, shape, datatype)
        if zeros:
            arr.fill(0)
        return arr

    def new_const_buffer(self, buf, dtype=np.float32):
        return self.to_device(np.ascontiguousarray(buf, dtype=dtype))

    def new_vec3_array(self, shape):
        return self.new_array(tuple(shape) + (4,), np.float32, True)

    @staticmethod
    def host_vec3_array(a):
        """(…, 3) -> (…, 4) float32 with a zero fourth lane (float3 storage = float4)."""
        a = np.asarray(a)
        assert a.shape[-1] == 3
        out = np.zeros(a.shape[:-1] + (4,), dtype=np.float32)
        out[..., :3] = a
        return out

    def make_vec3_array(self, a):
        return self.to_device(self.host_vec3_array(a))

    make_const_vec3_buffer = make_vec3_array

    def empty_like(self, a):
        return DeviceArray(self, a.shape, a.dtype)

    def zeros_like(self, a):
        return DeviceArray(self, a.shape, a.dtype).fill(0)

    def output_profiling_info(self):
        total = host_total = 0.0
        for name, v in self.profiling_info.items():
            n = v['n']
            print('%d\t%.2g\t%.2g\t%.2g\t%.2g\t%s' % (n, v['t'], v['ta'], v['t'] / n, v['ta'] / n, name))
            total += v['t']
            host_total += v['ta']
        print('----', total, 'or', host_total, 'seconds total')
