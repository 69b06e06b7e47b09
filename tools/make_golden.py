#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the reference's OWN code.

Runs only in the development container (it reads /root/reference, which never travels to
the GPU box).  Nothing of the reference is copied into the repository: the script

  1. makes a temporary Python 3 conversion of the reference under /tmp (lib2to3, plus the
     three integer divisions Python 3 changed), with a stub for the absent `pyopencl` import;
  2. lets the reference's own Renderer / Compiler / scene files pack a scene and render its
     OpenCL program text (the Jinja2 templates of the reference);
  3. compiles that OpenCL C for x86-64 with ROCm's clang (-O1, -ffp-contract=off), links it
     with a small shim defining the OpenCL built-ins it uses (dot, cross, length, normalize,
     native_sqrt, get_global_id, ...) and per-kernel trampolines;
  4. substitutes a numpy-backed Accelerator so that the reference's own render loop
     (Renderer.render_sample / compute_next_path_segment) drives those kernels, with
     np.random seeded;
  5. records what went in (camera rays, random numbers as uploaded, packed scene tables) and
     what came out (per-segment ray state of the first pass, image sums) as .npz.

Two patches are applied in the temporary copy and recorded in the fixture metadata:
  * with warp_size 1 the reference's culler reads `max_value` uninitialised; it
    is initialised to the ray's own intensity, which is what the warp reduction computes for
    a group of one ray;
  * ConvexIntersection.cl accumulates the sub-object offset with `set` inside a Jinja2 for
    loop, which Jinja2 < 2.9 carried across iterations and the installed 3.1 does not (every
    component would get offset 0 and the last component's normal would always win); the
    running offset is kept in a namespace object instead, restoring the intended output.

Usage:  python tools/make_golden.py [scene-dev scene-three-objects scene-test]
"""
import ctypes
import os
import re
import shutil
import subprocess
import sys
import types

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import golden_scenes  # noqa: E402  (scene descriptions shared with the tests)

REFERENCE = '/root/reference'
WORK = '/tmp/clray_golden'
REF3 = os.path.join(WORK, 'ref3')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tests', 'golden')
CLANG = '/opt/rocm/lib/llvm/bin/clang'

SHIM = r'''
// OpenCL C built-ins used by the reference's kernels, for the x86-64 build of its device code.
#include <cmath>
typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float3 __attribute__((ext_vector_type(3)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef int int3 __attribute__((ext_vector_type(3)));
typedef unsigned int uint;
static thread_local uint gid[3];
extern "C" void shim_set_gid(uint i, uint j) { gid[0] = i; gid[1] = j; gid[2] = 0; }
uint get_global_id(uint d) { return gid[d]; }
uint get_local_id(uint d) { return 0; }
void barrier(uint) {}
float native_sqrt(float x) { return sqrtf(x); }
float sqrt(float x) { return sqrtf(x); }
float exp(float x) { return expf(x); }
float log(float x) { return logf(x); }
float tan(float x) { return tanf(x); }
float max(float a, float b) { return fmaxf(a, b); }
float min(float a, float b) { return fminf(a, b); }
float clamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
float dot(float3 a, float3 b) { return a.x*b.x + a.y*b.y + a.z*b.z; }
float dot(float2 a, float2 b) { return a.x*b.x + a.y*b.y; }
float dot(float4 a, float4 b) { return a.x*b.x + a.y*b.y + a.z*b.z + a.w*b.w; }
float3 cross(float3 a, float3 b) { return (float3){a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x}; }
float length(float3 a) { return sqrtf(dot(a, a)); }
float length(float4 a) { return sqrtf(dot(a, a)); }
float3 normalize(float3 a) { return a / length(a); }
float3 fast_normalize(float3 a) { return a / length(a); }
float3 convert_float3(int3 a) { return (float3){(float)a.x, (float)a.y, (float)a.z}; }
int all(int3 a) { return (a.x < 0) && (a.y < 0) && (a.z < 0); }
int3 isfinite(float3 a) { return (int3){std::isfinite(a.x) ? -1 : 0, std::isfinite(a.y) ? -1 : 0, std::isfinite(a.z) ? -1 : 0}; }
'''


def sh(*cmd, **kw):
    subprocess.check_call(list(cmd), **kw)


def prepare_reference():
    """Temporary Python 3 copy of the reference + pyopencl stub (never committed)."""
    if os.path.exists(os.path.join(REF3, '.converted')):
        return
    shutil.rmtree(REF3, ignore_errors=True)
    os.makedirs(WORK, exist_ok=True)
    shutil.copytree(REFERENCE, REF3)
    sh(sys.executable, '-m', 'lib2to3', '-w', '-n', '.', cwd=REF3,
       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    def patch(path, old, new):
        p = os.path.join(REF3, path)
        text = open(p).read()
        assert old in text, (path, old)
        open(p, 'w').write(text.replace(old, new))
    patch('camera.py', 'n = n/2', 'n = n//2')
    patch('renderer.py', 'range(height / block_h)', 'range(height // block_h)')
    patch('renderer.py', 'range(width / block_w)', 'range(width // block_w)')
    # Jinja2 >= 2.9 no longer carries `set` assignments from one loop iteration to the next;
    # the reference's running sub-object offset (ConvexIntersection.cl:11,37,54,65) relies on
    # the pre-2.9 behaviour.  Restore it in the temporary copy with a namespace object.
    p = os.path.join(REF3, 'cl_templates', 'objects', 'ConvexIntersection.cl')
    text = open(p).read()
    assert text.count('### set subobj_offset = 0') == 2
    text = text.replace('### set subobj_offset = 0', '### set ns = namespace(off=0)')
    text = text.replace('### set subobj_offset = subobj_offset + component.n_subobjects',
                        '### set ns.off = ns.off + component.n_subobjects')
    text = text.replace('subobj_offset', 'ns.off')
    open(p, 'w').write(text)
    stub = os.path.join(REF3, 'pyopencl')
    os.makedirs(stub)
    open(os.path.join(stub, '__init__.py'), 'w').write(
        "VERSION = (2016, 2)\nclass mem_flags:\n    READ_ONLY = 1\n    COPY_HOST_PTR = 2\n    READ_WRITE = 4\n")
    open(os.path.join(stub, 'array.py'), 'w').write("class Array(object):\n    pass\n")
    open(os.path.join(stub, 'tools.py'), 'w').write("")
    open(os.path.join(REF3, '.converted'), 'w').write('ok')


def write_cie_table():
    """The reference reads the CIE 1931 colour matching functions from data/cie-1931.csv, a file
    it does not ship.  The temporary copy gets this package's table (the analytic
    fit in raytracer_amd/shading.py, on the 1 nm grid) in the format spectrum.py:40 expects:
    wavelength, x, y, z per line.  Everything downstream of the table is then the reference's
    own code: Spectrum.cie_1931_rgb / visible_intensity, SpectrumShader.initialize_material_buffers
    and init_sample (shader.py:148-194)."""
    sys.path.insert(0, ROOT)
    from raytracer_amd.shading import cie_1931_xyz_fit
    w = np.arange(360, 831, dtype=float)
    table = np.column_stack((w, cie_1931_xyz_fit(w).T))
    os.makedirs(os.path.join(REF3, 'data'), exist_ok=True)
    np.savetxt(os.path.join(REF3, 'data', 'cie-1931.csv'), table, delimiter=',', fmt='%.17g')


class HostArray(np.ndarray):
    """numpy array with the pyopencl Array methods the reference uses."""
    def get(self):
        return np.array(self)

    def __imul__(self, other):
        # pyopencl multiplies an Array by a host-side operand as by a scalar (shader.py:184 passes
        # a one-element array); the array holds 1.0 at that point, so the rounding is float32(other)
        if not isinstance(other, HostArray) and np.size(other) == 1:
            other = self.dtype.type(np.ravel(other)[0])
        return np.ndarray.__imul__(self, other)


def as_host(a):
    return np.ascontiguousarray(a).view(HostArray)


class FakeAccelerator:
    """numpy-backed stand-in for the reference's pyopencl Accelerator; kernels are the
    reference's OpenCL C compiled for the host."""
    warp_size = 1

    def __init__(self, *args):
        self.lib = None
        self.profiling_info = {}
        self.trace = []            # (kernel_name, value_args snapshot)
        self.hooks = []

    def finish(self):
        pass

    def build_program(self, prog_code, options=[]):
        if isinstance(prog_code, bytes):
            prog_code = prog_code.decode('ascii')
        old = "float max_value;\n    scratch[local_idx] = cur_intensity;"
        assert old in prog_code
        prog_code = prog_code.replace(old, "float max_value = cur_intensity;\n    scratch[local_idx] = cur_intensity;")
        self.source = prog_code
        tag = '%08x' % (hash(prog_code) & 0xffffffff)
        d = os.path.join(WORK, 'build_' + tag)
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, 'scene.cl'), 'w').write(prog_code)
        # kernel signatures -> trampolines that loop over the ND-range
        self.kernels = {}
        tramp = ['typedef unsigned int uint;', 'extern "C" void shim_set_gid(uint, uint);']
        for m in re.finditer(r'__kernel\s+void\s+(\w+)\s*\((.*?)\)\s*\{', prog_code, re.S):
            name, params = m.group(1), re.sub(r'//[^\n]*', '', m.group(2))
            kinds = []
            for p in params.split(','):
                p = p.strip()
                kinds.append('ptr' if '*' in p else ('float' if re.search(r'\bfloat\b', p) else 'int'))
            self.kernels[name] = kinds
            ctype = {'ptr': 'void*', 'float': 'float', 'int': 'int'}
            decl = ', '.join('%s a%d' % (ctype[k], i) for i, k in enumerate(kinds))
            call = ', '.join('a%d' % i for i in range(len(kinds)))
            tramp.append('extern "C" void %s(%s);' % (name, decl))
            tramp.append('extern "C" void run_%s(int n0, int n1, %s) { for (int j = 0; j < n1; ++j) '
                         'for (int i = 0; i < n0; ++i) { shim_set_gid(i, j); %s(%s); } }' % (name, decl, name, call))
        open(os.path.join(d, 'tramp.cpp'), 'w').write('\n'.join(tramp))
        open(os.path.join(d, 'shim.cpp'), 'w').write(SHIM)
        flags = ['-O1', '-ffp-contract=off', '-fPIC']
        sh(CLANG, '-x', 'cl', '-cl-std=CL1.2', '-Xclang', '-finclude-default-header',
           '-target', 'x86_64-unknown-linux-gnu', '-w', *flags, '-c', 'scene.cl', '-o', 'scene.o', cwd=d)
        sh(CLANG + '++', *flags, '-c', 'shim.cpp', '-o', 'shim.o', cwd=d)
        sh(CLANG + '++', *flags, '-c', 'tramp.cpp', '-o', 'tramp.o', cwd=d)
        sh(CLANG + '++', '-shared', 'scene.o', 'shim.o', 'tramp.o', '-o', 'refdev.so', cwd=d)
        undefined = subprocess.check_output(['nm', '-u', os.path.join(d, 'refdev.so')]).decode()
        missing = [l for l in undefined.splitlines() if '_Z' in l and 'GLIBC' not in l and 'CXXABI' not in l]
        assert not missing, missing
        self.lib = ctypes.CDLL(os.path.join(d, 'refdev.so'))
        return self.lib

    def call(self, kernel_name, ndrange_size, buffer_args, value_args=tuple([]), work_group_size=None):
        if isinstance(ndrange_size, (int, np.integer)):
            ndrange_size = (int(ndrange_size), 1)
        for hook in self.hooks:
            hook(kernel_name, ndrange_size, buffer_args, value_args)
        kinds = self.kernels[kernel_name]
        args = list(buffer_args) + list(value_args)
        assert len(args) == len(kinds), (kernel_name, len(args), len(kinds))
        conv = []
        for a, k in zip(args, kinds):
            if k == 'ptr':
                assert a is None or isinstance(a, np.ndarray), (kernel_name, type(a))
                conv.append(ctypes.c_void_p(None if a is None else a.ctypes.data))
            elif k == 'float':
                conv.append(ctypes.c_float(float(a)))
            else:
                conv.append(ctypes.c_int(int(a)))
        fn = getattr(self.lib, 'run_' + kernel_name)
        fn.restype = None
        fn(ctypes.c_int(int(ndrange_size[0])), ctypes.c_int(int(ndrange_size[1])), *conv)

    def enqueue_copy(self, dest, src):
        np.copyto(dest.reshape(-1)[:np.size(src)], np.ravel(src).astype(dest.dtype))

    def to_device(self, cpuarray):
        return as_host(np.array(cpuarray))

    def new_array(self, shape, datatype=np.float32, zeros=False):
        return as_host(np.zeros(shape, dtype=datatype))

    def new_const_buffer(self, buf, dtype=np.float32):
        return as_host(np.array(buf).astype(dtype))

    def new_vec3_array(self, shape):
        return as_host(np.zeros(tuple(shape) + (4,), dtype=np.float32))

    def _host_vec3(self, a):
        out = np.zeros(a.shape[:-1] + (4,), dtype=np.float32)
        out[..., :3] = a
        return out

    def make_vec3_array(self, a):
        return as_host(self._host_vec3(a))

    make_const_vec3_buffer = make_vec3_array

    def empty_like(self, a):
        return as_host(np.zeros_like(a))

    zeros_like = empty_like

    def find_non_negative(self, in_array, out_array, n):
        # stable copy-if over `__global int *` views, as pyopencl's GenericScanKernel does it
        signed = np.asarray(in_array[:n]).view(np.int32)
        keep = signed[signed >= 0]
        np.asarray(out_array).view(np.int32)[:keep.size] = keep
        return int(keep.size)

    def output_profiling_info(self):
        pass


def import_reference():
    prepare_reference()
    write_cie_table()
    os.chdir(REF3)                               # spectrum.py:40 opens data/cie-1931.csv relative to the cwd
    for p in (REF3, os.path.join(REF3, 'objects'), os.path.join(REF3, 'scenes')):
        if p not in sys.path:
            sys.path.insert(0, p)
    np.set_printoptions(legacy='1.25')       # numpy-2 scalar reprs would leak into the OpenCL text
    import jinja2
    jinja2.PackageLoader = lambda *a, **k: jinja2.FileSystemLoader(os.path.join(REF3, 'cl_templates'))
    import renderer as ref_renderer
    ref_renderer.Accelerator = FakeAccelerator
    return ref_renderer


SNAPSHOT_PASS = 0      # the pass whose per-segment ray state is recorded (development: set another)

COMPACT_KEYS = ('size', 'seed', 'n_passes', 'object_names', 'patch', 'pass_tape', 'bounce_tape', 'material_rows',
                'img', 'n_segments', 'seg0_pixel', 'seg0_whichobject', 'seg0_which_subobject', 'seg0_isec_dist')


def make_fixture(scene_name, size, n_passes, seed, compact=False):
    """compact: for a configuration's real size - keep the random numbers, the first hits of pass 0
    and the image sums only (the small fixture of the same scene pins everything else)."""
    ref_renderer = import_reference()
    np.random.seed(seed)
    for mod in [m for m in sys.modules if m == scene_name]:
        del sys.modules[mod]
    if scene_name in golden_scenes.VARIANTS:
        import types as _types
        import objects as ref_objects
        import shader as ref_shader
        import transformations as ref_transformations
        from scenes.default_scenes import BoxScene as RefBoxScene, DefaultScene as RefDefaultScene
        api = _types.SimpleNamespace(BoxScene=RefBoxScene, DefaultScene=RefDefaultScene,
                                     SpectrumShader=ref_shader.SpectrumShader,
                                     Affine=ref_transformations.Affine, Cube=ref_objects.Cube,
                                     Sphere=ref_objects.Sphere, TriangleMesh=ref_objects.TriangleMesh,
                                     Octree=ref_objects.Octree,
                                     Cylinder=ref_objects.Cylinder, HalfSpace=ref_objects.HalfSpace)
        scene = golden_scenes.VARIANTS[scene_name](api)
    else:
        scene = __import__(scene_name).scene
    scene.image.size = size
    scene.samples_per_pixel = max(n_passes, 4)
    args = types.SimpleNamespace(choose_opencl_context=False, cl_build_options=[])
    r = ref_renderer.Renderer(scene, args)
    acc = r.acc
    n_pixels = r.n_pixels
    max_bounces = scene.max_bounces

    # ---- inputs as the device sees them
    fx = {
        'size': np.array(size), 'seed': np.array(seed), 'n_passes': np.array(n_passes),
        'object_names': np.array([o.tracer.tracer_kernel_name for o in scene.objects]),
        'cam': np.array(r.cam), 'image_order': np.array(r.image_order),
        'root_object_id': np.array(r.root_object_id),
        'bidirectional_light_ids': np.array(r.bidirectional_light_ids),
        'data_pointer_offset': np.array(r.object_data_pointer_buffer_offset),
        'patch': np.array('culler: float max_value = cur_intensity (SURVEY F12); '
                          'ConvexIntersection.cl: running sub-object offset kept in a jinja2 '
                          'namespace (pre-2.9 loop scoping)'),
    }
    spectral = not r.shader.rgb
    if spectral:
        # host tables of SpectrumShader.initialize_material_buffers (shader.py:148-171), made by
        # the reference from the injected CIE table
        fx['spectral_host_mat_y'] = np.array(r.shader.host_mat_y)
        fx['spectral_color_responses'] = np.array(r.shader.color_responses)
        fx['spectral_intensity_pdf'] = np.array(r.shader.color_intensity_pdf)
        fx['spectral_wavelengths'] = np.array(scene.spectrum.wavelengths)
        fx['cie_table'] = np.array('CIE 1931 table: raytracer_amd.shading.cie_1931_xyz_fit on 360..830 nm, '
                                   'written to data/cie-1931.csv of the temporary copy (the reference ships none)')
        material_rows = np.zeros((n_passes, r.shader.host_mat_y.shape[0]), np.float32)
    else:
        fx['material_colors'] = np.array(r.shader.material_buffers[0])
        fx['material_scalars'] = np.array(r.shader.material_buffers[1])
    for key, buf in zip(['vector', 'integer'], r.tracer_data_buffers):
        if buf is not None:
            fx['data_' + key] = np.array(buf)
    for key, buf in zip(['param_float3', 'param_int', 'param_float'], r.tracer_const_data_buffers):
        if buf is not None:
            fx[key] = np.array(buf)

    # ---- record the random numbers as uploaded, per pass and per segment
    pass_tape = np.zeros((n_passes, 6, 4), np.float32)
    bounce_tape = np.zeros((n_passes, max_bounces, 5, 4), np.float32)
    state = {'pass': -1, 'segment': -1}
    snapshots = []

    def hook(name, ndrange, buffer_args, value_args):
        j, b = state['pass'], state['segment']
        if name == 'subsample_transform_camera':
            pass_tape[j, 0:4, :] = value_args[0][:4]
        elif name == 'fill_vec_broadcast':
            pass_tape[j, 4, 0:3] = value_args[0][0, :3]
            pass_tape[j, 4, 3] = 1.0
            pass_tape[j, 5, 0:3] = 1.0
        elif name == 'shader_volumetric':
            rec = bounce_tape[j, b]
            vb = value_args[-1]
            rec[0, 0:3] = vb[0, :3]
            rec[0, 3] = buffer_args[1][0]            # the shared stage sample p
            rec[1, 0:3] = vb[1, :3]
            if r.bidirectional:
                light_id1, light_area, min_dist = value_args[0], value_args[1], value_args[2]
                rec[2, 0:3], rec[2, 3] = vb[3, :3], light_area
                rec[3, 0:3], rec[3, 3] = vb[4, :3], min_dist
                rec[4, 0:3] = vb[5, :3]
                rec[4, 3:4] = np.array([light_id1], np.int32).view(np.float32)
        elif name == 'culler':
            bounce_tape[j, b, 1, 3] = value_args[0]
    acc.hooks.append(hook)

    if spectral:
        # per pass: what SpectrumShader.init_sample (shader.py:173-194) left on the "device"
        original_init_sample = r.shader.init_sample

        def init_sample(renderer):
            original_init_sample(renderer)
            j = state['pass']
            material_rows[j] = np.ravel(np.array(r.shader.device_material_buffer))
            pass_tape[j, 4, 3] = np.array(r.ray_state.raycolor)[0]          # 1 / pdf(wavelength)
            pass_tape[j, 5, 0:3] = r.vec_param_buf[2, :3]                    # colour mask
        r.shader.init_sample = init_sample

    # ---- wrap the per-segment method to snapshot the ray state of pass 0
    original = r.compute_next_path_segment

    def wrapped(sample_index, path_index, is_last):
        state['segment'] = path_index
        n = r.cur_n_pixels
        pixel_before = np.array(r.ray_state.pixel[:n])
        ok = original(sample_index, path_index, is_last)
        if state['pass'] == SNAPSHOT_PASS:
            rs = r.ray_state
            snap = {'n': n, 'pixel': pixel_before}
            # thread-indexed scratch -> gather as is; ray-indexed state -> index by pixel
            for key in ('isec_dist', 'normal', 'prob'):
                snap[key] = np.array(getattr(rs, key)[:n])
            if r.bidirectional:
                snap['shadow_mask'] = np.array(rs.shadow_mask[:n])
                snap['suppress_emission'] = np.array(rs.suppress_emission)[pixel_before]
            for key in ('whichobject', 'which_subobject', 'pos', 'ray', 'raycolor', 'inside', 'diffusions_left'):
                snap[key] = np.array(getattr(rs, key))[pixel_before]
            alive = np.zeros(n_pixels, bool)
            if not is_last:
                alive[np.array(rs.pixel[:r.cur_n_pixels])] = True
            snap['alive'] = alive[pixel_before]
            snap['img'] = np.array(r.img)
            snapshots.append(snap)
        return ok
    r.compute_next_path_segment = wrapped

    images = []
    for j in range(n_passes):
        state['pass'] = j
        r.render_sample(j)
        images.append(np.array(r.img))
    fx['pass_tape'], fx['bounce_tape'] = pass_tape, bounce_tape
    if spectral:
        fx['material_rows'] = material_rows
    fx['img_after_first_pass'] = images[0]
    if os.environ.get('CLRAY_GOLDEN_ALL_IMAGES'):      # development: image sums after every pass
        fx['img_per_pass'] = np.array(images)
    fx['img'] = images[-1]
    fx['n_segments'] = np.array(len(snapshots))
    for b, snap in enumerate(snapshots):
        for key, value in snap.items():
            fx['seg%d_%s' % (b, key)] = np.asarray(value)
    fx['program_kernels'] = np.array(sorted(acc.kernels))
    if compact:
        fx = {key: value for key, value in fx.items() if key in COMPACT_KEYS}
        fx['img'] = fx['img'][:, :3].copy()
        for key in ('seg0_pixel', 'seg0_whichobject', 'seg0_which_subobject'):
            fx[key] = fx[key].astype(np.int32)
    return fx


def mesh_file_texts(rng):
    """Generated OFF / zipper files that exercise the reference readers' record rules
    (mesh_formats.py:14-70): counts on the OFF line or the next one, polygons of 3-6 vertices
    (fan-triangulated), extra columns after the coordinates, repeated faces, exponent notation."""
    def vertex_lines(n, extra):
        lines = []
        for _ in range(n):
            xyz = rng.uniform(-2, 2, 3)
            fields = ['%.9g' % xyz[0], '%r' % float(np.float32(xyz[1])), '%.3e' % xyz[2]]
            fields += ['%.4f' % v for v in rng.uniform(0, 1, extra)]
            lines.append(' '.join(fields))
        return lines

    def face_lines(n_faces, n_vertices, sizes):
        lines, faces = [], []
        for i in range(n_faces):
            if faces and rng.rand() < 0.15:                  # a repeated face, rotated or reversed
                ids = list(faces[rng.randint(len(faces))])
                ids = ids[1:] + ids[:1] if rng.rand() < 0.5 else ids[::-1]
            else:
                ids = list(rng.choice(n_vertices, size=sizes[rng.randint(len(sizes))], replace=False))
            faces.append(ids)
            lines.append(' '.join(str(v) for v in [len(ids)] + ids))
        return lines
    texts = {}
    texts['a.off'] = '\n'.join(['OFF', '40 60 0'] + vertex_lines(40, 0) + face_lines(60, 40, [3, 4, 5, 6])) + '\n'
    texts['b.off'] = '\n'.join(['OFF 25 30 77'] + vertex_lines(25, 2) + face_lines(30, 25, [3, 3, 4])) + '\n'
    texts['c.ply'] = '\n'.join(['ply', 'format ascii 1.0', 'comment zipper output', 'element vertex 30',
                                'property float x', 'property float y', 'property float z',
                                'property float confidence', 'property float intensity',
                                'element face 45', 'property list uchar int vertex_indices', 'end_header']
                               + vertex_lines(30, 2) + face_lines(45, 30, [3, 3, 3, 4])) + '\n'
    return texts


def make_host_io_fixture(seed=5):
    """Rows f2 / f3 of SURVEY section 8: the reference's own mesh readers and tone mapper run on
    generated inputs (mesh_formats.read_off / read_zipper / remove_duplicate_faces,
    imgutils.Image._to_24bit without and with the flare effect, imgutils.flares by scipy.ndimage
    as installed here).  The fixture keeps the inputs (file texts
    written by this script, image sums) and what the reference returned."""
    import_reference()
    import io
    import mesh_formats as ref_mesh
    import imgutils as ref_img
    rng = np.random.RandomState(seed)
    fx = {}
    for name, text in mesh_file_texts(rng).items():
        reader = ref_mesh.read_off if name.endswith('.off') else ref_mesh.read_zipper
        with open(os.devnull, 'w') as sink:            # the readers print progress
            stdout, sys.stdout = sys.stdout, sink
            try:
                vertices, faces = reader(io.StringIO(text))
            finally:
                sys.stdout = stdout
        unique = ref_mesh.remove_duplicate_faces(faces)
        key = name.replace('.', '_')
        fx['mesh_%s_text' % key] = np.frombuffer(text.encode('ascii'), dtype=np.uint8)
        fx['mesh_%s_vertices' % key] = np.array([v[:3] for v in vertices], dtype=np.float64)
        fx['mesh_%s_vertex_columns' % key] = np.array([len(v) for v in vertices])
        fx['mesh_%s_faces' % key] = np.array(faces, dtype=np.int32)
        fx['mesh_%s_unique_faces' % key] = np.array(unique, dtype=np.int32)
    # tone mapping: sums as a float32 accumulator would hold them, with black, dim, over-exposed
    # and (for the nan_to_num step) a few NaN pixels
    height, width = 48, 64
    sums = rng.gamma(0.7, 30.0, size=(height, width, 3)).astype(np.float32)
    sums[:4] *= 0.001
    sums[10:14, 20:30] *= 50.0
    sums[5, 5] = 0.0
    base = (rng.gamma(0.7, 12.0, size=(height, width, 3)) + 0.5)
    fx['tonemap_sums'] = sums
    fx['tonemap_base'] = base
    cases = [('default', {}, False), ('gamma16_dim', {'gamma': 1.6, 'brightness': 0.3}, False),
             ('no_floor', {'low_normalization': None}, False), ('appended', {}, True),
             ('appended_no_floor', {'low_normalization': None, 'gamma': 2.2}, True),
             ('flares', {'flares': True}, False), ('flares_appended', {'flares': True, 'gamma': 1.6}, True)]
    for name, changes, with_base in cases:
        settings = ref_img.ColorEncodingSettings()
        for key, value in changes.items():
            setattr(settings, key, value)
        image = ref_img.Image(data=base.copy() if with_base else None, settings=settings)
        fx['tonemap_%s_rgb' % name] = image._to_24bit(sums.astype(np.float64))
        fx['tonemap_%s_settings' % name] = np.array([settings.gamma, settings.brightness,
                                                     0.0 if settings.low_normalization is None else 1.0,
                                                     1.0 if with_base else 0.0,
                                                     1.0 if settings.flares else 0.0])
    fx['tonemap_cases'] = np.array([c[0] for c in cases])
    return fx


def main():
    if sys.argv[1:] == ['host-io']:
        os.makedirs(OUT, exist_ok=True)
        path = os.path.join(OUT, 'host-io.npz')
        np.savez_compressed(path, **make_host_io_fixture())
        print('host-io ->', path, '%.1f kB' % (os.path.getsize(path) / 1e3))
        return
    names = sys.argv[1:] or ['scene-dev', 'scene-three-objects', 'scene-test', 'variant-camera', 'variant-media',
                             'spectral-dev', 'spectral-stairs']
    os.makedirs(OUT, exist_ok=True)
    for name in names:
        if '@' in name:                 # e.g. scene-test@320x240x16: a configuration's real size, compact
            name, shape = name.split('@')
            width, height, n_passes = (int(v) for v in shape.split('x'))
            fx = make_fixture(name, (width, height), n_passes=n_passes, seed=11, compact=True)
            path = os.path.join(OUT, '%s-%dx%d.npz' % (name, width, height))
        else:
            fx = make_fixture(name, (64, 48), n_passes=3, seed=11)
            path = os.path.join(OUT, name + '.npz')
        np.savez_compressed(path, **fx)
        print(name, '->', path, '%.1f kB' % (os.path.getsize(path) / 1e3),
              'segments recorded:', int(fx['n_segments']))


if __name__ == '__main__':
    main()
