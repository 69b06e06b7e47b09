"""ctypes front end of the CPU oracle (TEST INFRASTRUCTURE ONLY - see clray_oracle.h).

Builds an ``orc_scene`` from the shape objects of a scene (not from the product's packed
parameter table) and runs the C restatement on the same camera rays and random tape as the
HIP back end.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.
"""
import ctypes
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD_DIR = os.path.join(HERE, '_build')

(ORC_SPHERE, ORC_HALFSPACE, ORC_CONVEX, ORC_TRIMESH, ORC_OCTREE, ORC_QJS2, ORC_IMPLICIT, ORC_DISTFIELD,
 ORC_QJS_IA) = range(1, 10)
COMPONENT_KINDS = {'SphereComponent': 1, 'HalfSpaceComponent': 2, 'LayerComponent': 3,
                   'ZLayerComponent': 4, 'CylinderComponent': 5, 'ZCylinderComponent': 6,
                   'ConeComponent': 7, 'ZConeComponent': 8}

TRACER_FN = ctypes.CFUNCTYPE(None, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                             ctypes.c_float, ctypes.POINTER(ctypes.c_float),
                             ctypes.POINTER(ctypes.c_uint), ctypes.c_int, ctypes.c_int)
NORMAL_FN = ctypes.CFUNCTYPE(None, ctypes.POINTER(ctypes.c_float), ctypes.c_uint,
                             ctypes.POINTER(ctypes.c_float))


class Component(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int), ('n_subobjects', ctypes.c_int),
                ('pos', ctypes.c_float * 3), ('v', ctypes.c_float * 3), ('s', ctypes.c_float)]


class Object(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int), ('convex', ctypes.c_int), ('auto_flip_normal', ctypes.c_int),
                ('inv', ctypes.c_float * 12), ('comp_first', ctypes.c_int), ('comp_count', ctypes.c_int),
                ('vec_offset', ctypes.c_int), ('int_offset', ctypes.c_int),
                ('ip', ctypes.c_int * 8), ('fp', ctypes.c_float * 8),
                ('tracer', ctypes.c_void_p), ('normal', ctypes.c_void_p)]


class Scene(ctypes.Structure):
    _fields_ = [('n_objects', ctypes.c_int), ('objects', ctypes.POINTER(Object)),
                ('components', ctypes.POINTER(Component)),
                ('vector_data', ctypes.POINTER(ctypes.c_float)),
                ('integer_data', ctypes.POINTER(ctypes.c_int)),
                ('rgb', ctypes.c_int),
                ('material_colors', ctypes.POINTER(ctypes.c_float)),
                ('material_scalars', ctypes.POINTER(ctypes.c_float)),
                ('max_bounces', ctypes.c_int), ('min_bounces', ctypes.c_int),
                ('root_object_id', ctypes.c_int), ('bidirectional', ctypes.c_int),
                ('min_russian_prob', ctypes.c_float), ('max_ray_length', ctypes.c_float),
                ('light_samples', ctypes.c_int)]


def build(force=False):
    """Compile the oracle (gcc) if its shared objects are missing; returns the build dir."""
    target = os.path.join(BUILD_DIR, 'libclray_oracle.so')
    if force or not os.path.exists(target):
        subprocess.check_call(['make', '-C', HERE, '-s'] + (['-B'] if force else []))
    return BUILD_DIR


_libs = {}


class Rng(ctypes.Structure):
    _fields_ = [('seed_lo', ctypes.c_uint), ('seed_hi', ctypes.c_uint), ('first_pass', ctypes.c_int),
                ('first_ray', ctypes.c_int), ('n_lights', ctypes.c_int),
                ('lights', ctypes.POINTER(ctypes.c_float)), ('pass_stride', ctypes.c_int),
                ('pixel_ids', ctypes.POINTER(ctypes.c_int))]


def load(count_flops=False):
    name = 'libclray_oracle_flops.so' if count_flops else 'libclray_oracle.so'
    if name not in _libs:
        build()
        lib = ctypes.CDLL(os.path.join(BUILD_DIR, name))
        fp = ctypes.POINTER(ctypes.c_float)
        lib.orc_render.restype = ctypes.c_int
        lib.orc_render.argtypes = [ctypes.POINTER(Scene), fp, fp, fp, fp, ctypes.c_int, ctypes.c_int,
                                   fp, fp, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(Rng)]
        lib.orc_render_wavefront.restype = ctypes.c_int
        lib.orc_render_wavefront.argtypes = [ctypes.POINTER(Scene), fp, fp, fp, fp, ctypes.c_int, ctypes.c_int,
                                             fp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        lib.orc_first_hit.restype = ctypes.c_int
        lib.orc_first_hit.argtypes = [ctypes.POINTER(Scene), fp, fp, ctypes.c_int,
                                      ctypes.POINTER(ctypes.c_int), fp, fp, ctypes.c_int]
        _libs[name] = lib
    return _libs[name]


def _fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _c_float(value):
    return '%rf' % float(value)


def generated_shape_source(tracer, n):
    """C callbacks for one ImplicitSurface / DistanceField: the user's expressions (printed by
    sympy, or the code string) wrapped in the oracle's marching loops."""
    from raytracer_amd import objects as shapes
    bound = tracer.bndR
    has_bound, bound_r2 = (1, _c_float(bound * bound)) if bound else (0, '0.0f')
    if isinstance(tracer, shapes.ImplicitSurface):
        return '''
static ia_type f_%(n)d(ia_type x, ia_type y, ia_type z, const void *ctx) {
    ia_type f; (void)ctx;
    %(f_code)s
    return f;
}
static ia_type df_%(n)d(const ia_type ray[3], const ia_type origin[3], ia_type t, const void *ctx) {
    const ia_type ray_x = ray[0], ray_y = ray[1], ray_z = ray[2];
    const ia_type origin_x = origin[0], origin_y = origin[1], origin_z = origin[2];
    ia_type df; (void)ctx;
    %(df_code)s
    return df;
}
void orc_gen_tracer_%(n)d(const float origin[3], const float ray[3], float old_isec_dist, float *p_new,
                          unsigned *p_subobject, int inside, int origin_self) {
    const orc_ia_rayfn fn = {f_%(n)d, df_%(n)d, 0};
    (void)p_subobject;
    orc_implicit_tracer(&fn, origin, ray, old_isec_dist, p_new, inside, origin_self, %(no_self)d,
                        %(has_bound)d, %(bound_r2)s, %(max_itr)d, %(eps)s);
}
void orc_gen_normal_%(n)d(const float p[3], unsigned subobject, float normal[3]) {
    const struct { float x, y, z; } pos = {p[0], p[1], p[2]};
    (void)subobject;
    normal[0] = %(gx)s; normal[1] = %(gy)s; normal[2] = %(gz)s;
    orc_normalize(normal);
}
''' % dict(n=n, f_code=tracer.f_code, df_code=tracer.df_code, no_self=int(tracer.no_self_intersection),
           has_bound=has_bound, bound_r2=bound_r2, max_itr=tracer.max_itr, eps=_c_float(tracer.precision),
           gx=tracer.gradient_code[0], gy=tracer.gradient_code[1], gz=tracer.gradient_code[2])
    # DistanceField
    if tracer.normal_code:
        normal_body = '''
    float x = p[0], y = p[1], z = p[2];
    struct { float x, y, z; } normal_ = {0, 0, 0};
    #define normal normal_
    { %s }
    #undef normal
    normal[0] = normal_.x; normal[1] = normal_.y; normal[2] = normal_.z;''' % tracer.normal_code
    else:
        lines = ['    const float EPS = %s;' % _c_float(tracer.precision), '    float delta;']
        for axis, var in enumerate('xyz'):
            lines.append('    delta = 0.0f;')
            for sign in (1, -1):
                args = ['p[0]', 'p[1]', 'p[2]']
                lines.append('    { float x = p[0], y = p[1], z = p[2]; %s += 0.5f * EPS * %d; '
                             'delta += dist_%d(x, y, z) * %d; }' % (var, sign, n, sign))
            lines.append('    normal[%d] = delta;' % axis)
        normal_body = '\n'.join(lines)
    return '''
static float dist_%(n)d(float x, float y, float z) {
    float dist = 0.0f;
    { %(code)s; }
    return dist;
}
void orc_gen_tracer_%(n)d(const float origin[3], const float ray[3], float old_isec_dist, float *p_new,
                          unsigned *p_subobject, int inside, int origin_self) {
    (void)p_subobject;
    orc_distance_field_tracer(dist_%(n)d, origin, ray, old_isec_dist, p_new, inside, origin_self,
                              %(no_self)d, %(has_bound)d, %(bound_r2)s, %(max_itr)d, %(eps)s);
}
void orc_gen_normal_%(n)d(const float p[3], unsigned subobject, float normal[3]) {
    (void)subobject;
%(normal_body)s
    orc_normalize(normal);
}
''' % dict(n=n, code=tracer.tracer_code, no_self=int(tracer.no_self_intersection), has_bound=has_bound,
           bound_r2=bound_r2, max_itr=tracer.max_itr, eps=_c_float(tracer.precision),
           normal_body=normal_body)


class OracleScene:
    """The scene of a (host-only) Renderer in the oracle's data-driven form."""

    def __init__(self, renderer):
        self.renderer = renderer
        scene = renderer.scene
        self._keep = []
        self._generated = []
        objects = (Object * len(scene.objects))()
        components = []
        vectors, integers = [], []
        n_vec = n_int = 0
        for i, obj in enumerate(scene.objects):
            tracer = obj.tracer
            o = objects[i]
            o.convex = int(tracer.convex)
            o.auto_flip_normal = int(tracer.auto_flip_normal)
            inverse = tracer.global_to_tracer_coordinate_transform().inverse()
            o.inv[:] = list(np.float32(inverse.linear).ravel()) + list(np.float32(inverse.translation))
            o.vec_offset, o.int_offset = n_vec, n_int
            if tracer.has_data():
                data = tracer.get_data()
                raw = np.asarray(data['vector'], dtype=np.float32)
                vec = np.zeros((raw.shape[0], 4), dtype=np.float32)
                vec[:, :raw.shape[1]] = raw
                vectors.append(vec)
                n_vec += vec.shape[0]
                ints = np.ravel(data['integer']).astype(np.int32)
                integers.append(ints)
                n_int += ints.size
            self._fill_kind(o, tracer, components)
        self._attach_generated()
        comp_array = (Component * max(len(components), 1))(*components)
        self.vector_data = np.vstack(vectors) if vectors else np.zeros((1, 4), np.float32)
        self.integer_data = np.concatenate(integers) if integers else np.zeros(1, np.int32)
        sc = Scene()
        sc.n_objects = len(scene.objects)
        sc.objects = objects
        sc.components = comp_array
        sc.vector_data = _fptr(self.vector_data)
        sc.integer_data = self.integer_data.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
        sc.rgb = int(renderer.shader.rgb)
        if renderer.shader.rgb:
            self.colors = np.ascontiguousarray(renderer.material_colors, dtype=np.float32)
            self.scalars = np.ascontiguousarray(renderer.material_scalars, dtype=np.float32)
            sc.material_colors = _fptr(self.colors)
            sc.material_scalars = _fptr(self.scalars)
        sc.max_bounces = scene.max_bounces
        sc.light_samples = getattr(renderer, 'light_samples', 1)
        sc.min_bounces = scene.min_bounces
        sc.root_object_id = renderer.root_object_id
        sc.bidirectional = int(renderer.bidirectional)
        sc.min_russian_prob = float(scene.min_russian_prob or 0.0)
        sc.max_ray_length = float(scene.max_ray_length)
        self._keep += [objects, comp_array]
        self.c_scene = sc

    def _fill_kind(self, o, tracer, components):
        from raytracer_amd import objects as shapes
        if isinstance(tracer, shapes.Sphere):
            o.kind = ORC_SPHERE
        elif isinstance(tracer, shapes.HalfSpace):
            o.kind = ORC_HALFSPACE
        elif isinstance(tracer, shapes.ConvexIntersection):
            o.kind = ORC_CONVEX
            o.comp_first, o.comp_count = len(components), len(tracer.components)
            for c in tracer.components:
                rec = Component()
                rec.kind = COMPONENT_KINDS[type(c).__name__]
                rec.n_subobjects = c.n_subobjects
                rec.pos[:] = [float(x) for x in np.float32(c.position)]
                direction = getattr(c, 'normal', None) if hasattr(c, 'normal') else getattr(c, 'axis', (0, 0, 0))
                rec.v[:] = [float(x) for x in np.float32(direction)]
                for attr in ('h', 'R', 'slope'):
                    if hasattr(c, attr):
                        rec.s = float(np.float32(getattr(c, attr)))
                components.append(rec)
        elif isinstance(tracer, shapes.Octree):
            o.kind = ORC_OCTREE
            if tracer.layout != 'stream':
                raise NotImplementedError("the oracle follows the reference's octree stream; build the "
                                          "same tree with layout='stream' for it")
            o.ip[4], o.ip[5] = tracer.levels, tracer.max_steps
            values = tracer.parameter_values()
            o.ip[0], o.ip[1], o.ip[2], o.ip[3] = [int(v) for v in values[:4]]
            o.fp[0:3] = [float(x) for x in np.float32(values[4])]
            o.fp[3] = float(np.float32(values[5]))
            o.ip[7] = int(tracer.triangle_mesh.shading != 'flat')
        elif isinstance(tracer, shapes.TriangleMesh):
            o.kind = ORC_TRIMESH
            o.ip[0], o.ip[1] = tracer.n_vertices, tracer.n_faces
            o.ip[7] = int(tracer.shading != 'flat')
        elif isinstance(tracer, shapes.QuaternionJuliaSet2):
            o.kind = ORC_QJS2
            o.ip[0], o.ip[1] = tracer.julia_itr, tracer.max_itr
            o.fp[0:4] = [float(np.float32(x)) for x in tracer.c]
            o.fp[4], o.fp[5], o.fp[6] = tracer.bndR, tracer.precision, tracer.max_step
        elif isinstance(tracer, shapes.QuaternionJuliaSet):
            o.kind = ORC_QJS_IA
            o.ip[0], o.ip[1] = tracer.julia_itr, tracer.max_itr
            o.fp[0:4] = [float(np.float32(x)) for x in tracer.c]
            o.fp[4], o.fp[5] = tracer.bndR, tracer.precision
        elif isinstance(tracer, (shapes.ImplicitSurface, shapes.DistanceField)):
            o.kind = ORC_IMPLICIT if isinstance(tracer, shapes.ImplicitSurface) else ORC_DISTFIELD
            self._generated.append((o, tracer))
        else:
            raise NotImplementedError("oracle has no restatement of %s yet" % type(tracer).__name__)

    # ---- shapes whose function is user input: generate, compile and attach C callbacks
    def _attach_generated(self):
        if not self._generated:
            return
        import hashlib
        parts = ['#include <tgmath.h>', '#include "clray_oracle_gen.h"', '']
        for n, (_, tracer) in enumerate(self._generated):
            parts.append(generated_shape_source(tracer, n))
        source = '\n'.join(parts)
        tag = hashlib.sha1(source.encode()).hexdigest()[:16]
        os.makedirs(BUILD_DIR, exist_ok=True)
        c_path = os.path.join(BUILD_DIR, 'gen_%s.c' % tag)
        so_path = os.path.join(BUILD_DIR, 'gen_%s.so' % tag)
        if not os.path.exists(so_path):
            with open(c_path, 'w') as f:
                f.write(source)
            subprocess.check_call(['gcc', '-O2', '-fPIC', '-std=gnu11', '-ffp-contract=off', '-fno-fast-math',
                                   '-I', HERE, '-shared', c_path, '-o', so_path, '-lm'])
        lib = ctypes.CDLL(so_path)
        self._keep.append(lib)
        for n, (o, _) in enumerate(self._generated):
            o.tracer = ctypes.cast(getattr(lib, 'orc_gen_tracer_%d' % n), ctypes.c_void_p).value
            o.normal = ctypes.cast(getattr(lib, 'orc_gen_normal_%d' % n), ctypes.c_void_p).value

    # ---- runs
    def render(self, tape, img=None, debug=False, n_threads=0, count_flops=False, first_pass=0, pixels=None):
        """Add the passes of `tape` to img (n_pixels x 4 float32).  Returns (img, dump, flops).
        With a renderer in per-pixel random number mode the bounce part of the tape is None and
        `first_pass` is the global index of the tape's first pass.
        pixels: ray indices to walk instead of the whole frame (a pixel's path depends on its camera
        ray and the shared tape alone, or, with per-pixel random numbers, on its own ray index as
        well, which then keys its stream); img then has one row per index."""
        r = self.renderer
        passes, bounces, rows = tape
        full = self.renderer
        pixel_ids = None
        ray_index = getattr(full, 'ray_index', None)       # a tile: global index of every local ray
        if ray_index is not None:
            pixel_ids = np.ascontiguousarray(ray_index, dtype=np.int32)
        if pixels is not None:
            assert not debug
            cam = np.ascontiguousarray(r.host_cam[pixels], np.float32)
            pixels = np.asarray(pixels, dtype=np.int64)
            pixel_ids = np.ascontiguousarray(ray_index[pixels] if ray_index is not None else pixels, dtype=np.int32)
            r = type('Subset', (), dict(host_cam=cam, n_pixels=cam.shape[0], scene=full.scene))()
        rng = None
        if getattr(full, 'rng_mode', 'tape') == 'philox':
            lights = np.ascontiguousarray(full.light_table, np.float32)
            rng = Rng(full.rng_seed & 0xffffffff, (full.rng_seed >> 32) & 0xffffffff, int(first_pass),
                      full.ray_range[0] if full.ray_range else 0, full.n_lights, _fptr(lights),
                      int(getattr(full, 'pass_stride', 1)),
                      pixel_ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int)) if pixel_ids is not None else None)
            bounces = np.zeros(1, np.float32)
        n = passes.shape[0]
        if img is None:
            img = np.zeros((r.n_pixels, 4), dtype=np.float32)
        dump = None
        if debug:
            assert n == 1
            dump = np.zeros((r.scene.max_bounces, r.n_pixels, 6, 4), dtype=np.float32)
        passes = np.ascontiguousarray(passes, np.float32)
        bounces = np.ascontiguousarray(bounces, np.float32)
        rows_p = _fptr(np.ascontiguousarray(rows, np.float32)) if rows is not None else None
        flops = ctypes.c_double(0)
        lib = load(count_flops)
        rc = lib.orc_render(ctypes.byref(self.c_scene), _fptr(r.host_cam), _fptr(passes), _fptr(bounces),
                            rows_p, r.n_pixels, n, _fptr(img), _fptr(dump) if dump is not None else None,
                            n_threads, ctypes.byref(flops), ctypes.byref(rng) if rng is not None else None)
        assert rc == 0
        lib.orc_last_mesh_bytes.restype = ctypes.c_double
        self.last_mesh_bytes = lib.orc_last_mesh_bytes()
        return img, dump, flops.value

    def render_wavefront(self, tape, img=None, warp_size=1, n_threads=0):
        """The passes of `tape` (broadcast tape only) in the reference's execution structure: one
        sweep over the live rays per object and step, culler over groups of `warp_size` rays, stable
        compaction (orc_render_wavefront).  Returns (img, number of sweeps)."""
        r = self.renderer
        assert getattr(r, 'rng_mode', 'tape') == 'tape'
        passes, bounces, rows = tape
        if img is None:
            img = np.zeros((r.n_pixels, 4), dtype=np.float32)
        passes = np.ascontiguousarray(passes, np.float32)
        bounces = np.ascontiguousarray(bounces, np.float32)
        rows_p = _fptr(np.ascontiguousarray(rows, np.float32)) if rows is not None else None
        launches = ctypes.c_double(0)
        rc = load().orc_render_wavefront(ctypes.byref(self.c_scene), _fptr(r.host_cam), _fptr(passes), _fptr(bounces),
                                         rows_p, r.n_pixels, passes.shape[0], _fptr(img), int(warp_size), n_threads,
                                         ctypes.byref(launches))
        assert rc == 0
        return img, launches.value

    def first_hit(self, tape, n_threads=0, pixels=None):
        """pixels: ray indices to test instead of the whole frame (one row of results per index)."""
        r = self.renderer
        if pixels is not None:
            cam = np.ascontiguousarray(r.host_cam[pixels], np.float32)
            r = type('Subset', (), dict(host_cam=cam, n_pixels=cam.shape[0]))()
        passes = np.ascontiguousarray(tape[0], np.float32)
        ids = np.zeros((r.n_pixels, 2), dtype=np.int32)
        t = np.zeros(r.n_pixels, dtype=np.float32)
        normal = np.zeros((r.n_pixels, 4), dtype=np.float32)
        load().orc_first_hit(ctypes.byref(self.c_scene), _fptr(r.host_cam), _fptr(passes), r.n_pixels,
                             ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _fptr(t), _fptr(normal), n_threads)
        return ids, t, normal[:, :3]
