"""Base class of all shapes ("tracers") and the template hooks the code generator calls.

Keeps the interface of /root/reference/tracer.py:4-154 (naming scheme, parameter
introspection, ``make_functions``/``make_kernels(template_env) -> {name: source}``,
``convex`` / ``auto_flip_normal`` flags, ``linear_transform``/``rotate``/``position``).
The templates rendered here are HIP (``hip_templates/shapes/<family>.hip``), and the
"kernels" of a class are ``__device__`` step functions that the fused bounce kernel
calls instead of separate launches.
"""
import itertools
import numpy
from ..geometry import Affine

_instance_counter = itertools.count(1)


def next_unique_id():
    """Stable replacement for the reference's ``id(self)`` suffix."""
    return "%04d" % next(_instance_counter)


# template family of every shipped shape class (hip_templates/shapes/<family>.hip)
SHAPE_FAMILY = dict(
    [(name, 'quadrics') for name in ('Sphere', 'HalfSpace')]
    + [(name, 'convex') for name in (
        'ConvexIntersection', 'SphereComponent', 'HalfSpaceComponent', 'LayerComponent',
        'ZLayerComponent', 'CylinderComponent', 'ZCylinderComponent', 'ConeComponent', 'ZConeComponent')]
    + [(name, 'marching') for name in ('ImplicitSurface', 'QuaternionJuliaSet2', 'DistanceField')]
    + [(name, 'meshes') for name in ('TriangleMesh', 'Octree')])

# scalar type names used in parameter declarations -> HIP type
HIP_TYPES = {'float3': 'vec3', 'float': 'float', 'int': 'int'}


class _GeneratedName(object):
    """Read-only attribute giving the name of one generated function of a shape."""

    def __init__(self, role):
        self.role = role

    def __get__(self, shape, owner=None):
        return self if shape is None else shape.generated_name(self.role)


def pack_slot(type_name, value):
    """One float4 slot of the device parameter table: float3 in xyz, float or bit-cast int in x."""
    slot = numpy.zeros(4, dtype=numpy.float32)
    if type_name == 'float3':
        slot[:3] = numpy.ravel(value)
    elif type_name == 'int':
        slot[:1] = numpy.array([value], dtype=numpy.int32).view(numpy.float32)
    elif type_name == 'float':
        slot[0] = value
    else:
        raise RuntimeError('unsupported parameter type %s' % type_name)
    return slot


class Tracer(object):
    """A three-dimensional body with a ray-intersection and a normal function."""

    # True for shapes whose ray test is a few divisions / square roots: the renderer then uses
    # the speculative exact arithmetic of hip_templates/clr_exact.inl for the scene
    speculation_friendly = False

    # Optional HIP expression over `local_pos` and `local_ray` (tracer coordinates, ray not yet
    # normalised).  Where it is false the class's ray test is KNOWN to report no hit, and the
    # step functions skip the normalisation and the test.  It must be exact: false only if the
    # full IEEE evaluation also ends without a hit (true is always allowed).
    can_hit_expression = None

    # True: the step that normalises the world-space normal uses the guard-aware short sequences
    # (clr_make_direction).  Only for shapes whose normal has no structurally zero component - a
    # zero component of a direction fails the guard and the path is walked again.
    speculative_normal = False

    # True for shapes whose ray test is a long loop with many live values (ray marchers, mesh
    # traversals).  In scenes with many shape classes the renderer then generates the test as a
    # function of its own instead of inlining it into both object sweeps (`outlined`, set by the
    # renderer): one copy of the code, and its registers are live only while it runs.
    outline_candidate = False
    outlined = False

    def __init__(self, position=(0, 0, 0), **affine_kwargs):
        # non-empty => one generated function per instance instead of one per class
        self.unique_tracer_id = ""
        self.coordinates = Affine(translation=position, **affine_kwargs)

    # ---- naming scheme (reference tracer.py:19-40): <Class><unique id>_<role>
    def generated_name(self, role):
        return "%s%s_%s" % (self.__class__.__name__, self.unique_tracer_id, role)

    tracer_function_name = _GeneratedName('tracer')
    normal_function_name = _GeneratedName('normal')
    tracer_kernel_name = _GeneratedName('tracer_kernel')
    shadow_kernel_name = _GeneratedName('shadow_kernel')
    normal_kernel_name = _GeneratedName('normal_kernel')

    # ---- template hooks (role of reference tracer.py:42-79)
    def template_name(self):
        """Key of the macro pair ``<name>__tracer`` / ``<name>__normal`` that renders this class."""
        return self.__class__.__name__

    def template_file_name(self):
        """Template file holding that macro pair.  The shipped shapes live in four family files
        under hip_templates/shapes/; a user-defined Tracer overrides this method."""
        return 'shapes/%s.hip' % SHAPE_FAMILY[self.template_name()]

    def _make_code(self, template_file, macro, template_env):
        text = "{%% import '%s' as a with context %%}{{ a.%s(obj) }}\n" % (template_file, macro)
        return template_env.from_string(text).render(obj=self)

    def make_functions(self, template_env):
        """HIP ``__device__`` functions of this class: {function name: source}."""
        source, key = self.template_file_name(), self.template_name()
        return {
            self.tracer_function_name: self._make_code(source, key + '__tracer', template_env),
            self.normal_function_name: self._make_code(source, key + '__normal', template_env),
        }

    def make_kernels(self, template_env):
        """Per-class step functions (closest hit, shadow test, normal): {name: source}."""
        return {
            self.tracer_kernel_name: self._make_code('tracer.hip', 'tracer_kernel', template_env),
            self.shadow_kernel_name: self._make_code('tracer.hip', 'shadow_kernel', template_env),
            self.normal_kernel_name: self._make_code('tracer.hip', 'normal_kernel', template_env),
        }

    # ---- parameters (reference tracer.py:81-108)
    def parameter_declarations(self):
        return []

    def _typed_parameters(self):
        for decl in self.parameter_declarations():
            type_name, name = decl.split()
            yield type_name, name

    def parameter_types(self):
        return [t for t, _ in self._typed_parameters()]

    def parameter_names(self):
        return [n for _, n in self._typed_parameters()]

    def parameter_values(self):
        return [getattr(self, n) for n in self.parameter_names()]

    def parameter_declaration_string(self):
        params = ["%s %s" % (HIP_TYPES[t], n) for t, n in self._typed_parameters()]
        if self.has_data():
            params = ["const float4 * __restrict__ vector_data",
                      "const int * __restrict__ integer_data"] + params
        return hip_parameter_string(params)

    def has_data(self):
        return hasattr(self, 'get_data')

    def parameter_slots(self):
        """Rows of this shape in the device parameter table (k x 4 float32): the declared
        parameters, then the three rows and the translation of the INVERSE of
        global_to_tracer_coordinate_transform() (replaces the per-launch scalar and float3
        kernel arguments of reference renderer.py:191-241)."""
        rows = [pack_slot(t, v) for t, v in zip(self.parameter_types(), self.parameter_values())]
        to_local = self.global_to_tracer_coordinate_transform().inverse()
        for vec in (to_local.linear[0, :], to_local.linear[1, :], to_local.linear[2, :],
                    to_local.translation):
            rows.append(pack_slot('float3', vec))
        return numpy.array(rows, dtype=numpy.float32)

    # ---- coordinates (reference tracer.py:110-150)
    def tracer_coordinate_system(self):
        return Affine.identity()

    def global_to_tracer_coordinate_transform(self):
        return self.coordinates(self.tracer_coordinate_system())

    @property
    def auto_flip_normal(self):
        return False

    @property
    def convex(self):
        return False

    def linear_transform(self, linear=None, translation=None, **kwargs):
        if translation is not None:
            raise RuntimeError('translation not allowed')
        extra = Affine(linear=linear, **kwargs).linear
        self.coordinates = Affine(linear=numpy.dot(extra, self.coordinates.linear),
                                  translation=self.coordinates.translation)
        return self

    def rotate(self, axis, deg):
        return self.linear_transform(rotation_axis=axis, rotation_deg=deg)

    @property
    def position(self):
        return self.coordinates.translation

    @position.setter
    def position(self, value):
        self.coordinates = Affine(linear=self.coordinates.linear, translation=value)


def hip_parameter_string(params):
    return ''.join(', ' + p for p in params)
