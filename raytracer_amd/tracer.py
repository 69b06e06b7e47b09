"""Base class of all shapes ("tracers") and the template hooks the code generator calls.

Keeps the interface of /root/reference/tracer.py:4-154 (naming scheme, parameter
introspection, ``make_functions``/``make_kernels(template_env) -> {name: source}``,
``convex`` / ``auto_flip_normal`` flags, ``linear_transform``/``rotate``/``position``).
The templates rendered here are HIP (``hip_templates/objects/<Class>.hip``), and the
"kernels" of a class are ``__device__`` step functions that the fused bounce kernel
calls instead of separate launches.
"""
import itertools
import numpy
from .transformations import Affine

_instance_counter = itertools.count(1)


def next_unique_id():
    """Stable replacement for the reference's ``id(self)`` suffix."""
    return "%04d" % next(_instance_counter)


# scalar type names used in parameter declarations -> HIP type
HIP_TYPES = {'float3': 'vec3', 'float': 'float', 'int': 'int'}


class Tracer(object):
    """A three-dimensional body with a ray-intersection and a normal function."""

    def __init__(self, position=(0, 0, 0), **affine_kwargs):
        # non-empty => one generated function per instance instead of one per class
        self.unique_tracer_id = ""
        self.coordinates = Affine(translation=position, **affine_kwargs)

    # ---- naming scheme (reference tracer.py:19-40)
    def _function_name_prefix(self):
        return self.__class__.__name__ + self.unique_tracer_id

    @property
    def tracer_function_name(self):
        return self._function_name_prefix() + '_tracer'

    @property
    def normal_function_name(self):
        return self._function_name_prefix() + '_normal'

    @property
    def tracer_kernel_name(self):
        return self._function_name_prefix() + '_tracer_kernel'

    @property
    def shadow_kernel_name(self):
        return self._function_name_prefix() + '_shadow_kernel'

    @property
    def normal_kernel_name(self):
        return self._function_name_prefix() + '_normal_kernel'

    # ---- template hooks (reference tracer.py:42-79)
    def template_name(self):
        return self.__class__.__name__

    def template_file_name(self):
        return 'objects/%s.hip' % self.template_name()

    def _make_code(self, macro, template_env):
        text = "{%% import '%s' as a with context %%}{{ a.%s }}\n" % (self.template_file_name(), macro)
        return template_env.from_string(text).render(obj=self)

    def make_functions(self, template_env):
        """HIP ``__device__`` functions of this class: {function name: source}."""
        return {
            self.tracer_function_name: self._make_code('tracer_function(obj)', template_env),
            self.normal_function_name: self._make_code('normal_function(obj)', template_env),
        }

    def make_kernels(self, template_env):
        """Per-class step functions (closest hit, shadow test, normal): {name: source}."""
        return {
            self.tracer_kernel_name: self._make_code('tracer_kernel(obj)', template_env),
            self.shadow_kernel_name: self._make_code('shadow_kernel(obj)', template_env),
            self.normal_kernel_name: self._make_code('normal_kernel(obj)', template_env),
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
