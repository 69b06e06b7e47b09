"""Intersections of convex components (reference objects/convex_intersection.py:5-120)."""
import numpy
from ..tracer import Tracer, next_unique_id


def float_literal(value):
    """A C float literal for a baked-in scalar."""
    text = repr(float(value))
    if 'e' not in text and '.' not in text and 'inf' not in text and 'nan' not in text:
        text += '.0'
    return text + 'f'


def vec_literal(vec):
    return "make_vec3(%s)" % ", ".join(float_literal(c) for c in numpy.ravel(vec))


class ConvexIntersection(Tracer):
    """Intersection of convex bodies; run-time parameters, one function per component list."""

    class Component(Tracer):
        def __init__(self, pos=(0, 0, 0)):
            Tracer.__init__(self, position=pos)

    def __init__(self, origin, components):
        Tracer.__init__(self, position=origin)
        self.components = components
        self.unique_tracer_id = '_' + '_'.join(c.__class__.__name__ for c in components)

    def template_name(self):
        return 'ConvexIntersection'

    def make_functions(self, template_env):
        funcs = dict(Tracer.make_functions(self, template_env))
        for component in self.components:
            funcs.update(component.make_functions(template_env))
        return funcs

    @property
    def convex(self):
        return True

    # ---- template helpers
    def component_position_expr(self, component_idx, base):
        return '%s - _pos_%d' % (base, component_idx)

    def component_parameter_string(self, component_idx):
        names = self.components[component_idx].parameter_names()
        return ''.join(', %s_%d' % (n, component_idx) for n in names)

    def component_tracer_call_params(self, component_idx):
        return '%s, ray, &c_enter, &c_exit, &c_face, inside%s' % (
            self.component_position_expr(component_idx, 'rel'),
            self.component_parameter_string(component_idx))

    def component_normal_call_params(self, component_idx, subobject_offset):
        return '%s, subobject - %d, p_normal%s' % (
            self.component_position_expr(component_idx, 'p'), subobject_offset,
            self.component_parameter_string(component_idx))

    # ---- run-time parameters: component position + the component's own parameters
    def parameter_declarations(self):
        decls = []
        for i, c in enumerate(self.components):
            decls.append('float3 _pos_%d' % i)
            decls += ['%s_%d' % (d, i) for d in c.parameter_declarations()]
        return decls

    def parameter_values(self):
        values = []
        for c in self.components:
            values.append(c.position)
            values += c.parameter_values()
        return values


class UniqueConvexIntersection(ConvexIntersection):
    """All component parameters baked into the generated source as literals."""

    def __init__(self, origin, components):
        ConvexIntersection.__init__(self, origin, components)
        self.unique_tracer_id = next_unique_id()

    def parameter_declarations(self):
        return []

    def parameter_values(self):
        return []

    def component_position_expr(self, component_idx, base):
        pos = self.components[component_idx].position
        if numpy.linalg.norm(pos) < 1e-10:
            return base
        return '%s - %s' % (base, vec_literal(pos))

    def component_parameter_string(self, component_idx):
        parts = []
        for val in self.components[component_idx].parameter_values():
            parts.append(vec_literal(val) if numpy.size(val) > 1 else float_literal(val))
        return ''.join(', ' + p for p in parts)


class FixedConvexIntersection(UniqueConvexIntersection):
    """Baked literals shared by every instance of the subclass (pose via the affine only)."""

    def __init__(self, *args, **kwargs):
        UniqueConvexIntersection.__init__(self, *args, **kwargs)
        self.unique_tracer_id = ''
