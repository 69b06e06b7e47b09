"""Quaternion Julia sets (reference objects/julia_sets.py:7-41)."""
from ..tracer import Tracer
from .implicit_surface import ImplicitSurface
from .convex_intersection import float_literal


class QuaternionJuliaSet(ImplicitSurface):
    """Interval-arithmetic root isolation on f = |z_n|^2 - 4, z -> z^2 + c."""

    def __init__(self, c, julia_itr, *args, **argd):
        self.c = c
        self.julia_itr = julia_itr
        argd['bndR'] = 1.5
        argd['self_intersection'] = False
        ImplicitSurface.__init__(self, "x^2 + y^2 + z^2 - 1", *args, **argd)

    def f_code_template(self):
        return 'objects/QuaternionJuliaSet.hip'

    def compute_f_code(self):
        self.c_code = [float_literal(x) for x in self.c]

    def compute_df_code(self):
        pass


class QuaternionJuliaSet2(Tracer):
    """Distance-estimator marching (Hart et al. 1989)."""

    def __init__(self, c, julia_itr, center=(0, 0, 0), scale=1.0, **kwargs):
        Tracer.__init__(self, position=center, scaling=scale)
        self.c = c
        self.julia_itr = julia_itr
        self.bndR = 4.0
        self.max_itr = 100
        self.precision = 0.001
        self.max_step = 0.1
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def c_as_float4(self):
        return "make_vec4(%s)" % ", ".join(float_literal(x) for x in self.c)
