"""Ray-marched shapes: ImplicitSurface (sympy-generated interval arithmetic), the two quaternion
Julia sets and DistanceField (user code string).  Same constructors as the reference's
objects/implicit_surface.py:6-131, julia_sets.py:7-41 and distance_field.py:3-17.
The equation of an ImplicitSurface must be positive outside the body; a DistanceField's
``tracer_code`` assigns ``dist`` from the local coordinates ``x, y, z``."""
from .base import Tracer, next_unique_id
from .convex import float_literal


def _printers():
    import sympy
    from sympy.printing.str import StrPrinter

    class FloatPrinter(StrPrinter):
        """Plain C expression; integer powers become repeated products."""

        def _print_Float(self, expr):
            return "%rf" % float(expr)

        def _print_Pow(self, expr):
            base, exponent = expr.args
            if exponent.is_integer and int(exponent) > 0:
                factor = "(" + self._print(base) + ")"
                return "(" + "*".join([factor] * int(exponent)) + ")"
            return "powf(%s,%s)" % (self._print(base), self._print(exponent))

    class IntervalPrinter(StrPrinter):
        """Expression over ia_type operands: ia_powN / ia_mul / ia_mul_{pos,neg}_exact."""

        def _print_Float(self, expr):
            return "%gf" % expr

        def _print_Pow(self, expr):
            base, exponent = expr.args
            return "ia_pow%d(%s)" % (int(exponent), self._print(base))

        def _mul_chain(self, args):
            a, b = args[0], args[1]
            if b.is_number:
                a, b = b, a
            rest = self._mul_chain(args[1:]) if len(args) > 2 else self._print(b)
            if a.is_number:
                if b.is_number:
                    return "((%s)*(%s))" % (self._print(a), self._print(b))
                if a >= 0:
                    return "ia_mul_pos_exact(%s,%s)" % (rest, self._print(a))
                return "ia_mul_neg_exact(%s,%s)" % (rest, self._print(a))
            return "ia_mul(%s,%s)" % (self._print(a), rest)

        def _print_Mul(self, expr):
            return self._mul_chain(expr.args)

    return sympy, FloatPrinter(), IntervalPrinter()


class ImplicitSurface(Tracer):
    outline_candidate = True

    def __init__(self, eq, center=(0, 0, 0), scale=1.0, bndR=None,
                 max_itr=1500, precision=0.001, self_intersection=True):
        Tracer.__init__(self, position=center, scaling=scale)
        self.unique_tracer_id = next_unique_id()
        self.center = tuple(center)
        self.no_self_intersection = not self_intersection
        self.precision = precision
        self.max_itr = max_itr
        self.bndR = bndR

        sympy, float_printer, interval_printer = _printers()
        x, y, z = xyz = sympy.symbols('x y z')
        t = sympy.symbols('t')
        ray = sympy.symbols('ray_x ray_y ray_z')
        origin = sympy.symbols('origin_x origin_y origin_z')
        pos = sympy.symbols('pos.x pos.y pos.z')

        self.eq = sympy.sympify(eq)
        self.ray_paramd = self.eq.subs([(xyz[i], ray[i] * t + origin[i]) for i in range(3)])
        pos_eq = self.eq.subs([(xyz[i], pos[i]) for i in range(3)])
        self.gradient = [sympy.diff(pos_eq, pos[i]) for i in range(3)]
        self.derivative = sympy.diff(self.ray_paramd, t)

        self._interval_printer = interval_printer
        self.f_code = self.compute_f_code()
        self.df_code = self.compute_df_code()
        self.gradient_code = [float_printer.doprint(g) for g in self.gradient]

    def template_name(self):
        return 'ImplicitSurface'

    def f_code_template(self):
        """Key of a hand-written interval function in shapes/marching.hip that replaces the
        sympy-printed one (None = use f_code / df_code)."""
        return None

    def compute_f_code(self):
        return "\n            f = %s;" % self._interval_printer.doprint(self.eq)

    def compute_df_code(self):
        return "\n            df = %s; " % self._interval_printer.doprint(self.derivative)


class QuaternionJuliaSet(ImplicitSurface):
    """Interval-arithmetic root isolation on f = |z_n|^2 - 4, z -> z^2 + c."""

    def __init__(self, c, julia_itr, *args, **argd):
        self.c = c
        self.julia_itr = julia_itr
        argd['bndR'] = 1.5
        argd['self_intersection'] = False
        ImplicitSurface.__init__(self, "x^2 + y^2 + z^2 - 1", *args, **argd)

    def f_code_template(self):
        return 'QuaternionJuliaSet'

    def compute_f_code(self):
        self.c_code = [float_literal(x) for x in self.c]

    def compute_df_code(self):
        pass


class QuaternionJuliaSet2(Tracer):
    outline_candidate = True
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


class DistanceField(Tracer):
    outline_candidate = True

    def __init__(self, tracer_code, normal_code=None, center=(0, 0, 0), bndR=None,
                 max_itr=1000, precision=1e-5, self_intersection=True):
        Tracer.__init__(self, position=center)
        self.max_itr = max_itr
        self.precision = precision
        self.bndR = bndR
        self.no_self_intersection = not self_intersection
        self.tracer_code = tracer_code.strip()
        self.normal_code = normal_code
        self.unique_tracer_id = next_unique_id()
