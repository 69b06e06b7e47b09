"""Implicit surfaces f(x,y,z) = 0 traced with interval arithmetic.

Same constructor and code-generation role as /root/reference/objects/implicit_surface.py:6-131:
sympy differentiates the equation and two printers turn it into (a) an interval-arithmetic
expression in the ``ia_*`` helpers for f and df/dt along the ray and (b) plain float
expressions for the gradient.  The equation must be positive outside the body.
"""
from ..tracer import Tracer, next_unique_id


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
        return None

    def compute_f_code(self):
        return "\n            f = %s;" % self._interval_printer.doprint(self.eq)

    def compute_df_code(self):
        return "\n            df = %s; " % self._interval_printer.doprint(self.derivative)
