"""Convex polyhedra as intersections of half-spaces / slabs (reference objects/polyhedra.py:5-102)."""
import math
from ..utils import vec_norm
from .convex_intersection import ConvexIntersection, FixedConvexIntersection
from .planes import HalfSpaceComponent, LayerComponent

PHI = 0.5 * (1 + math.sqrt(5))  # golden ratio


class Parallelepiped(ConvexIntersection):
    def __init__(self, origin, ax1, ax2, ax3):
        ConvexIntersection.__init__(self, origin, [LayerComponent(ax) for ax in (ax1, ax2, ax3)])
        self.unique_tracer_id = ''


class HyperplaneRepresentation(FixedConvexIntersection):
    """Polyhedron given by outward plane vectors p (plane at distance |p| along p)."""

    def __init__(self, center, planes, R):
        FixedConvexIntersection.__init__(
            self, center, [HalfSpaceComponent(p, vec_norm(p)) for p in planes])
        self.linear_transform(scaling=R)


class SymmetricDualPolyhedron(FixedConvexIntersection):
    """Dual of the point-symmetric vertex set {±v}: one slab per listed vertex."""

    def __init__(self, center, vertices, R):
        slabs = []
        for v in vertices:
            slab = LayerComponent(tuple(-x for x in v), vec_norm(v) * 2.0)
            slab.position = v
            slabs.append(slab)
        FixedConvexIntersection.__init__(self, center, slabs)
        self.linear_transform(scaling=R)


class Tetrahedron(HyperplaneRepresentation):
    def __init__(self, center, R):
        q = 1.0 / math.sqrt(2)
        HyperplaneRepresentation.__init__(
            self, center, [(1, 0, -q), (-1, 0, -q), (0, 1, q), (0, -1, q)], R / (q + 1 / q))


class Cube(SymmetricDualPolyhedron):
    def __init__(self, center, side=None, R=None):
        if side is None:
            # as in the reference, calling with neither argument is a TypeError (R is None)
            side = R / math.sqrt(3) * 2.0
        SymmetricDualPolyhedron.__init__(
            self, center, [(1, 0, 0), (0, 1, 0), (0, 0, 1)], side / 2.0)


class Octahedron(SymmetricDualPolyhedron):
    def __init__(self, center, R):
        SymmetricDualPolyhedron.__init__(
            self, center, [(1, 1, 1), (1, -1, 1), (-1, 1, 1), (-1, -1, 1)], R / 3.0)


class Dodecahedron(SymmetricDualPolyhedron):
    def __init__(self, center, R):
        SymmetricDualPolyhedron.__init__(
            self, center,
            [(0, 1, PHI), (1, PHI, 0), (PHI, 0, 1), (0, -1, PHI), (-1, PHI, 0), (PHI, 0, -1)],
            R / (PHI + 1.0 / PHI))


class Icosahedron(SymmetricDualPolyhedron):
    def __init__(self, center, R):
        q = 1 / PHI
        SymmetricDualPolyhedron.__init__(
            self, center,
            [(1, 1, 1), (-1, 1, 1), (1, -1, 1), (-1, -1, 1), (0, q, PHI), (0, -q, PHI),
             (q, PHI, 0), (-q, PHI, 0), (PHI, 0, q), (PHI, 0, -q)],
            R / (PHI + 1 / PHI ** 3))
