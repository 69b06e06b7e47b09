"""Finite cone and the semi-infinite cone component (reference objects/cone.py:5-43)."""
from ..utils import normalize_tuple
from ..transformations import Affine, rotation_aligning_vectors
from .convex_intersection import ConvexIntersection, FixedConvexIntersection
from .planes import HalfSpaceComponent


class ConeComponent(ConvexIntersection.Component):
    n_subobjects = 1

    def __init__(self, pos, axis, slope):
        ConvexIntersection.Component.__init__(self, pos)
        self.axis = normalize_tuple(axis)
        self.slope = slope

    def parameter_declarations(self):
        return ['float3 axis', 'float slope']


class ZConeComponent(ConeComponent):
    def __init__(self):
        ConeComponent.__init__(self, (0, 0, 0), (0, 0, 1), 1.0)

    def parameter_declarations(self):
        return []


class Cone(FixedConvexIntersection):
    """{z <= 1} ∩ unit z-cone with its tip at the origin, posed like Cylinder."""

    def __init__(self, tip, axis, height, R):
        self.axis = axis
        self.height = height
        self.R = R
        FixedConvexIntersection.__init__(self, tip, [
            HalfSpaceComponent(normal=(0, 0, 1), h=1.0), ZConeComponent()])

    def tracer_coordinate_system(self):
        rotation = Affine(linear=rotation_aligning_vectors((0, 0, 1), self.axis))
        return rotation(Affine(scaling=(self.R, self.R, self.height)))
