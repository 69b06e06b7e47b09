"""Finite cylinder and the infinite-cylinder component (reference objects/cylinder.py:5-39)."""
from ..utils import normalize_tuple
from ..transformations import Affine, rotation_aligning_vectors
from .convex_intersection import ConvexIntersection, FixedConvexIntersection
from .planes import ZLayerComponent


class CylinderComponent(ConvexIntersection.Component):
    n_subobjects = 1

    def __init__(self, axis, R):
        ConvexIntersection.Component.__init__(self)
        self.axis = normalize_tuple(axis)
        self.R = R

    def parameter_declarations(self):
        return ['float3 axis', 'float R']


class ZCylinderComponent(CylinderComponent):
    def __init__(self):
        CylinderComponent.__init__(self, (0, 0, 1), 1.0)

    def parameter_declarations(self):
        return []


class Cylinder(FixedConvexIntersection):
    """Unit z-slab ∩ unit z-cylinder, posed by rotation(axis) ∘ scale(R, R, height)."""

    def __init__(self, bottom_center, axis, height, R):
        self.axis = axis
        self.height = height
        self.R = R
        FixedConvexIntersection.__init__(self, bottom_center,
                                         [ZLayerComponent(), ZCylinderComponent()])

    def tracer_coordinate_system(self):
        rotation = Affine(linear=rotation_aligning_vectors((0, 0, 1), self.axis))
        return rotation(Affine(scaling=(self.R, self.R, self.height)))
