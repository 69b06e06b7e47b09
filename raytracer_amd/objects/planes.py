"""Half-space and slab shapes (reference objects/planes.py:7-54)."""
import numpy
from ..tracer import Tracer
from ..utils import normalize_tuple, vec_norm
from ..transformations import Affine, rotation_aligning_vectors
from .convex_intersection import ConvexIntersection


class HalfSpace(Tracer):
    """{x : normal.x <= h}; in tracer coordinates the boundary is the plane x = 0."""

    def __init__(self, normal, h):
        Tracer.__init__(self)
        self.normal = normalize_tuple(normal)
        self.h = h

    def tracer_coordinate_system(self):
        return Affine(linear=rotation_aligning_vectors((1, 0, 0), self.normal),
                      translation=-numpy.ravel(self.normal) * self.h)

    @property
    def convex(self):
        return True


class HalfSpaceComponent(ConvexIntersection.Component):
    n_subobjects = 1

    def __init__(self, normal, h):
        ConvexIntersection.Component.__init__(self)
        self.normal = normalize_tuple(normal)
        self.h = h

    def parameter_declarations(self):
        return ['float3 normal', 'float h']


class LayerComponent(HalfSpaceComponent):
    """Infinite slab of thickness h along ``axis`` (h defaults to |axis|)."""
    n_subobjects = 2

    def __init__(self, axis, h=None):
        HalfSpaceComponent.__init__(self, axis, vec_norm(axis) if h is None else h)


class ZLayerComponent(LayerComponent):
    def __init__(self):
        LayerComponent.__init__(self, (0, 0, 1), 1.0)

    def parameter_declarations(self):
        return []
