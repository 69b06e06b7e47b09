"""Sphere shape, also usable as a next-event-estimation light (reference objects/sphere.py:6-49)."""
import numpy
from ..tracer import Tracer
from ..transformations import Affine
from .convex_intersection import ConvexIntersection


class Sphere(Tracer):

    def __init__(self, pos, R):
        Tracer.__init__(self, position=pos)
        self.R = R

    def tracer_coordinate_system(self):
        return Affine(scaling=self.R)

    @property
    def convex(self):
        return True

    # ---- light interface (only Sphere implements it, as in the reference)
    def _check_no_distortion(self):
        if not self.coordinates.is_orthogonal():
            raise RuntimeError("illegal operation for scaled sphere")

    def surface_area(self):
        self._check_no_distortion()
        return 4.0 * numpy.pi * self.R ** 2

    def random_surface_point_and_normal(self, rng=numpy.random):
        self._check_no_distortion()
        center = numpy.array(self.position)
        direction = rng.normal(0, 1, center.shape)
        direction = direction / numpy.linalg.norm(direction)
        return center + direction * self.R, direction

    def center_and_min_sampling_distance(self):
        self._check_no_distortion()
        return numpy.array(self.position), self.R * 2.0


class SphereComponent(ConvexIntersection.Component):
    n_subobjects = 1

    def __init__(self, pos, R):
        ConvexIntersection.Component.__init__(self, pos)
        self.R = R

    def parameter_declarations(self):
        return ['float R']
