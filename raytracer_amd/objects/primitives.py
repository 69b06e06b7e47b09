"""Stand-alone analytic shapes: Sphere (also the only next-event light shape) and HalfSpace.
Constructors as in the reference's objects/sphere.py:6-37 and objects/planes.py:7-21."""
import numpy
from .base import Tracer
from ..geometry import normalize_tuple
from ..geometry import Affine, rotation_aligning_vectors


class Sphere(Tracer):
    speculation_friendly = True
    speculative_normal = True
    # most rays of a frame miss a given sphere: they are told apart before the ray is normalised
    # (a square root, a reciprocal, three quotients), with margins that cover the roundings of
    # the full test (static.hip, clr_unit_sphere_may_be_hit)
    can_hit_expression = '(within || clr_unit_sphere_may_be_hit(local_pos, local_ray))'

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
        # asked per light point, i.e. per path segment of every pass: checked once per pose
        # (every transformation installs a new Affine)
        if getattr(self, '_undistorted_pose', None) is self.coordinates:
            return
        if not self.coordinates.is_orthogonal():
            raise RuntimeError("illegal operation for scaled sphere")
        self._undistorted_pose = self.coordinates

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


class HalfSpace(Tracer):
    """{x : normal.x <= h}; in tracer coordinates the boundary is the plane x = 0."""
    speculation_friendly = True
    # dist = -origin.x / (ray.x / |ray|) is positive only for operands of strictly opposite signs:
    # the sign of a quotient is exact, a zero or NaN operand gives a zero, an infinity (never below
    # the current distance) or a NaN.  About half of the walls of a room drop out this way, and
    # neighbouring rays agree on which, so whole waves skip the test.
    # (bitwise operators: four compares and three mask operations, no short-circuit branches)
    # Second condition: the plane is not certainly beyond the distance that counts - the closest
    # hit so far, or the light point (static.hip, clr_plane_before).
    can_hit_expression = ('(((local_pos.x > 0.0f) & (local_ray.x < 0.0f)) | '
                          '((local_pos.x < 0.0f) & (local_ray.x > 0.0f))) & '
                          'clr_plane_before(local_pos.x, local_ray.x, isec_dist)')

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
