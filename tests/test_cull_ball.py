"""objects/convex.py: plane_body_cull_ball - the ball around a body bounded by planes that the
device test clr_ball_may_be_hit (hip_templates/static.hip) drops rays against.  CPU: the ball holds
the body, unbounded or thin bodies get none, and a binary64 model of the test never drops a ray
that passes through the body."""
import numpy as np

from raytracer_amd.objects import convex
from raytracer_amd.objects.convex import (CULL_BALL_FACTOR, CULL_BALL_SLACK, HalfSpaceComponent, LayerComponent,
                                          plane_body_cull_ball)


def planes_of(components):
    planes = []
    for c in components:
        n = np.ravel(np.asarray(c.normal, dtype=float))
        offset = float(np.dot(np.ravel(np.asarray(c.position, dtype=float)), n))
        planes.append((n, offset + c.h))
        if isinstance(c, LayerComponent):
            planes.append((-n, -offset))
    return np.array([n for n, _ in planes]), np.array([d for _, d in planes])


def test_the_ball_holds_the_body_and_no_ray_through_the_body_is_dropped():
    rng = np.random.RandomState(5)
    for cls in (convex.Dodecahedron, convex.Cube, convex.Octahedron, convex.Icosahedron, convex.Tetrahedron):
        body = cls((0, 0, 0), 1.0)
        centre, rho2 = plane_body_cull_ball(body.components)
        assert 'clr_ball_may_be_hit' in body.can_hit_expression
        normals, heights = planes_of(body.components)
        # points of the body: rejection sampling in the ball
        radius = np.sqrt(rho2 / CULL_BALL_SLACK) / CULL_BALL_FACTOR
        points = centre + radius * rng.uniform(-1, 1, size=(20000, 3))
        inside = (points.dot(normals.T) <= heights).all(axis=1)
        assert inside.sum() > 500
        assert (np.linalg.norm(points[inside] - centre, axis=1) <= radius).all()
        # and nothing of the body lies outside that ball: directions scaled to its surface
        far = centre + radius * 1.0001 * (lambda v: v / np.linalg.norm(v, axis=1)[:, None])(rng.normal(size=(20000, 3)))
        assert not (far.dot(normals.T) <= heights).all(axis=1).any()
        # the model of the device test: rays from outside through points of the body are kept
        origins = centre + radius * rng.uniform(2, 50) * (lambda v: v / np.linalg.norm(v, axis=1)[:, None])(rng.normal(size=(inside.sum(), 3)))
        d = (points[inside] - origins) * rng.uniform(0.1, 10, size=(inside.sum(), 1))
        o = origins - centre
        od, oo, dd = (o * d).sum(1), (o * o).sum(1), (d * d).sum(1)
        dropped = (oo > rho2) & (((od > 0) & (od * od > 2.0 ** -36 * dd * oo)) | (oo * dd - od * od > rho2 * dd))
        assert not dropped.any()


def test_bodies_without_a_ball():
    slab = lambda axis: LayerComponent(axis, 1.0)
    # no vertices at all, an open prism, a body closed by a fourth plane
    assert plane_body_cull_ball([slab((1, 0, 0)), slab((0, 1, 0))]) is None
    assert plane_body_cull_ball([slab((1, 0, 0)), slab((0, 1, 0)), HalfSpaceComponent((0, 0, 1), 1.0)]) is None
    assert plane_body_cull_ball([slab((1, 0, 0)), slab((0, 1, 0)), HalfSpaceComponent((0, 0, 1), 1.0),
                                 HalfSpaceComponent((0.1, 0, -1), 1.0)]) is not None
    # a plate 100 x thinner than it is wide: the centre is too close to its faces for the proof's margin
    thin = [LayerComponent((1, 0, 0), 1.0), LayerComponent((0, 1, 0), 1.0), LayerComponent((0, 0, 1), 0.01)]
    assert plane_body_cull_ball(thin) is None
    # other components: no statement
    assert plane_body_cull_ball([slab((1, 0, 0)), convex.SphereComponent((0, 0, 0), 1.0)]) is None
