"""Randomly assembled scenes, HIP against the CPU oracle, bit for bit.

The shipped scenes place most bodies on the coordinate axes, which is where the renderer's
shortcuts apply (baked zero coefficients of the affine maps, operand-range speculation).  The
scenes here are drawn from a seeded generator instead: every shape family with a closed-form ray
test, rotated / stretched / sheared into general position, random materials (blur, volume
scattering, refraction), random cameras with and without depth of field, one or two lights.
Per scene the per-segment state dump of one pass and the image accumulated over further passes
must equal the oracle's in every bit, with the speculative arithmetic on and off.
"""
import numpy as np
import pytest

from raytracer_amd import objects as shapes
from raytracer_amd.renderer import Renderer
from raytracer_amd.scenes.library import BoxScene

pytestmark = pytest.mark.gpu

MATERIALS = ['white', 'green', 'red', 'mirror', 'glass', 'brushed metal', 'wax']


def random_axis(rng):
    v = rng.normal(size=3)
    return tuple(v / np.linalg.norm(v))


def random_body(rng, kind):
    center = rng.uniform((-2.0, -3.0, 0.3), (2.0, 3.0, 2.5))
    size = rng.uniform(0.25, 0.8)
    if kind == 'sphere':
        body = shapes.Sphere(tuple(center), size)
    elif kind == 'cylinder':
        body = shapes.Cylinder(tuple(center), random_axis(rng), rng.uniform(0.3, 1.2), size * 0.6)
    elif kind == 'cone':
        body = shapes.Cone(tuple(center), random_axis(rng), rng.uniform(0.4, 1.2), size * 0.7)
    elif kind == 'box':
        axes = np.diag(rng.uniform(0.3, 1.0, size=3)) + rng.uniform(-0.2, 0.2, size=(3, 3))
        body = shapes.Parallelepiped(tuple(center), *(tuple(a) for a in axes))
    elif kind == 'cylinders':
        body = shapes.ConvexIntersection(tuple(center), [shapes.CylinderComponent(random_axis(rng), size)
                                                         for _ in range(3)])
    elif kind == 'lens':
        offset = size * rng.uniform(0.2, 0.7)
        body = shapes.ConvexIntersection(tuple(center), [shapes.SphereComponent((0, 0, offset), size),
                                                         shapes.SphereComponent((0, 0, -offset), size)])
    else:
        body = getattr(shapes, kind)(tuple(center), size)
    pose = rng.integers(4)
    if pose == 1:
        body.rotate(random_axis(rng), rng.uniform(0, 360))
    elif pose == 2:
        body.linear_transform(scaling=tuple(rng.uniform(0.5, 2.0, size=3)))
    elif pose == 3:
        body.linear_transform(linear=np.identity(3) + rng.uniform(-0.4, 0.4, size=(3, 3)))
    return body


KINDS = ['sphere', 'cylinder', 'cone', 'box', 'cylinders', 'lens', 'Tetrahedron', 'Cube', 'Octahedron',
         'Dodecahedron', 'Icosahedron']


def random_scene(seed):
    rng = np.random.default_rng(seed)
    s = BoxScene()
    s.materials['glass']['transparency_blur'] = float(rng.choice([0.0, 0.2]))
    s.materials['fog'] = {'diffuse': 0.3, 'transparency': 0.6, 'volume_scattering': float(rng.uniform(0.3, 2.0)),
                          'volume_scattering_blur': float(rng.choice([1.0, 0.4])),
                          'volume_absorption': tuple(rng.uniform(0.0, 1.5, size=3))}
    s.get_object('light').bidirectional_light = bool(rng.integers(4) > 0)
    if rng.integers(2):
        lamp = shapes.Sphere(tuple(rng.uniform((-2, -3, 2.5), (2, 3, 3.5))), 0.3)
        s.add_object(lamp, 'light', name='lamp')
        s.get_object('lamp').bidirectional_light = True
    if rng.integers(2):
        s.add_object(shapes.HalfSpace(random_axis(rng), rng.uniform(3.0, 5.0)), 'sky')
    for kind in rng.choice(KINDS, size=rng.integers(3, 8)):
        s.add_object(random_body(rng, str(kind)), str(rng.choice(MATERIALS + ['fog'])))
    s.camera_position = tuple(rng.uniform((-2.5, -4.5, 0.5), (2.5, -2.0, 3.0)))
    s.direct_camera_towards(tuple(rng.uniform((-1, 0, 0), (1, 3, 2))))
    s.camera_fov = float(rng.uniform(35, 80))
    s.camera_flat_ccd = bool(rng.integers(2))
    s.camera_dof_fstop = float(rng.choice([0.0, 0.05]))
    s.min_bounces = int(rng.integers(1, 4))
    s.max_bounces = int(rng.integers(2, 7))
    s.min_russian_prob = float(rng.choice([0.0, 0.15, 0.5]))
    s.image.size = (48, 32)
    return s


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize('seed', range(101, 109))
def test_random_scene_bit_exact(seed):
    from oracle import oracle
    scene = random_scene(seed)
    r = Renderer(scene, seed=seed, speculate=True)
    osc = oracle.OracleScene(r)
    first, rest = r.make_tape(0, 1), r.make_tape(1, 5)
    _, dump = r.render_tape(first, debug=True)
    oimg, odump, _ = osc.render(first, debug=True)
    assert np.array_equal(bits(dump), bits(odump))
    r.render_tape(rest)
    oimg, _, _ = osc.render(rest, img=oimg)
    img = r.img.get()
    assert np.isfinite(img).all() and img[:, :3].max() > 0
    assert np.array_equal(bits(img), bits(oimg))

    generic = Renderer(random_scene(seed), seed=seed, speculate=False, bake_parameters=False, split_passes=False)
    generic.render_tape(first)
    generic.render_tape(rest)
    assert np.array_equal(bits(generic.img.get()), bits(oimg))


@pytest.mark.parametrize('seed', range(201, 209))
def test_random_scene_with_per_pixel_random_numbers_bit_exact(seed):
    """The same generator in the per-pixel random number mode, where every pixel takes its own
    path: far more distinct (shape, ray) pairs per frame than the shared tape gives."""
    from oracle import oracle
    r = Renderer(random_scene(seed), seed=seed, rng_mode='philox')
    osc = oracle.OracleScene(r)
    first, rest = r.make_tape(0, 1), r.make_tape(1, 7)
    _, dump = r.render_tape(first, debug=True)
    oimg, odump, _ = osc.render(first, debug=True)
    assert np.array_equal(bits(dump), bits(odump))
    r.render_tape(rest, first_pass=1)
    oimg, _, _ = osc.render(rest, img=oimg, first_pass=1)
    img = r.img.get()
    assert np.isfinite(img).all() and img[:, :3].max() > 0
    assert np.array_equal(bits(img), bits(oimg))
