"""Scene library: default settings, the box room, and one builder function per shipped scene.

The scene files next to this module (scene-dev.py, scene-test.py, ...) are two-line shims that
call a builder, so that `python clray.py raytracer_amd/scenes/scene-dev.py` works like the
reference's command line.  Object and camera values are those of the reference's scene files
(/root/reference/scenes/default_scenes.py:80-148, scene-dev.py:6-30, scene-three-objects.py:7-64,
scene-test.py:12-109, scene-stairs.py:8-60); scene_marchers and scene_mesh are the BASELINE
configurations C5 and C4, which the reference has no file for.
"""
import io
import math

import numpy
import numpy as np

from ..scene import Scene
from .. import shading as shader
from ..shading import (Spectrum, RgbShader, SpectrumShader, default_rgb_materials,
                       default_spectrum_materials)
from ..geometry import Affine, vec_norm
from ..objects import *  # noqa: F401,F403
from ..objects import meshes as mesh_formats


# ================================================================ defaults and the box room
def default_settings(scene):
    scene.image = Scene.ImageSettings()
    scene.image.size = (800, 600)

    scene.tent_filter = True
    scene.quasirandom = False
    scene.samples_per_pixel = 10000
    scene.min_bounces = 2      # really: max diffuse bounces
    scene.max_bounces = 4
    scene.min_russian_prob = 0.15

    scene.camera_up = (0, 0, 1)
    scene.camera_fov = 55
    scene.camera_flat_ccd = False
    scene.camera_dof_fstop = 0.0

    scene.root_object = None
    scene.max_ray_length = 1000


def make_world_box(scene, material, dims, center=(0, 0, 0)):
    walls = [((1, 0, 0), dims[0] - center[0], 'wall'), ((-1, 0, 0), dims[0] + center[0], 'wall'),
             ((0, 1, 0), dims[1] - center[1], 'wall'), ((0, -1, 0), dims[1] + center[1], 'wall'),
             ((0, 0, 1), dims[2] - center[2], 'floor'), ((0, 0, -1), dims[2] + center[2], 'ceiling')]
    for normal, h, name in walls:
        scene.add_object(HalfSpace(normal, h), material, name)


class DefaultScene(Scene):

    def __init__(self, shader_class=shader.RgbShader):
        default_settings(self)
        self.shader = shader_class
        if shader_class == shader.SpectrumShader:
            self.spectrum = Spectrum()
            self.materials = default_spectrum_materials(self.spectrum)
        else:
            self.materials = default_rgb_materials()
        self.objects = []
        self.set_up_objects_and_camera()


class BoxScene(DefaultScene):
    """A 6 x 10 x 4 room with a light bulb on one wall; override attributes as required."""

    def set_up_objects_and_camera(self):
        make_world_box(self, 'white', (3, 5, 2), (0, 0, 2))
        self.objects[-1].material = "sky"       # ceiling
        self.objects[-2].material = "green"     # floor
        self.add_object(Sphere((-3, -1, 2), 0.5), 'light', name='light')

        self.camera_position = (1, -5, 2)
        self.camera_fov = 55
        self.direct_camera_towards((0, 2, 0.5))


# ================================================================ scene-dev (BASELINE C3)
def scene_dev(spectral=False):
    s = BoxScene(SpectrumShader if spectral else RgbShader)

    for wall in s.get_objects('wall'):
        wall.material = 'red'
    s.get_object('floor').material = 'white'
    s.get_object('ceiling').material = 'green'
    s.get_object('light').bidirectional_light = True
    s.materials['black'] = {'diffuse': 0.2}
    s.materials['light']['diffuse'] = 0.0
    s.add_object(HalfSpace((-1, -1, -2), 5), 'sky')

    egg = Sphere((0, 0, 0.6), 0.3)
    egg.linear_transform(scaling=(1, 1, 2))
    s.add_object(egg, 'mirror')

    # the reference assigns the unused attribute `image_size`; the image stays 800 x 600
    s.image_size = (8 * 120, 6 * 120)
    s.samples_per_pixel = 15000
    s.camera_position = (-2, -3, 1)
    s.direct_camera_towards((0, 0, 0.6))
    s.camera_fov = 50
    s.min_bounces = 2
    s.max_bounces = 5
    return s


# ================================================================ scene-three-objects (C2)
def three_cylinders(origin, axes, radius):
    return ConvexIntersection(origin, [CylinderComponent(a, radius) for a in axes])


def scene_three_objects():
    s = BoxScene()
    m = s.materials
    m['floor'] = {'diffuse': 0.1, 'reflection': 0.15, 'reflection_blur': 0.05}
    m['glass']['transparency'] = (0.4, 0.6, 0.9)
    m['glass']['ior'] = 1.5
    m['wax']['volume_absorption'] = (1.0 - np.array((0.2, 0.8, 0.4))) * 7.0
    m['wax']['diffuse'] = (0.02, 0.03, 0.04)
    m['wax']['reflection'] = (0.01, 0.04, 0.02)

    s.get_object('floor').material = 'floor'
    s.get_object('ceiling').material = 'white'
    s.get_object('light').bidirectional_light = True
    for wall in s.get_objects('wall'):
        wall.material = 'white'

    s.add_object(HalfSpace((-1, -1, -2), 5), 'sky')
    s.add_object(three_cylinders((-0.2, 2.5, 1.0), [(1, 1, 0), (0, 1, 0), (1, 0, 1)], 1), 'mirror')
    cyl_r = .4
    s.add_object(three_cylinders((-0.7, -0.8, cyl_r), [(1, 0, 0), (0, 1, 0), (0, 0, 1)], cyl_r), 'wax')
    s.add_object(Parallelepiped((1.3, -0.5, 0.0), (1, 0, 0), (0, 1.3, 0), (0, 0, 0.6)), 'red')
    dodecahedron_center = (1.8, 0.2, .5 + 0.6)
    s.add_object(Dodecahedron(dodecahedron_center, 0.5), 'glass')

    s.image.size = (1024, 768)
    s.samples_per_pixel = 10000
    s.quasirandom = False
    s.gamma = 1.6
    s.max_bounces = 5
    s.camera_fov = 60
    s.camera_position = (-2, -3, 1.5)
    s.direct_camera_towards((0, 0, 0.5))
    s.camera_sharp_distance = vec_norm(np.array(dodecahedron_center) - np.array(s.camera_position))
    return s


# ================================================================ scene-test (C1): every shape class
UNIT_CUBE_OFF = """OFF
8 6 24
0 0 0
0 0 1
0 1 0
0 1 1
1 0 0
1 0 1
1 1 0
1 1 1
4 0 1 3 2
4 2 3 7 6
4 4 6 7 5
4 0 4 5 1
4 1 5 7 3
4 0 2 6 4
"""

JULIA_C = (-0.2, -0.4, -0.4, -0.4)


def cube_mesh(p, R, **kwargs):
    vertices, faces = mesh_formats.read_off(io.StringIO(UNIT_CUBE_OFF))
    faces = mesh_formats.remove_duplicate_faces(faces)
    return TriangleMesh(vertices, faces, auto_scale=True, center=p, scale=R, **kwargs)


def cube_mesh_in_octree(p, R, **kwargs):
    return Octree(cube_mesh(p, R, **kwargs), max_depth=2, max_faces_per_leaf=1)


def below(p, R):
    return p - numpy.array([0, 0, R])


SHAPES = [
    ('sphere', lambda p, R: Sphere(p, R)),
    ('cone', lambda p, R: Cone(below(p, R), (0, 0, 1), R, R)),
    ('cylinder', lambda p, R: Cylinder(below(p, R), (0, 0, 1), R, R)),
    ('tetrahedron', lambda p, R: Tetrahedron(p, R)),
    ('cube', lambda p, R: Cube(p, R=R)),
    ('octahedron', lambda p, R: Octahedron(p, R)),
    ('dodecahedron', lambda p, R: Dodecahedron(p, R)),
    ('icosahedron', lambda p, R: Icosahedron(p, R)),
    ('parallelepiped', lambda p, R: Parallelepiped(
        p - numpy.array((R * 0.5, R * 0.45, R)), (R, 0, 0), (0, R * 0.9, 0), (0, 0, 1.5 * R))),
    ('julia-ia', lambda p, R: QuaternionJuliaSet(JULIA_C, 4, center=p, scale=R)),
    ('julia-de', lambda p, R: QuaternionJuliaSet2(JULIA_C, 4, center=p, scale=R)),
    ('implicit', lambda p, R: ImplicitSurface(
        'x**4 - 5*x**2 + y**4 - 5*y**2 + z**4 - 5*z**2 + 11.8', p, R * 0.5, 4)),
    ('convex-5', lambda p, R: ConvexIntersection(p, [
        CylinderComponent((0, 1, 0), R),
        ConeComponent((0, -R / 0.2, 0), (0, 1, 0), R),
        SphereComponent((0, 0, 0), R * 1.1),
        HalfSpaceComponent((1, 1, 0), R),
        LayerComponent((1, 0, 0), 0.3 * R / 0.2)])),
    ('mesh', cube_mesh),
    ('mesh-autoflip', lambda p, R: cube_mesh(p, R, auto_flip_normal=True)),
    ('mesh-smooth', lambda p, R: cube_mesh(p, R, shading='smooth', auto_smooth_normals=True)),
    ('octahedron-small', lambda p, R: Octahedron(p, R * 0.5)),   # same tracer, other size
    ('convex-3', lambda p, R: ConvexIntersection(p, [             # a different tracer
        CylinderComponent((1, 1, 0), R),
        SphereComponent((0, 0, 0), R * 1.2),
        LayerComponent((1, 0, 0), 2.0 * R)])),
    ('octree', cube_mesh_in_octree),
    ('distance-field', lambda p, R: DistanceField(
        tracer_code="dist = sqrt(x*x + y*y + z*z) - %g" % R, center=p)),
]

MATERIAL_CYCLE = ['white', 'mirror', 'glass', 'wax']


def scene_test():
    s = BoxScene()
    s.get_object('light').bidirectional_light = True

    side = int(math.ceil(math.sqrt(len(SHAPES))))
    extent = 4.0
    radius = 0.5 * extent / side * 0.5
    coord = lambda i: ((i + 0.5) / float(side) - 0.5) * extent

    for k, (name, factory) in enumerate(SHAPES):
        ix, iy = divmod(k, side)
        tracer = factory(numpy.array((coord(ix), coord(iy), radius)), radius)
        tracer.rotate(axis='z', deg=-45)
        s.add_object(tracer, MATERIAL_CYCLE[k % len(MATERIAL_CYCLE)], name)

    s.max_bounces = 4
    s.min_bounces = 2
    s.image.brighness_reference = numpy.max   # (sic) misspelt in the reference: no effect
    s.brightness = 1.0
    s.direct_camera_towards((0, 0, 0.0))
    return s


# ================================================================ scene-stairs (spectral, depth of field)
class DomeScene(DefaultScene):

    def set_up_objects_and_camera(self):
        self.materials['dome'] = {'diffuse': 0.0}
        self.materials['red'] = {'diffuse': self.spectrum.gaussian(670, 30) * 0.9 + 0.05,
                                 'reflection': 0.1, 'reflection_blur': 0.05}
        sky_distance = 100
        dome = self.add_object(Sphere((0, 0, 0), sky_distance), 'dome', 'dome')
        self.add_object(HalfSpace((0, 0, -1), sky_distance * 0.7), 'sky', 'sky')
        self.add_object(HalfSpace((0, 0, 1), 0.0), 'white', 'floor')
        self.root_object = dome

        self.camera_position = (1.5, -7.5, 1.5)
        self.direct_camera_towards((-0.5, 0, 1.5))


def scene_stairs(n_steps=65):
    s = DomeScene(SpectrumShader)
    s.camera_dof_fstop = 0.05
    s.image.size = (1920 // 2, 1080 // 2)
    s.samples_per_pixel = 100000

    step_pose = Affine(translation=(1, 0, 0))
    ball_pose = Affine(translation=(0, 1.5, 0))
    advance = Affine(rotation_axis=(-0.5, -1, 3.1), rotation_deg=10, translation=(0, 0, 0.07))

    for j in range(n_steps):
        step = Cube((0, 0, 0), 1.0)
        step.coordinates = step_pose * Affine(scaling=(0.5, 0.2, 0.02))
        s.add_object(step, {8: 'light', 0: 'red'}.get(j % 16, 'glass'))

        ball = Sphere((0, 0, 0), 0.1)
        ball.coordinates = ball_pose
        s.add_object(ball, 'red')

        step_pose = advance * step_pose
        ball_pose = advance * ball_pose
    return s


# ================================================================ scene-marchers (C5)
MARCHER_RADIUS = 0.45

ROUNDED_BOX_TORUS = (
    "float qx = fabs(x) - 0.30f, qy = fabs(y) - 0.30f, qz = fabs(z) - 0.30f; "
    "float ox = fmax(qx, 0.0f), oy = fmax(qy, 0.0f), oz = fmax(qz, 0.0f); "
    "float box = sqrt(ox*ox + oy*oy + oz*oz) + fmin(fmax(qx, fmax(qy, qz)), 0.0f) - 0.08f; "
    "float ring = sqrt(x*x + y*y) - 0.42f; "
    "float torus = sqrt(ring*ring + z*z) - 0.07f; "
    "dist = fmin(box, torus)")


def scene_marchers(only=None):
    """only: names of the ray-marched objects to keep (development: cost of each one alone)."""
    s = BoxScene()
    s.get_object('light').bidirectional_light = True
    R = z = MARCHER_RADIUS
    marchers = [
        (ImplicitSurface('x**4 - 5*x**2 + y**4 - 5*y**2 + z**4 - 5*z**2 + 11.8',
                         (-1.2, 0.8, z), R * 0.5, 4), 'mirror', 'implicit'),
        (DistanceField("dist = sqrt(x*x + y*y + z*z) - %g" % R, center=(0.0, 1.6, z)), 'glass', 'sdf-sphere'),
        (DistanceField(ROUNDED_BOX_TORUS, center=(1.2, 0.6, z), bndR=0.7), 'white', 'sdf-box-torus'),
        (QuaternionJuliaSet(JULIA_C, 4, center=(-0.4, -0.6, z), scale=R), 'white', 'julia-ia'),
        (QuaternionJuliaSet2(JULIA_C, 4, center=(0.9, -0.9, z), scale=R), 'wax', 'julia-de'),
    ]
    for tracer, material, name in marchers:
        if only is None or name in only:
            tracer.rotate(axis='z', deg=-30)
            s.add_object(tracer, material, name)
    s.image.size = (3840, 2160)
    s.samples_per_pixel = 16
    s.max_bounces = 4
    s.min_bounces = 2
    s.direct_camera_towards((0, 0.3, 0.3))
    return s


# ================================================================ scene-mesh (C4)
def torus_knot_mesh(n_u=1000, n_v=500, tube=0.45, bump=0.05):
    """(vertices (n_u*n_v, 3) float32, faces (2*n_u*n_v, 3) int32) of a (2,3) torus knot tube."""
    u = np.linspace(0.0, 2 * np.pi, n_u, endpoint=False)
    v = np.linspace(0.0, 2 * np.pi, n_v, endpoint=False)

    def curve(t):
        r = 2.0 + np.cos(3 * t)
        return np.stack((r * np.cos(2 * t), r * np.sin(2 * t), np.sin(3 * t)), axis=-1)
    c = curve(u)
    tangent = curve(u + 1e-4) - curve(u - 1e-4)
    tangent /= np.linalg.norm(tangent, axis=1, keepdims=True)
    side = np.cross(tangent, np.array([0.0, 0.0, 1.0]))
    side /= np.linalg.norm(side, axis=1, keepdims=True)
    up = np.cross(side, tangent)
    U, V = np.meshgrid(u, v, indexing='ij')
    radius = tube + bump * np.sin(13 * U) * np.sin(7 * V)
    pts = (c[:, None, :] + radius[..., None] * (np.cos(V)[..., None] * side[:, None, :]
                                                 + np.sin(V)[..., None] * up[:, None, :]))
    vertices = pts.reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(n_u), np.arange(n_v), indexing='ij')
    a = (i * n_v + j).ravel()
    b = (((i + 1) % n_u) * n_v + j).ravel()
    cc = (((i + 1) % n_u) * n_v + (j + 1) % n_v).ravel()
    d = (i * n_v + (j + 1) % n_v).ravel()
    faces = np.concatenate((np.stack((a, b, cc), 1), np.stack((a, cc, d), 1))).astype(np.int32)
    return vertices, faces


def mesh_through_off_file(vertices, faces, path):
    """the generated mesh "written as OFF and read back through
    mesh_formats.read_off" - here through the native reader (csrc/mesh_io.cpp, which returns what
    the reference's read_off returns, tests/test_mesh_io.py).  Returns (vertices, faces, seconds the
    parse took).  %.9g round-trips binary32, so the arrays come back as they were."""
    import time
    from ..objects import meshes
    meshes.write_off(path, vertices, faces)
    t0 = time.perf_counter()
    vertices, faces = meshes.read_mesh_file(path, 'off')
    return vertices, faces, time.perf_counter() - t0


def scene_mesh(n_u=1000, n_v=500, layout='flat', max_depth=None, max_faces_per_leaf=8,
               max_steps=2000, material='white', pooled=None, overlap='sat', mesh=None, via_off=None, ropes=None, two_level=None):
    """overlap: how the builder files faces under leaves - 'sat' (triangle against box, 3.0 leaf
    records per face on this mesh) or 'reference' (the reference's bounding-sphere rule, 12.7);
    the hits are the same (tests/test_gpu_parity.py).
    mesh: (vertices, faces) to use instead of the generated knot, e.g. what a mesh reader returned;
    via_off: path of an OFF file the generated knot is written to and read back from through the
    native reader before it is used (scene.mesh_parse_seconds then holds the parse time)."""
    s = BoxScene()
    s.get_object('light').bidirectional_light = True
    vertices, faces = mesh if mesh is not None else torus_knot_mesh(n_u, n_v)
    if via_off is not None:
        vertices, faces, s.mesh_parse_seconds = mesh_through_off_file(vertices, faces, via_off)
    if max_depth is None:       # about one level per factor of 8 in the face count
        max_depth = int(min(10, max(3, np.ceil(np.log2(max(len(faces), 8)) / 3.0) + 2)))
    mesh = TriangleMesh(vertices, faces, auto_scale=True, center=(0, 0, 1), scale=1.0,
                        auto_flip_normal=True)
    s.add_object(Octree(mesh, max_depth=max_depth, max_faces_per_leaf=max_faces_per_leaf,
                        builder='native', layout=layout, max_steps=max_steps, pooled=pooled,
                        overlap=overlap, ropes=ropes, two_level=two_level), material, 'knot')
    s.image.size = (1920, 1080)
    s.samples_per_pixel = 16
    s.max_bounces = 4
    s.min_bounces = 2
    s.camera_position = (1.5, -4.2, 2.2)
    s.direct_camera_towards((0, 0, 0.9))
    return s
