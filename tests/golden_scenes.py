"""Extra parity scenes, described once against the scene API and built twice: by
tools/make_golden.py with the reference's own modules (to produce fixtures) and by the tests
with this package's modules.  `api` needs BoxScene, Sphere, Cylinder, HalfSpace."""


def variant_camera(api):
    """scene-dev-like room exercising the host-side sampling options: quasi-random directions
    on the second segment, flat CCD camera, depth of field, no tent filter."""
    s = api.BoxScene()
    for wall in s.get_objects('wall'):
        wall.material = 'red'
    s.get_object('light').bidirectional_light = True
    s.add_object(api.HalfSpace((-1, -1, -2), 5), 'sky')
    s.add_object(api.Sphere((0.2, 0.1, 0.5), 0.5), 'mirror')
    s.add_object(api.Cylinder((-1.0, 0.5, 0.0), (0, 0, 1), 0.9, 0.3), 'glass')
    s.quasirandom = True
    s.tent_filter = False
    s.camera_flat_ccd = True
    s.camera_dof_fstop = 0.05
    s.camera_position = (-2, -3, 1)
    s.direct_camera_towards((0, 0, 0.6))
    s.camera_fov = 50
    s.min_bounces = 2
    s.max_bounces = 5
    return s


def variant_media(api):
    """Participating media and blur paths: fog in the ambient medium, wax (scattering +
    coloured absorption + blurred refraction), brushed metal (blurred reflection); the room is
    enclosed in a dome that is the root object; no Russian-roulette floor."""
    s = api.BoxScene()
    s.materials['default']['volume_scattering'] = 0.15
    s.materials['default']['volume_scattering_blur'] = 0.03
    s.get_object('light').bidirectional_light = True
    dome = s.add_object(api.Sphere((0, 0, 0), 50.0), 'white', 'dome')
    s.root_object = dome
    s.add_object(api.Sphere((-0.6, 0.3, 0.5), 0.5), 'wax')
    s.add_object(api.Sphere((0.8, -0.4, 0.4), 0.4), 'brushed metal')
    s.add_object(api.Cylinder((0.2, 1.2, 0.0), (0, 0, 1), 1.0, 0.35), 'glass')
    s.min_russian_prob = 0
    s.max_bounces = 5
    s.min_bounces = 3
    s.direct_camera_towards((0, 0, 0.4))
    return s


def spectral_dev(api):
    """scene-dev with the spectral shader and materials, i.e. the two lines the reference's
    scene-dev.py has commented out (scenes/scene-dev.py:3-6), plus a dispersive glass ball so
    that the wavelength-dependent index of refraction is on the path."""
    s = api.BoxScene(api.SpectrumShader)
    for wall in s.get_objects('wall'):
        wall.material = 'red'
    s.get_object('floor').material = 'white'
    s.get_object('ceiling').material = 'green'
    s.get_object('light').bidirectional_light = True
    s.materials['black'] = {'diffuse': 0.2}
    s.materials['light']['diffuse'] = 0.0
    s.add_object(api.HalfSpace((-1, -1, -2), 5), 'sky')
    egg = api.Sphere((0, 0, 0.6), 0.3)
    egg.linear_transform(scaling=(1, 1, 2))
    s.add_object(egg, 'mirror')
    s.add_object(api.Sphere((-0.9, -0.6, 0.35), 0.35), 'glass')
    s.add_object(api.Sphere((0.9, -0.2, 0.3), 0.3), 'wax')
    s.camera_position = (-2, -3, 1)
    s.direct_camera_towards((0, 0, 0.6))
    s.camera_fov = 50
    s.min_bounces = 2
    s.max_bounces = 5
    return s


def spectral_stairs(api, n_steps=12):
    """The first steps of the reference's only spectral scene (scenes/scene-stairs.py:8-60):
    dome as root object, depth of field, glass / red / light cubes under a chain of affine maps."""
    class DomeScene(api.DefaultScene):
        def set_up_objects_and_camera(self):
            self.materials['dome'] = {'diffuse': 0.0}
            self.materials['red'] = {'diffuse': self.spectrum.gaussian(670, 30) * 0.9 + 0.05,
                                     'reflection': 0.1, 'reflection_blur': 0.05}
            dome = self.add_object(api.Sphere((0, 0, 0), 100), 'dome', 'dome')
            self.add_object(api.HalfSpace((0, 0, -1), 70.0), 'sky', 'sky')
            self.add_object(api.HalfSpace((0, 0, 1), 0.0), 'white', 'floor')
            self.root_object = dome
            self.camera_position = (1.5, -7.5, 1.5)
            self.direct_camera_towards((-0.5, 0, 1.5))

    s = DomeScene(api.SpectrumShader)
    s.camera_dof_fstop = 0.05
    step_pose = api.Affine(translation=(1, 0, 0))
    ball_pose = api.Affine(translation=(0, 1.5, 0))
    advance = api.Affine(rotation_axis=(-0.5, -1, 3.1), rotation_deg=10, translation=(0, 0, 0.07))
    for j in range(n_steps):
        step = api.Cube((0, 0, 0), 1.0)
        step.coordinates = step_pose * api.Affine(scaling=(0.5, 0.2, 0.02))
        s.add_object(step, {8: 'light', 0: 'red'}.get(j % 16, 'glass'))
        ball = api.Sphere((0, 0, 0), 0.1)
        ball.coordinates = ball_pose
        s.add_object(ball, 'red')
        step_pose = advance * step_pose
        ball_pose = advance * ball_pose
    return s


def bumpy_torus(n_u=60, n_v=40):
    """(vertices (n_u * n_v, 3), faces (2 n_u n_v, 3)): a torus with a rippled tube, 4800 triangles."""
    import numpy as np
    u, v = np.meshgrid(np.linspace(0, 2 * np.pi, n_u, endpoint=False),
                       np.linspace(0, 2 * np.pi, n_v, endpoint=False), indexing='ij')
    tube = 0.33 + 0.05 * np.sin(5 * u) * np.cos(3 * v)
    ring = 1.0 + tube * np.cos(v)
    vertices = np.stack((ring * np.cos(u), ring * np.sin(u), tube * np.sin(v) + 0.08 * np.sin(2 * u)), axis=-1)
    i, j = np.meshgrid(np.arange(n_u), np.arange(n_v), indexing='ij')
    a = (i * n_v + j).ravel()
    b = (((i + 1) % n_u) * n_v + j).ravel()
    c = (((i + 1) % n_u) * n_v + (j + 1) % n_v).ravel()
    d = (i * n_v + (j + 1) % n_v).ravel()
    faces = np.concatenate((np.stack((a, b, c), 1), np.stack((a, c, d), 1)))
    return vertices.reshape(-1, 3), faces.astype(np.int32)


def variant_mesh(api):
    """A smooth-shaded 4800-triangle mesh inside an Octree (depth 4) built by the scene API's own
    builder, tilted, glass next to it: pins the octree stream at a size where the tree is deep,
    the smooth normals (vertex normals interpolated by barycentric weights) and everything a path
    does after hitting a mesh.  (The reference generates the smooth-normal code for every Octree,
    which is only right for a smooth mesh like this one - tests/test_golden.py, flat_octree_ids.)"""
    s = api.BoxScene()
    s.get_object('light').bidirectional_light = True
    vertices, faces = bumpy_torus()
    mesh = api.TriangleMesh(vertices, faces, center=(0.1, 0.4, 0.9), scale=0.9, auto_scale=True,
                            shading='smooth', auto_smooth_normals=True)
    mesh.rotate(axis='x', deg=35)
    s.add_object(api.Octree(mesh, max_depth=4, max_faces_per_leaf=6), 'white', 'torus')
    s.add_object(api.Sphere((-1.0, -0.8, 0.4), 0.4), 'glass')
    s.max_bounces = 4
    s.min_bounces = 2
    s.direct_camera_towards((0, 0.3, 0.7))
    return s


VARIANTS = {'variant-camera': variant_camera, 'variant-mesh': variant_mesh, 'variant-media': variant_media,
            'spectral-dev': spectral_dev, 'spectral-stairs': spectral_stairs}
