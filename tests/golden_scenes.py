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


VARIANTS = {'variant-camera': variant_camera, 'variant-media': variant_media}
