"""Host-side scene API: the reference's own unit tests (tests.py:11-88, TestAffine) restated
for this package, plus the packing helpers the renderer relies on."""
import io

import numpy as np
import pytest

from raytracer_amd import geometry as camera
from raytracer_amd import geometry as transformations
from raytracer_amd.objects import meshes as mesh_formats
from raytracer_amd.geometry import Affine
from raytracer_amd.renderer import Renderer
from raytracer_amd import objects as shapes
from conftest import load_scene

EPSILON = 1e-9


def close(a, b):
    return np.linalg.norm(np.ravel(a) - np.ravel(b)) < EPSILON


class TestAffine:
    rng = np.random.RandomState(7)

    def test_identity(self):
        vec = self.rng.normal(size=3)
        t = Affine.identity()
        assert close(vec, t(vec))
        assert t.is_identity()
        assert not Affine(translation=(1, 2, 3)).is_identity()
        assert not Affine(rotation_axis='y', rotation_deg=10).is_identity()
        ccw = Affine(rotation_axis=(1, 0, 0), rotation_deg=90)
        cw = Affine(rotation_axis='x', rotation_deg=-90)
        assert ccw(cw).is_identity()

    def test_translation(self):
        t = Affine(translation=(1, 2, 3))
        assert close(t((0, -1, -3)), (1, 1, 0))
        assert close(t * (0, -1, -3), (1, 1, 0))

    def test_inverse(self):
        t = Affine(linear=self.rng.normal(size=(3, 3)), translation=self.rng.normal(size=3))
        assert t.inverse()(t).is_identity()
        assert (t.inverse() * t).is_identity()

    def test_roto_translation(self):
        t = Affine(rotation_axis='z', rotation_deg=90, translation=(2, 0, 0))
        assert close(t((1, 0, 0)), (2, 1, 0))

    def test_scaling(self):
        assert close(Affine(scaling=(1, 2, 3))((1, 1, 1)), (1, 2, 3))
        assert close(Affine(scaling=10)((1, 2, 3)), (10, 20, 30))

    def test_orthogonality(self):
        assert Affine.identity().is_orthogonal()
        assert Affine(rotation_axis=self.rng.normal(size=3),
                      rotation_deg=self.rng.rand() * 360).is_orthogonal()
        assert not Affine(scaling=2.0).is_orthogonal()

    def test_rotation_aligning_vectors(self):
        mat = transformations.rotation_aligning_vectors((2, 0, 0), (0, 3, 0))
        assert close(mat @ np.array((2, 0, 0)), (0, 2, 0))
        assert Affine(linear=transformations.rotation_aligning_vectors((2, 0, 0), (2, 0, 0))).is_identity()


def test_image_order_matches_reference_loop_nest():
    """renderer.py:438-448 (the reference's `width` is the image height)."""
    r = Renderer(load_scene('scene-dev', (16, 24)), device=False, seed=0)
    rows, cols = r.img_shape
    expected = []
    for col_block in range(cols // 4):
        for row_block in range(rows // 8):
            for in_col in range(4):
                for in_row in range(8):
                    expected.append([in_row + row_block * 8, in_col + col_block * 4])
    assert np.array_equal(r.image_order, np.array(expected))
    # every pixel exactly once
    flat = r.image_order[:, 0] * cols + r.image_order[:, 1]
    assert np.array_equal(np.sort(flat), np.arange(rows * cols))


def test_image_order_fallback_for_ragged_sizes():
    r = Renderer(load_scene('scene-dev', (6, 10)), device=False, seed=0)
    assert r.image_order.shape == (60, 2)
    assert np.array_equal(r.image_order[:7, 1], [0, 1, 2, 3, 4, 5, 0])


def test_object_grouping_and_ids():
    r = Renderer(load_scene('scene-dev', (16, 8)), device=False, seed=0)
    names = [o.tracer.tracer_kernel_name for o in r.scene.objects]
    assert names == sorted(names)
    assert [(t.__class__.__name__, c, o) for t, c, o in r.object_groups] == \
        [('HalfSpace', 7, 0), ('Sphere', 2, 7)]
    assert r.bidirectional and r.bidirectional_light_ids == [7]
    # 4 affine slots per object, no declared parameters for these classes
    assert r.param_table.shape == (9 * 4, 4)


def test_parameter_table_holds_inverse_affine():
    r = Renderer(load_scene('scene-three-objects', (16, 8)), device=False, seed=0)
    for obj, base in zip(r.scene.objects, r.object_param_base):
        n_decl = len(obj.tracer.parameter_types())
        inv = obj.tracer.global_to_tracer_coordinate_transform().inverse()
        rows = r.param_table[base + n_decl: base + n_decl + 4, :3]
        assert np.allclose(rows[:3], inv.linear, rtol=1e-6, atol=1e-7)
        assert np.allclose(rows[3], inv.translation, rtol=1e-6, atol=1e-7)


def test_material_tables_are_property_major():
    r = Renderer(load_scene('scene-dev', (16, 8)), device=False, seed=0)
    n = len(r.scene.objects) + 1
    offsets = dict(r.shader.get_material_property_offsets())
    assert offsets['DIFFUSE'] == 0 and offsets['EMISSION'] == n and offsets['IOR'] == 0
    light_id = r.bidirectional_light_ids[0] + 1
    assert np.allclose(r.material_colors[offsets['EMISSION'] + light_id, :3], (4, 2, .7))
    assert np.allclose(r.material_colors[offsets['DIFFUSE'] + 0, :3], (1, 0, 1))     # ambient
    assert r.material_scalars[offsets['IOR'] + 0] == 1.0


def test_tape_is_deterministic_and_ordered():
    a = Renderer(load_scene('scene-dev', (16, 8)), device=False, seed=3).make_tape(0, 4)
    b = Renderer(load_scene('scene-dev', (16, 8)), device=False, seed=3).make_tape(0, 4)
    for x, y in zip(a[:2], b[:2]):
        assert np.array_equal(x, y)
    passes, bounces, rows = a
    assert rows is None
    assert np.allclose(np.linalg.norm(bounces[:, :, 0, :3], axis=-1), 1.0, atol=1e-6)   # unit rvec
    assert np.all((bounces[:, :, 0, 3] >= 0) & (bounces[:, :, 0, 3] < 1))                 # p
    light_ids = bounces[:, :, 4, 3].view(np.int32)
    assert np.all(light_ids[:, :-1] == 8) and np.all(light_ids[:, -1] == 0)               # 0 on last


def test_read_off_and_dedupe():
    text = "OFF\n4 2 0\n0 0 0\n1 0 0\n1 1 0\n0 1 0\n4 0 1 2 3\n3 0 2 1\n"
    vertices, faces = mesh_formats.read_off(io.StringIO(text))
    assert len(vertices) == 4 and faces == [[0, 1, 2], [0, 2, 3], [0, 2, 1]]
    assert mesh_formats.remove_duplicate_faces(faces) == [[0, 1, 2], [0, 2, 3]]
    with pytest.raises(RuntimeError):
        mesh_formats.read_off(io.StringIO("PLY\n"))


def test_read_zipper():
    text = ("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nelement face 1\n"
            "end_header\n0 0 0 0.5\n1 0 0 0.5\n0 1 0 0.5\n3 0 1 2\n")
    vertices, faces = mesh_formats.read_zipper(io.StringIO(text))
    assert vertices == [[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]] and faces == [[0, 1, 2]]


def random_triangle_soup(n, seed=0, size=0.08):
    rng = np.random.RandomState(seed)
    centers = rng.uniform(-1, 1, (n, 1, 3))
    vertices = (centers + rng.normal(0, size, (n, 3, 3))).reshape(-1, 3)
    return vertices, np.arange(3 * n).reshape(n, 3)


@pytest.mark.parametrize('source', ['scene-test', 'soup'])
def test_octree_stream_is_self_consistent(source):
    if source == 'scene-test':
        scene = load_scene('scene-test', (16, 8))
        octree = [o.tracer for o in scene.objects if isinstance(o.tracer, shapes.Octree)][0]
    else:
        vertices, faces = random_triangle_soup(400)
        octree = shapes.Octree(shapes.TriangleMesh(vertices, faces), max_depth=4)
    data = octree.get_data()['integer']
    n_faces = octree.total_faces
    assert np.array_equal(data[:3 * n_faces], np.ravel(octree.triangle_mesh.faces))
    tree = data[3 * n_faces:]
    mask, ptr = tree[octree.root_data_offset], tree[octree.root_data_offset + 1]
    if source == 'soup':
        assert mask not in (0, 0x100)      # 400 small faces force splits
        assert len(octree.leaves) > 8
    seen_faces = set()

    def walk(mask, ptr, depth):
        if mask == 0x100:
            return
        if mask == 0:
            n = tree[ptr]
            seen_faces.update(int(f) for f in tree[ptr + 1: ptr + 1 + n])
            return
        assert depth < octree.max_depth + 1
        for child in range(8):
            walk(tree[ptr + 2 * child], tree[ptr + 2 * child + 1], depth + 1)
    walk(mask, ptr, 0)
    assert seen_faces == set(range(n_faces))


def test_cube_without_size_is_an_error_like_the_reference():
    with pytest.raises(TypeError):
        shapes.Cube((0, 0, 0))


def test_camera_helpers():
    rot = camera.camera_rotmat((0, 1, 0), (0, 0, 1))
    assert np.allclose(rot.T @ rot, np.eye(3), atol=1e-12)
    rays = camera.camera_rays((8, 4), False, 60, (0, 1, 0), (0, 0, 1))
    assert rays.shape == (4, 8, 3)
    assert np.allclose(np.linalg.norm(rays, axis=-1), 1.0)
    q = camera.quasi_random_direction_sample(10)
    assert q.shape == (10, 3) and np.all(q[:, 2] > 0)


def test_ray_range_selects_column_blocks():
    whole = Renderer(load_scene('scene-dev', (16, 16)), device=False, seed=0)
    part = Renderer(load_scene('scene-dev', (16, 16)), device=False, seed=0, ray_range=(64, 128))
    assert part.n_pixels == 128
    assert np.array_equal(part.host_cam, whole.host_cam[64:192])
    cols = np.unique(part.image_order[:, 1])
    assert list(cols) == list(range(4, 12))          # two 4-pixel-wide column blocks of 16 rows
    img = part.unpermute(np.ones((128, 3), np.float32))
    assert img.shape == (16, 16, 3) and img[:, 4:12].all() and not img[:, :4].any() and not img[:, 12:].any()


@pytest.mark.parametrize('n,depth,leaf', [(400, 4, 5), (3000, 5, 4), (50, 3, 1)])
def test_native_octree_builder_reproduces_the_reference_stream(n, depth, leaf):
    """C++ builder (top-down, per-node face lists) vs the restated reference construction:
    identical integer stream, root cube and leaf count."""
    from raytracer_amd import accelerator
    vertices, faces = random_triangle_soup(n, seed=n)
    py = shapes.Octree(shapes.TriangleMesh(vertices, faces), max_depth=depth, max_faces_per_leaf=leaf)
    native = accelerator.build_octree(np.float32(vertices), faces, depth, leaf)
    assert np.array_equal(py.get_data()['integer'][3 * n:], native['stream'])
    assert py.root_data_offset == native['root_data_offset']
    assert np.allclose(py.root.origin, native['root_origin'], rtol=0, atol=1e-12)
    assert len(py.leaves) == native['n_leaves']
    # flattened layout: every (leaf, face) reference once, edges pre-subtracted in binary32
    tris = native['flat_tris'].reshape(-1, 3, 4)
    assert tris.shape[0] == native['flat_nodes'][native['flat_nodes'][:, 0] > 0, 0].sum()
    face_ids = tris[:, 0, 3].copy().view(np.int32)
    v32 = np.float32(vertices)
    assert np.array_equal(tris[:, 0, :3], v32[faces[face_ids, 0]])
    assert np.array_equal(tris[:, 1, :3], v32[faces[face_ids, 1]] - v32[faces[face_ids, 0]])
    assert set(face_ids) == set(range(n))


def test_native_octree_rejects_bad_input():
    from raytracer_amd import accelerator
    vertices, faces = random_triangle_soup(10)
    with pytest.raises(accelerator.ClrError):
        accelerator.build_octree(np.float32(vertices), faces + 1000, 3, 2)
    with pytest.raises(accelerator.ClrError):
        accelerator.build_octree(np.float32(vertices), faces, 99, 2)


def test_reference_style_scene_file_imports_unchanged(tmp_path):
    """A scene file written for the reference (flat module names) loads through the aliases."""
    import runpy
    import raytracer_amd
    raytracer_amd.install_flat_aliases()
    path = tmp_path / 'scene-flat.py'
    path.write_text(
        "from objects import *\n"
        "from utils import vec_norm\n"
        "from transformations import Affine\n"
        "from scenes.default_scenes import BoxScene\n"
        "from shader import RgbShader\n"
        "import mesh_formats\n"
        "scene = BoxScene()\n"
        "scene.add_object(Sphere((0, 0, 0.5), 0.5), 'mirror')\n"
        "scene.add_object(Cube((1, 1, 0.5), 1.0), 'glass')\n"
        "scene.image.size = (16, 8)\n")
    scene = runpy.run_path(str(path))['scene']
    r = Renderer(scene, device=False, seed=0)
    assert len(r.scene.objects) == 9 and 'Cube_tracer' in r.program_code


def test_speculation_is_off_for_marching_and_mesh_scenes():
    assert Renderer(load_scene('scene-dev', (16, 8)), device=False).speculate
    assert not Renderer(load_scene('scene-test', (16, 8)), device=False).speculate
    assert 'clr_segment_generic' not in Renderer(load_scene('scene-marchers', (16, 8)), device=False).program_code


def test_parallelepiped_light_interface():
    """Light interface of a box: sample points lie on the surface, normals point
    outwards, faces are chosen in proportion to their area."""
    from raytracer_amd.objects import Parallelepiped
    box = Parallelepiped((1.0, -2.0, 0.5), (2, 0, 0), (0, 1, 0), (0, 0, 0.5))
    box.rotate((0, 0, 1), 30)
    assert abs(box.surface_area() - 2 * (2 * 1 + 2 * 0.5 + 1 * 0.5)) < 1e-9
    centre, min_dist = box.center_and_min_sampling_distance()
    assert abs(min_dist - np.sqrt(4 + 1 + 0.25)) < 1e-9
    to_local = box.global_to_tracer_coordinate_transform().inverse()
    rng = np.random.RandomState(0)
    counts = {}
    for _ in range(3000):
        point, normal = box.random_surface_point_and_normal(rng)
        assert abs(np.linalg.norm(normal) - 1) < 1e-12 and np.dot(point - centre, normal) > 0
        local = to_local(point) if callable(to_local) else None
        slab = np.array([np.dot(np.ravel(c.normal), local) / c.h for c in box.components])
        on_face = np.isclose(slab, 0, atol=1e-9) | np.isclose(slab, 1, atol=1e-9)
        assert on_face.sum() >= 1 and ((slab > -1e-9) & (slab < 1 + 1e-9)).all()
        counts[int(np.argmax(on_face))] = counts.get(int(np.argmax(on_face)), 0) + 1
    share = np.array([counts[i] for i in range(3)]) / 3000.0       # faces normal to x, y, z
    assert np.allclose(share, np.array([0.5, 1.0, 2.0]) / 3.5, atol=0.03)


def test_pass_split_policy():
    """Which frames get the pass-split program and how many lanes share a pixel (renderer.py)."""
    from raytracer_amd.renderer import pass_groups_for
    assert pass_groups_for(320 * 240, 16) == 16
    assert pass_groups_for(640 * 480, 32) == 6
    assert pass_groups_for(800 * 600, 32) == 4
    assert pass_groups_for(800 * 600, 1) == 1 and pass_groups_for(800 * 600, 0) == 1
    assert pass_groups_for(64 * 48, 7) == 7                       # never more groups than passes
    assert pass_groups_for(320 * 240, 30) == 15                   # 21 lanes wanted, 2 passes per group: 15 groups
    for size, split in (((320, 240), True), ((800, 600), True), ((1024, 768), False), ((1920, 1080), False)):
        assert Renderer(load_scene('scene-dev', size), device=False, seed=0).split_passes is split


def test_occupancy_choice_avoids_builds_that_spill_for_noise_level_gains():
    from raytracer_amd.renderer import pick_occupancy_hint
    # hint 7 is 0.4 % faster but needs scratch that the others do not: not worth it
    timings = {None: 1.40, 4: 1.38, 5: 1.30, 6: 1.249, 7: 1.244, 8: 1.9}
    scratch = {None: 64, 4: 64, 5: 64, 6: 64, 7: 264, 8: 900}
    assert pick_occupancy_hint(timings, scratch) == 6
    # a build that spills but wins clearly is taken
    timings[7] = 1.10
    assert pick_occupancy_hint(timings, scratch) == 7
    # no difference in scratch: plain minimum
    assert pick_occupancy_hint({None: 2.0, 5: 1.5}, {None: 0, 5: 0}) == 5
    # a few spilled registers: the margin grows with their number (0.4 % each, at most 3 %)
    assert pick_occupancy_hint({None: 1.464, 7: 1.464, 8: 1.447}, {None: 0, 7: 0, 8: 2}) == 8
    assert pick_occupancy_hint({None: 1.464, 7: 1.464, 8: 1.458}, {None: 0, 7: 0, 8: 2}) in (None, 7)
    assert pick_occupancy_hint({None: 2.0, 6: 1.96}, {None: 0, 6: 40}) is None
    assert pick_occupancy_hint({None: 2.0, 6: 1.90}, {None: 0, 6: 40}) == 6


def test_pass_split_launches_are_cut_to_bound_the_addend_log():
    from raytracer_amd.renderer import Renderer, log_passes_limit, LOG_BYTES_LIMIT
    # 1080p, max_bounces 4: 25 floats per pass and pixel = 207 MB per pass -> 41 passes in 8 GB
    assert log_passes_limit(1920 * 1080, 4) == LOG_BYTES_LIMIT // (25 * 1920 * 1080 * 4) == 41
    assert log_passes_limit(3840 * 2160, 5) == 8
    assert log_passes_limit(10 ** 9, 5) == 1                       # never less than one pass
    scene = load_scene('scene-dev', (64, 48))
    r = Renderer(scene, device=False, seed=0, split_passes=True)
    assert r.max_passes_per_launch() == log_passes_limit(64 * 48, scene.max_bounces)
    assert Renderer(load_scene('scene-dev', (64, 48)), device=False, seed=0, split_passes=False).max_passes_per_launch() is None
    # render_tape cuts a long tape into launches of at most that many passes, numbered globally
    calls = []
    r.max_passes_per_launch = lambda: 3
    r._upload_tape = lambda tape: None
    r.dev_pass_tape = r.dev_bounce_tape = r.cam = r.img = r.dev_vector_data = r.dev_integer_data = type('B', (), {'ptr': 0})()
    r.dev_rows = None
    r._split_args = lambda n, debug=False, static_log=False: {}

    class Acc:
        def render_passes(self, args, stream=None, timed=True, static_log=False):
            calls.append((args.n_passes, args.first_pass))
            return 1.0
    r.acc = Acc()
    ms, _ = r.render_tape(r.make_tape(0, 8), first_pass=10)
    assert calls == [(3, 10), (3, 13), (2, 16)] and ms == 3.0


def _leaf_boxes(native):
    """(origin, size, first record, count) of every leaf of a flattened native tree."""
    nodes, out = native['flat_nodes'], []
    stack = [(0, np.asarray(native['root_origin'], float), float(native['root_size']))]
    while stack:
        index, origin, size = stack.pop()
        a, b = nodes[index]
        if a >= 0:
            out.append((origin, size, b, a))
            continue
        for k in range(8):
            xyz = np.array([(k >> 2) & 1, (k >> 1) & 1, k & 1], float)
            stack.append((b + k, origin + xyz * 0.5 * size, 0.5 * size))
    return out


@pytest.mark.parametrize('n,depth,leaf', [(400, 4, 5), (1500, 5, 4)])
def test_triangle_box_overlap_rule_gives_a_tighter_tree_that_misses_nothing(n, depth, leaf):
    """overlap='sat' (no counterpart in the reference, whose rule is objects/octree.py:201-233):
    every leaf lists a subset of what the reference rule lists for the same box, and every
    triangle that really reaches into a leaf's box - checked by sampling points of the triangle -
    is listed there, which is what the traversal needs (it accepts hits inside the leaf only)."""
    from raytracer_amd import accelerator
    vertices, faces = random_triangle_soup(n, seed=n + 1)
    v32 = np.float32(vertices)
    ref = accelerator.build_octree(v32, faces, depth, leaf, overlap='reference')
    sat = accelerator.build_octree(v32, faces, depth, leaf, overlap='sat')
    assert sat['leaf_records'] < 0.7 * ref['leaf_records']
    ids = lambda tree, first, count: set(tree['flat_tris'].reshape(-1, 3, 4)[first:first + count, 0, 3].copy().view(np.int32))
    ref_leaves = {(tuple(np.round(o, 9)), round(s, 9)): ids(ref, f, c) for o, s, f, c in _leaf_boxes(ref)}
    rng = np.random.RandomState(5)
    w = rng.dirichlet((1, 1, 1), size=60)                        # points of a triangle
    tri = v32[faces].astype(float)                               # (n, 3, 3)
    points = np.einsum('sk,nkd->nsd', w, tri)                    # (n, 60, 3)
    checked = 0
    for origin, size, first, count in _leaf_boxes(sat):
        listed = ids(sat, first, count)
        # a SAT leaf may be a coarser node than the reference tree's leaves (fewer faces -> it
        # stops splitting earlier); where the same box is a leaf of both, SAT lists a subset
        key = (tuple(np.round(origin, 9)), round(size, 9))
        if key in ref_leaves:
            assert listed <= ref_leaves[key]
            checked += 1
        inside = ((points >= origin) & (points <= origin + size)).all(axis=2).any(axis=1)
        assert set(np.flatnonzero(inside)) <= listed
    assert checked > 10


def test_octree_hits_equal_brute_force_under_both_overlap_rules():
    """first hits through the octree (oracle walking the reference stream of the
    reference-rule tree and of the triangle-box tree) equal a brute-force sweep over all faces
    of the same mesh: same face, same distance bits."""
    from oracle import oracle
    from raytracer_amd.scenes.library import BoxScene, torus_knot_mesh
    vertices, faces = torus_knot_mesh(60, 40)                    # 4800 triangles

    def scene_with(make_tracer):
        s = BoxScene()
        mesh = shapes.TriangleMesh(vertices, faces, auto_scale=True, center=(0, 0, 1), scale=1.0,
                                   auto_flip_normal=True)
        s.add_object(make_tracer(mesh), 'white', 'knot')
        s.image.size = (96, 64)
        s.camera_position = (1.5, -4.2, 2.2)
        s.direct_camera_towards((0, 0, 0.9))
        return s
    results = []
    for make_tracer in (lambda m: m,
                        lambda m: shapes.Octree(m, max_depth=5, max_faces_per_leaf=6, builder='native'),
                        lambda m: shapes.Octree(m, max_depth=5, max_faces_per_leaf=6, builder='native', overlap='sat')):
        r = Renderer(scene_with(make_tracer), device=False, seed=2)
        ids, t, normal = oracle.OracleScene(r).first_hit(r.make_tape(0, 1))
        # objects are numbered after sorting by class name, so the knot's id depends on its class
        knot_id = [o.name for o in r.scene.objects].index('knot') + 1
        results.append((ids[:, 0] == knot_id, ids[:, 1], t, normal))
    on_knot = results[0][0]
    assert on_knot.mean() > 0.1
    for hit, face, t, normal in results[1:]:
        assert np.array_equal(hit, on_knot)
        assert np.array_equal(face[on_knot], results[0][1][on_knot])
        assert np.array_equal(t.view(np.uint32), results[0][2].view(np.uint32))
        assert np.array_equal(normal.view(np.uint32), results[0][3].view(np.uint32))


def tonemap_cases():
    """(name, settings, base or None, sums, expected uint8 image) of tests/golden/host-io.npz: what
    the reference's Image._to_24bit (/root/reference/imgutils.py:60-80, with imgutils.flares :121-148
    in the flare cases) returned."""
    import os
    from conftest import GOLDEN_DIR
    from raytracer_amd.imgutils import ColorEncodingSettings
    fx = np.load(os.path.join(GOLDEN_DIR, 'host-io.npz'))
    for name in fx['tonemap_cases']:
        gamma, brightness, floor, with_base, with_flares = fx['tonemap_%s_settings' % name]
        settings = ColorEncodingSettings()
        settings.flares = bool(with_flares)
        settings.gamma, settings.brightness = float(gamma), float(brightness)
        settings.low_normalization = np.min if floor else None
        yield (str(name), settings, fx['tonemap_base'] if with_base else None, fx['tonemap_sums'],
               fx['tonemap_%s_rgb' % name])


def test_host_tone_mapping_equals_the_reference():
    from raytracer_amd.imgutils import Image
    seen = 0
    for name, settings, base, sums, want in tonemap_cases():
        got = Image(data=None if base is None else base.copy(), settings=settings)._to_24bit(sums.astype(np.float64))
        assert got.dtype == np.uint8 and np.array_equal(got, want), name
        assert want.std() > 10 and (want == 255).any()
        seen += 1
    assert seen == 7


def _on_surface(tracer, point, normal, tol=1e-9):
    """A world point lies on the boundary of a plane-bounded convex body, and `normal` is the
    outward unit normal of the face it lies on."""
    to_world = tracer.global_to_tracer_coordinate_transform()
    local = np.linalg.solve(np.asarray(to_world.linear, float), point - np.ravel(to_world.translation))
    slack = np.array([d - np.dot(n, local) for n, d in tracer._half_spaces()])
    if not (slack > -tol).all() or not (np.abs(slack) < 1e-7).any():
        return False
    tight = np.argmin(np.abs(slack))
    face_normal = np.linalg.solve(np.asarray(to_world.linear, float).T, tracer._half_spaces()[tight][0])
    face_normal /= np.linalg.norm(face_normal)
    return abs(np.linalg.norm(normal) - 1) < 1e-12 and np.dot(normal, face_normal) > 1 - 1e-9


def test_light_interface_of_polyhedra_cylinder_and_cone():
    """surface_area / random_surface_point_and_normal /
    center_and_min_sampling_distance beyond the reference's Sphere (objects/sphere.py:12-25).
    Areas against closed forms, points on the surface with the outward normal of their face,
    faces drawn in proportion to their area, every point within the stated sampling distance."""
    rng = np.random.RandomState(3)
    phi = (1 + 5 ** 0.5) / 2
    cube = shapes.Cube((1, 2, 3), side=2.0)
    assert abs(cube.surface_area() - 24.0) < 1e-9
    # circumradius-style sizes of the regular solids -> edge lengths -> closed-form areas
    for solid, area_of_edge, faces in ((shapes.Tetrahedron((0, 0, 0), 1.0), lambda a: 3 ** 0.5 * a * a, 4),
                                       (shapes.Octahedron((0, 0, 0), 1.0), lambda a: 2 * 3 ** 0.5 * a * a, 8),
                                       (shapes.Icosahedron((0, 0, 0), 1.0), lambda a: 5 * 3 ** 0.5 * a * a, 20),
                                       (shapes.Dodecahedron((0, 0, 0), 1.0), lambda a: 3 * (25 + 10 * 5 ** 0.5) ** 0.5 * a * a, 12)):
        triangles = solid._world_triangles()
        corners = np.unique(np.round(np.array([p for t in triangles for p in t[:3]]), 9), axis=0)
        edge = min(np.linalg.norm(a - b) for i, a in enumerate(corners) for b in corners[i + 1:])
        assert abs(solid.surface_area() - area_of_edge(edge)) < 1e-6 * solid.surface_area()
        assert len({tuple(np.round(t[3], 9)) for t in triangles}) == faces
    # a sheared, stretched, rotated box: the parallelogram areas of its world-space faces
    box = shapes.Cube((0.3, -0.2, 0.5), side=1.0)
    box.linear_transform(linear=[[1.0, 0.4, 0.0], [0.0, 1.5, 0.2], [0.1, 0.0, 0.7]])
    box.rotate(axis='z', deg=25)
    # in tracer coordinates the cube is [-1, 1]^3: its world edges are twice the matrix columns
    m = 2.0 * np.asarray(box.global_to_tracer_coordinate_transform().linear, float)
    expected = 2 * sum(np.linalg.norm(np.cross(m[:, i], m[:, j])) for i, j in ((0, 1), (1, 2), (0, 2)))
    assert abs(box.surface_area() - expected) < 1e-9 * expected
    for solid in (cube, box, shapes.Tetrahedron((1, 0, 0), 0.7), shapes.Dodecahedron((0, 1, 0), 0.5)):
        centre, reach = solid.center_and_min_sampling_distance()
        shares = {}
        for _ in range(400):
            point, normal = solid.random_surface_point_and_normal(rng)
            assert _on_surface(solid, point, normal)
            assert np.linalg.norm(point - centre) <= 0.5 * reach + 1e-9
            key = tuple(np.round(normal, 6))
            shares[key] = shares.get(key, 0) + 1
        by_face = {}
        for t in solid._world_triangles():
            by_face[tuple(np.round(t[3], 6))] = by_face.get(tuple(np.round(t[3], 6)), 0.0) + t[4]
        total = sum(by_face.values())
        for key, area in by_face.items():
            assert abs(shares.get(key, 0) / 400.0 - area / total) < 0.09, (key, shares.get(key, 0), area / total)
    # round bodies: closed-form areas, points on the surface, unit outward normals
    cyl = shapes.Cylinder((0.5, 0.2, 0.1), (1, 1, 0), 2.0, 0.5)
    cone = shapes.Cone((0, 0, 2), (0, 0, -1), 1.5, 0.6)
    assert abs(cyl.surface_area() - (2 * np.pi * 0.25 + 2 * np.pi * 0.5 * 2.0)) < 1e-12
    assert abs(cone.surface_area() - (np.pi * 0.36 + np.pi * 0.6 * np.hypot(0.6, 1.5))) < 1e-12
    axis = np.array([1, 1, 0]) / 2 ** 0.5
    on_side = 0
    for _ in range(300):
        p, n = cyl.random_surface_point_and_normal(rng)
        rel = p - np.array([0.5, 0.2, 0.1])
        z, rho = rel.dot(axis), np.linalg.norm(rel - rel.dot(axis) * axis)
        assert abs(np.linalg.norm(n) - 1) < 1e-12 and -1e-9 <= z <= 2 + 1e-9 and rho <= 0.5 + 1e-9
        lateral = abs(rho - 0.5) < 1e-9 and abs(n.dot(axis)) < 1e-9
        cap = (abs(z) < 1e-9 and n.dot(axis) < -1 + 1e-9) or (abs(z - 2) < 1e-9 and n.dot(axis) > 1 - 1e-9)
        assert lateral or cap
        on_side += lateral
        assert np.linalg.norm(p - cyl.center_and_min_sampling_distance()[0]) <= 0.5 * cyl.center_and_min_sampling_distance()[1] + 1e-9
    assert abs(on_side / 300.0 - 0.8) < 0.08                    # lateral share 2 pi R h / total = 0.8
    for _ in range(300):
        p, n = cone.random_surface_point_and_normal(rng)
        rel = p - np.array([0, 0, 2.0])
        depth, rho = -rel[2], np.hypot(rel[0], rel[1])           # along the axis (0, 0, -1)
        assert abs(np.linalg.norm(n) - 1) < 1e-12 and -1e-9 <= depth <= 1.5 + 1e-9
        lateral = abs(rho - 0.6 * depth / 1.5) < 1e-9 and abs(n.dot(rel)) < 1e-9      # normal is perpendicular to the generator
        base = abs(depth - 1.5) < 1e-9 and n[2] < -1 + 1e-9
        assert lateral or base
    # stretched round bodies are refused like a scaled Sphere light
    squashed = shapes.Cylinder((0, 0, 0), (0, 0, 1), 1.0, 1.0)
    squashed.linear_transform(scaling=(1, 2, 1))
    with pytest.raises(RuntimeError):
        squashed.surface_area()


def test_light_samples_extend_the_tape_record():
    """Renderer(light_samples=k): the tape record of a segment grows by (point, normal) of k - 1
    further points of the light it names; the first five rows keep their meaning and, up to the
    first extra draw, their values."""
    from conftest import load_scene
    from raytracer_amd.renderer import Renderer, BOUNCE_RECORD
    one = Renderer(load_scene('scene-dev', (32, 24)), seed=3, device=False)
    four = Renderer(load_scene('scene-dev', (32, 24)), seed=3, device=False, light_samples=4)
    assert one.bounce_record == BOUNCE_RECORD == 5 and four.bounce_record == 11
    t1, t4 = one.make_tape(0, 3)[1], four.make_tape(0, 3)[1]
    assert t1.shape == (3, one.scene.max_bounces, 5, 4) and t4.shape == (3, one.scene.max_bounces, 11, 4)
    assert np.array_equal(t1[0, 0, 2:5], t4[0, 0, 2:5])
    light = four.scene.objects[four.bidirectional_light_ids[0]].tracer
    centre, radius = np.ravel(np.asarray(light.center_and_min_sampling_distance()[0], float)), float(light.R)
    for rec in t4.reshape(-1, 11, 4):
        for i in range(3):
            point, normal = rec[5 + 2 * i, :3], rec[6 + 2 * i, :3]
            assert abs(np.linalg.norm(point - centre) - radius) < 1e-5
            assert np.allclose((point - centre) / radius, normal, atol=1e-5)
    assert len({tuple(rec[5]) for rec in t4.reshape(-1, 11, 4)}) == t4.shape[0] * t4.shape[1]
    with pytest.raises(ValueError):
        Renderer(load_scene('scene-dev', (32, 24)), device=False, light_samples=0)


def test_slices_of_the_frame_for_launch_chains():
    """Renderer.part_bounds: the frame cut into overlap_parts slices of contiguous rays at multiples
    of the block size; frames too small to be worth it stay whole."""
    for size, parts, want_parts in (((1920, 1080), 2, 2), ((1920, 1080), 3, 3), ((320, 240), 8, 8),
                                    ((320, 240), 64, 18), ((64, 48), 4, 1), ((100, 77), 2, 1)):
        r = Renderer(load_scene('scene-dev', size), device=False, overlap_parts=parts)
        cuts = r.part_bounds()
        assert cuts[0] == 0 and cuts[-1] == r.n_pixels and len(cuts) == want_parts + 1, (size, parts, cuts)
        assert all(b > a for a, b in zip(cuts, cuts[1:]))
        assert all(c % 256 == 0 for c in cuts[:-1])
        sizes = np.diff(cuts)
        assert sizes.max() - sizes.min() <= 512
    assert Renderer(load_scene('scene-dev', (320, 240)), device=False).part_bounds() == [0, 320 * 240]
    # 'auto': two slices for a one-lane program, one slice and two steps in flight for a pass-split one
    big = Renderer(load_scene('scene-dev', (1920, 1080)), device=False, overlap_parts='auto', steps_in_flight='auto')
    small = Renderer(load_scene('scene-dev', (320, 240)), device=False, overlap_parts='auto', steps_in_flight='auto')
    assert not big.split_passes and (big.overlap_parts, big.steps_in_flight) == (2, 1)
    assert small.split_passes and (small.overlap_parts, small.steps_in_flight) == (1, 2)
    plain = Renderer(load_scene('scene-dev', (320, 240)), device=False)
    assert (plain.overlap_parts, plain.steps_in_flight) == (1, 1)
    plain.steps_in_flight = 3
    assert plain.steps_in_flight == 3


@pytest.mark.parametrize('rng_mode', ['tape', 'philox'])
def test_oracle_pixel_subsets_equal_the_whole_frame(rng_mode):
    """The oracle walks a subset of the frame's rays (`pixels=`) with the numbers those rays get in
    the whole frame: shared tape, or per-pixel streams keyed by the ray's own index (also inside a
    tile).  The full-size GPU parity tests rest on this."""
    from oracle import oracle
    from raytracer_amd.renderer import Renderer
    for ray_range in (None, (256, 1024)):
        r = Renderer(load_scene('scene-dev', (64, 48)), seed=3, rng_mode=rng_mode, device=False, ray_range=ray_range)
        osc = oracle.OracleScene(r)
        tape = r.make_tape(0, 3)
        whole, _, _ = osc.render(tape, first_pass=5)
        pixels = np.arange(1, r.n_pixels, 37)
        part, _, _ = osc.render(tape, first_pass=5, pixels=pixels)
        assert part.shape == (len(pixels), 4) and part[:, :3].max() > 0
        assert np.array_equal(part.view(np.uint32), whole[pixels].view(np.uint32))
