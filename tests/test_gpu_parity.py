"""Parity tests proper: the HIP back end, called through the C ABI, against the CPU oracle on
the same scene, camera rays and random tape.

Bar: object / sub-object ids, hit distances, normals, per-segment path state and the
accumulated image are BIT-EXACT (both sides evaluate the same IEEE binary32 operation
sequence; the device program is built with -ffp-contract=off).  DistanceField objects run
user-supplied code with double literals through the platform math library, so for scenes
containing them the state is compared with the tolerance north_star states for floats
(t within 1e-4 relative) and ids must still match exactly.
"""
import numpy as np
import pytest

from raytracer_amd.renderer import Renderer
from conftest import load_scene

pytestmark = pytest.mark.gpu

SCENES = ['scene-dev', 'scene-three-objects', 'scene-test']


def make(name, size, seed=1):
    from oracle import oracle
    r = Renderer(load_scene(name, size), seed=seed)
    return r, oracle.OracleScene(r)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize('name', SCENES)
def test_first_hit_ids_exact_and_t_within_tolerance(name):
    r, osc = make(name, (160, 120))
    tape = r.make_tape(0, 1)
    ids, t, normal = r.first_hit(tape)
    oids, ot, onormal = osc.first_hit(tape)
    assert np.array_equal(ids, oids)
    assert (ids[:, 0] > 0).all()                      # closed scene: every ray hits something
    assert np.all(np.abs(t - ot) <= 1e-4 * np.abs(ot))
    assert np.array_equal(bits(t), bits(ot))
    assert np.array_equal(bits(normal), bits(onormal))


@pytest.mark.parametrize('name', SCENES)
def test_segment_state_matches_oracle(name):
    r, osc = make(name, (64, 48))
    tape = r.make_tape(0, 1)
    _, dump = r.render_tape(tape, debug=True)
    oimg, odump, _ = osc.render(tape, debug=True)
    assert dump.shape == odump.shape
    ints = slice(0, 1)          # record 0: ids / inside / diffusions_left as int bits
    assert np.array_equal(bits(dump[:, :, ints]), bits(odump[:, :, ints]))
    assert np.array_equal(bits(dump), bits(odump))
    assert np.array_equal(bits(r.img.get()), bits(oimg))


@pytest.mark.parametrize('name', SCENES)
@pytest.mark.parametrize('regenerate', [False, True])
def test_accumulated_image_bit_exact(name, regenerate):
    r, osc = make(name, (96, 64))
    r.regenerate = regenerate
    tape = r.make_tape(0, 12)
    r.render_tape(tape)
    oimg, _, _ = osc.render(tape)
    img = r.img.get()
    assert np.isfinite(img).all() and img[:, :3].mean() > 0
    assert np.array_equal(bits(img), bits(oimg))


@pytest.mark.parametrize('name', ['scene-dev-spectral', 'scene-stairs-small'])
def test_spectral_mode_bit_exact(name):
    """Spectral shading (one wavelength per pass, material row and colour mask in the tape,
    shader.py:148-194).  Parity vs the reference is unpinned here (its CIE table is not
    shipped): HIP is checked against the oracle only."""
    r, osc = make(name, (64, 40))
    assert not r.shader.rgb
    tape = r.make_tape(0, 1)
    assert tape[2] is not None and tape[2].shape == (1, 10 * (len(r.scene.objects) + 1))
    _, dump = r.render_tape(tape, debug=True)
    oimg, odump, _ = osc.render(tape, debug=True)
    assert np.array_equal(bits(dump), bits(odump))
    assert np.array_equal(bits(r.img.get()), bits(oimg))
    tape = r.make_tape(1, 10)
    r.render_tape(tape)
    oimg, _, _ = osc.render(tape, img=oimg)
    img = r.img.get()
    assert np.isfinite(img).all() and img[:, :3].std() > 0
    assert np.array_equal(bits(img), bits(oimg))


def test_ray_marching_scene_matches_oracle():
    """BASELINE config C5 at reduced size: implicit surface, two distance fields (user code
    with libm calls: compared with tolerance), both Julia-set tracers."""
    r, osc = make('scene-marchers', (96, 56))
    tape = r.make_tape(0, 1)
    ids, t, normal = r.first_hit(tape)
    oids, ot, onormal = osc.first_hit(tape)
    assert np.array_equal(ids, oids)
    assert set(np.unique(ids[:, 0])) >= {o + 1 for o, obj in enumerate(r.scene.objects) if obj.name in
                                         ('implicit', 'sdf-sphere', 'sdf-box-torus', 'julia-ia', 'julia-de')}
    assert np.all(np.abs(t - ot) <= 1e-4 * np.abs(ot))
    assert np.allclose(normal, onormal, atol=1e-4)
    tape = r.make_tape(1, 6)
    r.render_tape(tape)
    oimg, _, _ = osc.render(tape)
    img = r.img.get()
    rel = np.sqrt(np.mean((img[:, :3] - oimg[:, :3]) ** 2)) / np.sqrt(np.mean(oimg[:, :3] ** 2))
    assert rel <= 1e-4, rel


def test_tile_partition_reassembles_the_frame():
    """Tiles = contiguous ranges of ray indices; rendering the tiles separately and adding the
    frames equals rendering the whole frame, bit for bit."""
    whole = Renderer(load_scene('scene-three-objects', (64, 48)), seed=2)
    tape = whole.make_tape(0, 3)
    whole.render_tape(tape)
    full = whole.get_image()
    total = np.zeros_like(full)
    n = whole.n_pixels
    for first in range(0, n, 1024):
        part = Renderer(load_scene('scene-three-objects', (64, 48)), seed=2, ray_range=(first, min(1024, n - first)))
        part.render_tape(tape)
        tile = part.get_image()
        assert np.count_nonzero(tile.sum(axis=2)) <= 1024
        total += tile
    assert np.array_equal(total, full)


@pytest.mark.parametrize('rng_mode', ['tape', 'philox'])
def test_interleaved_band_tiles_reassemble_the_frame(rng_mode):
    """SURVEY section 8e, tiles that differ in cost: rank g of N renders every N-th band of the ray
    order (multi_gpu.Shard, clr_render_args.ray_band...).  Three such tiles - one-lane and pass-split
    programs, each as two launch chains, so that launches start inside a tile - put back by
    gather_index() equal the frame a single renderer makes, bit for bit; with per-pixel random
    numbers this needs every ray to find its index in the whole frame."""
    from raytracer_amd.multi_gpu import Shard
    size, world = (320, 240), 3
    whole = Renderer(load_scene('scene-dev', size), seed=2, rng_mode=rng_mode, split_passes=False)
    tape = whole.make_tape(0, 5)
    whole.upload_tape(tape)
    whole.render_resident(0, 5)
    want = whole.img.get()
    frame = None
    for g in range(world):
        shard = Shard('tiles', g, world, whole.n_pixels, band_blocks=2)
        part = Renderer(load_scene('scene-dev', size), seed=2, rng_mode=rng_mode, split_passes=(g == 1),
                        overlap_parts=2, **shard.renderer_arguments())
        assert len(part.part_bounds()) == 3 and part.n_pixels in (25600, 26112)
        part.upload_tape(tape)
        part.render_resident(0, 2, timed=False)
        part.render_resident(2, 3, timed=False)
        if frame is None:
            frame = np.zeros((shard.frame_rows, 4), dtype=np.float32)
        shard.my_rows(frame)[:part.n_pixels] = part.img.get()
    assert np.array_equal(bits(frame[shard.gather_index()]), bits(want))


@pytest.mark.parametrize('flat_layout', ['flat', 'flat-lanes', 'flat-ropes', 'flat-two'])
def test_flattened_octree_matches_reference_stream_and_oracle(flat_layout):
    """BASELINE config C4 at reduced size (14 400 triangles): the flattened HBM layout with the
    LDS stack (triangle tests pooled over the wave, or per lane), the reference-stream layout
    and the oracle (reference stream) agree bit for bit."""
    from oracle import oracle
    flat = Renderer(load_scene('scene-mesh:120x60:' + flat_layout, (96, 64)), seed=4)
    assert ('pool_best' in flat.program_code) == (flat_layout != 'flat-lanes')
    assert ('node_ropes' in flat.program_code) == (flat_layout == 'flat-ropes')
    assert ('inner_children' in flat.program_code) == (flat_layout == 'flat-two')
    stream = Renderer(load_scene('scene-mesh:120x60:stream', (96, 64)), seed=4)
    osc = oracle.OracleScene(stream)
    tape = flat.make_tape(0, 4)
    one = tuple(x[:1] if x is not None else None for x in tape)
    ids_f, t_f, n_f = flat.first_hit(one)
    ids_s, t_s, n_s = stream.first_hit(one)
    ids_o, t_o, n_o = osc.first_hit(one)
    knot = [i + 1 for i, o in enumerate(flat.scene.objects) if o.name == 'knot'][0]
    assert (ids_o[:, 0] == knot).mean() > 0.05
    for ids, t, n in ((ids_f, t_f, n_f), (ids_s, t_s, n_s)):
        assert np.array_equal(ids, ids_o)
        assert np.array_equal(bits(t), bits(t_o))
        assert np.array_equal(bits(n), bits(n_o))
    flat.regenerate = True          # lanes of a wave in different passes and segments
    flat.render_tape(tape)
    stream.render_tape(tape)
    oimg, _, _ = osc.render(tape)
    assert np.array_equal(bits(flat.img.get()), bits(oimg))
    assert np.array_equal(bits(stream.img.get()), bits(oimg))


def test_batching_is_additive_and_order_preserving():
    """Size-independent property at the full 1080p size: rendering 6 passes in one launch
    equals rendering them as 2 + 4, bit for bit (a lane visits its passes in order)."""
    r = Renderer(load_scene('scene-dev', (1920, 1080)), seed=5)
    tape = r.make_tape(0, 6)
    r.render_tape(tape)
    whole = r.img.get()
    r.img.fill(0)
    r.render_tape(tuple(x[:2] if x is not None else None for x in tape))
    r.render_tape(tuple(x[2:] if x is not None else None for x in tape))
    assert np.array_equal(bits(whole), bits(r.img.get()))
    # and path regeneration does not change any pixel either
    r.img.fill(0)
    r.regenerate = True
    r.render_tape(tape)
    assert np.array_equal(bits(whole), bits(r.img.get()))
    image = r.get_image()
    assert image.shape == (1080, 1920, 3) and np.isfinite(image).all()


def test_full_size_rows_match_oracle():
    """At 1080p the oracle is too slow for the whole frame; compare the first-hit AOV of the
    full frame (cheap) bit for bit."""
    r, osc = make('scene-dev', (1920, 1080))
    tape = r.make_tape(0, 1)
    ids, t, normal = r.first_hit(tape)
    oids, ot, onormal = osc.first_hit(tape)
    assert np.array_equal(ids, oids)
    assert np.array_equal(bits(t), bits(ot))


def test_ragged_image_size_uses_row_major_order():
    r, osc = make('scene-dev', (30, 22))          # not a multiple of 4 x 8
    tape = r.make_tape(0, 3)
    r.render_tape(tape)
    oimg, _, _ = osc.render(tape)
    assert np.array_equal(bits(r.img.get()), bits(oimg))
    assert r.get_image().shape == (22, 30, 3)


def test_zero_passes_is_a_no_op():
    r = Renderer(load_scene('scene-dev', (32, 16)), seed=0)
    tape = r.make_tape(0, 2)
    r.render_tape(tape)
    before = r.img.get()
    empty = tuple(x[:0] if x is not None else None for x in tape)
    r.upload_tape(tape)
    assert r.render_resident(0, 0) == 0.0
    assert np.array_equal(before, r.img.get())


def test_bad_arguments_are_rejected():
    from raytracer_amd.accelerator import ClrError, RenderArgs
    r = Renderer(load_scene('scene-dev', (32, 16)), seed=0)
    with pytest.raises(ClrError):
        r.acc.render_passes(RenderArgs(n_pixels=10, n_passes=1))


@pytest.mark.parametrize('name', SCENES + ['variant-camera', 'variant-media', 'variant-mesh', 'spectral-dev', 'spectral-stairs'])
def test_hip_against_reference_golden(name):
    """HIP directly against the fixtures made from the reference's own device code
    (tests/test_golden.py explains the fixtures and the bars)."""
    import os
    from conftest import GOLDEN_DIR
    from test_golden import flat_octree_ids, near, rows_of
    fx = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    size = tuple(int(v) for v in fx['size'])
    r = Renderer(load_scene(name, size, samples_per_pixel=4), seed=int(fx['seed']))
    assert np.array_equal(r.host_cam, fx['cam'])
    tape = (fx['pass_tape'][:1], fx['bounce_tape'][:1], rows_of(fx, slice(0, 1)))
    ids, t, normal = r.first_hit(tape)
    pixel = fx['seg0_pixel'].astype(np.int64)
    post = fx['seg0_whichobject'].astype(np.int32)
    hit = post != 0
    assert np.array_equal(ids[pixel, 0][hit], post[hit])                      # ids: bit-exact
    assert np.array_equal(ids[pixel, 1][hit], fx['seg0_which_subobject'].astype(np.int32)[hit])
    t_ref = fx['seg0_isec_dist']
    assert np.all(np.abs(t[pixel] - t_ref) <= 1e-4 * np.abs(t_ref))           # t: 1e-4 relative
    # image after all recorded passes, per-channel relative RMSE 1e-4
    img = np.zeros((r.n_pixels, 4), np.float32)
    tainted = np.zeros(r.n_pixels, bool)
    for j in range(int(fx['n_passes'])):
        _, dump = r.render_tape((fx['pass_tape'][j:j + 1], fx['bounce_tape'][j:j + 1], rows_of(fx, slice(j, j + 1))),
                                debug=True)
        tainted |= np.isin(dump[:, :, 0, 0].view(np.int32), flat_octree_ids(r)).any(axis=0)
    img = r.img.get()
    ref = fx['img'][~tainted, :3].astype(np.float64)
    got = img[~tainted, :3].astype(np.float64)
    for c in range(3):
        rmse = np.sqrt(np.mean((got[:, c] - ref[:, c]) ** 2))
        assert rmse <= 1e-4 * max(np.sqrt(np.mean(ref[:, c] ** 2)), 1e-12), (c, rmse)
    assert near(got, ref, 1e-4).all()


def test_scene_test_at_its_real_size_against_reference_golden():
    """BASELINE configuration C1 at its stated size (320 x 240, 16 passes) on the device, against
    the compact fixture the reference's own code produced (tests/test_golden.py)."""
    import os
    from conftest import GOLDEN_DIR
    from test_golden import compact_fixture_checks, compare_compact_image
    fx = np.load(os.path.join(GOLDEN_DIR, 'scene-test-320x240.npz'))
    r = Renderer(load_scene('scene-test', (320, 240), samples_per_pixel=16), seed=int(fx['seed']))
    assert r.split_passes                                   # the small-frame program form

    def render_pass(tape):
        return r.render_tape(tape, debug=True)[1]
    tainted = compact_fixture_checks('scene-test', fx, r, r.first_hit, render_pass)
    compare_compact_image(r.img.get(), fx, tainted)
    # and all 16 passes in one pass-split launch give the same sums as pass by pass
    from test_golden import rows_of
    pass_by_pass = r.img.get()
    r.img.fill(0)
    r.render_tape((fx['pass_tape'], fx['bounce_tape'], rows_of(fx)))
    assert np.array_equal(bits(r.img.get()), bits(pass_by_pass))


def test_empty_scene_renders_black():
    """No objects at all: every primary ray misses, paths end at once, nothing is accumulated."""
    from raytracer_amd.scenes.default_scenes import BoxScene
    scene = BoxScene()
    scene.objects[:] = []
    scene.image.size = (32, 16)
    r = Renderer(scene, seed=0)
    assert not r.bidirectional and r.n_param_slots == 0
    r.render_tape(r.make_tape(0, 2))
    assert not r.img.get().any()
    ids, t, _ = r.first_hit()
    assert not ids.any() and np.all(t == np.float32(scene.max_ray_length))


def test_4k_frame_first_hit_matches_oracle():
    """Largest configured frame (3840 x 2160, BASELINE C5 size): ids and distances of all
    8.3 M primary rays against the oracle, bit for bit, and one pass accumulates finite sums."""
    r, osc = make('scene-dev', (3840, 2160))
    tape = r.make_tape(0, 1)
    ids, t, _ = r.first_hit(tape)
    oids, ot, _ = osc.first_hit(tape)
    assert np.array_equal(ids, oids)
    assert np.array_equal(bits(t), bits(ot))
    r.render_tape(tape)
    img = r.img.get()
    assert np.isfinite(img).all() and img[:, :3].mean() > 0


@pytest.mark.parametrize('name,rng_mode', [('scene-dev', 'tape'), ('scene-three-objects', 'tape'),
                                           ('scene-dev-spectral', 'tape'), ('scene-dev', 'philox'),
                                           ('scene-dev-spectral', 'philox')])
def test_bench_workload_at_1080p_bit_exact_on_a_subsample(name, rng_mode):
    """BASELINE configs C3 (RGB as shipped and the spectral variant) / C2 as bench.py runs them:
    1920 x 1080, 32 passes per step, two steps back to back in the automatic launch pattern (two
    launch chains), with the broadcast tape and with per-pixel Philox numbers (regenerating form).
    The sums of every 257th ray equal the oracle's bit for bit (a pixel's path depends on its
    camera ray and the shared tape only, or on its own stream, keyed by its ray index)
    (/root/reference/shader.py:148-194 for the spectral rows)."""
    from oracle import oracle
    r = Renderer(load_scene(name, (1920, 1080)), seed=4, overlap_parts='auto', steps_in_flight='auto',
                 rng_mode=rng_mode)
    assert len(r.part_bounds()) == 3 and r.speculate and r.regenerate == (rng_mode == 'philox')
    assert r.shader.rgb == (name != 'scene-dev-spectral')
    tape = r.make_tape(0, 64)
    r.upload_tape(tape)
    for step in range(2):
        r.render_resident(32 * step, 32, timed=False)
    pixels = np.arange(0, r.n_pixels, 257)
    oimg, _, _ = oracle.OracleScene(r).render(tape, pixels=pixels)
    img = r.img.get()
    assert np.isfinite(img).all() and img[:, :3].std() > 0
    assert np.array_equal(bits(img[pixels]), bits(oimg))


@pytest.mark.parametrize('name,rng_mode', [('scene-dev', 'tape'), ('scene-three-objects', 'tape'),
                                           ('scene-dev-spectral', 'tape'), ('scene-dev', 'philox')])
def test_headline_workload_every_pixel_equals_the_oracle(name, rng_mode):
    """BASELINE's headline workload exactly as bench.py's default step runs it - scene-dev,
    1920 x 1080, one step of 32 passes in the automatic launch pattern, broadcast tape - on ALL
    2 073 600 pixels: the sums equal the oracle's bit for bit (66 M samples, a few seconds of the
    oracle on the box's CPUs), and so do first-hit ids, distances and normals of the whole frame.
    The same for C2, the spectral variant and the per-pixel random numbers."""
    from oracle import oracle
    r = Renderer(load_scene(name, (1920, 1080)), seed=11, overlap_parts='auto', steps_in_flight='auto',
                 rng_mode=rng_mode)
    assert r.n_pixels == 2073600 and r.speculate
    tape = r.make_tape(0, 32)
    r.upload_tape(tape)
    r.render_resident(0, 32, timed=False)
    osc = oracle.OracleScene(r)
    oimg, _, _ = osc.render(tape)
    img = r.img.get()
    assert img[:, :3].std() > 0
    assert np.array_equal(bits(img), bits(oimg))
    one = tuple(x[:1] if x is not None else None for x in tape)
    ids, t, normal = r.first_hit(one)
    oids, ot, onormal = osc.first_hit(one)
    assert np.array_equal(ids, oids) and np.array_equal(bits(t), bits(ot)) and np.array_equal(bits(normal), bits(onormal))


def test_4k_ray_marching_workload_matches_oracle():
    """BASELINE config C5 at its real size (scene-marchers, 3840 x 2160, 16 passes per launch as
    bench.py runs it): object ids of all 8.3 M primary rays equal the oracle's, distances within
    1e-4 relative (marching loops with libm calls in user code: the tolerance of
    test_ray_marching_scene_matches_oracle), and the sums of one 16-pass launch agree with the
    oracle's on every 257th ray to 1e-4 relative RMSE."""
    r, osc = make('scene-marchers', (3840, 2160))
    tape = r.make_tape(0, 1)
    ids, t, _ = r.first_hit(tape)
    oids, ot, _ = osc.first_hit(tape)
    assert np.array_equal(ids, oids), np.count_nonzero((ids != oids).any(axis=1))
    assert np.all(np.abs(t - ot) <= 1e-4 * np.abs(ot))
    tape = r.make_tape(1, 16)
    r.upload_tape(tape)
    r.render_resident(0, 16)
    pixels = np.arange(0, r.n_pixels, 257)
    oimg, _, _ = osc.render(tape, pixels=pixels)
    img = r.img.get()[pixels]
    assert np.isfinite(img).all()
    rel = np.sqrt(np.mean((img[:, :3] - oimg[:, :3]) ** 2)) / np.sqrt(np.mean(oimg[:, :3] ** 2))
    assert rel <= 1e-4, rel


def test_run_time_parameter_table_bit_exact():
    """Scenes with many objects read their parameters from __constant__ memory filled at run
    time instead of baked literals; same bits either way."""
    from oracle import oracle
    r = Renderer(load_scene('scene-three-objects', (64, 48)), seed=1, bake_parameters=False)
    tape = r.make_tape(0, 5)
    r.render_tape(tape)
    oimg, _, _ = oracle.OracleScene(r).render(tape)
    assert np.array_equal(bits(r.img.get()), bits(oimg))


def test_generic_launch_and_buffer_helpers():
    """Accelerator.call (clr_launch) and the array helpers, the analogue of the reference's
    per-kernel launch interface (accelerator.py:63-153): launching clr_first_hit by name gives
    the same result as the dedicated entry point."""
    r = Renderer(load_scene('scene-dev', (64, 48)), seed=3)
    acc = r.acc
    tape = r.make_tape(0, 1)
    ids_ref, t_ref, _ = r.first_hit(tape)
    ids = acc.new_array((r.n_pixels, 2), np.int32, zeros=True)
    t = acc.new_array((r.n_pixels,), np.float32)
    t.fill(-1.0)
    assert (t.get() == -1.0).all()
    normal = acc.new_vec3_array((r.n_pixels,))
    acc.call('clr_first_hit', r.n_pixels,
             (r.cam, r.dev_pass_tape, r.dev_vector_data, r.dev_integer_data),
             value_args=(np.int32(r.n_pixels), ids, t, normal))
    assert np.array_equal(ids.get(), ids_ref) and np.array_equal(t.get(), t_ref)
    assert 'clr_first_hit' in acc.profiling_info and acc.profiling_info['clr_first_hit']['n'] == 1
    # slices are views of the same device memory
    t[:16].fill(7.0)
    host = t.get()
    assert (host[:16] == 7.0).all() and np.array_equal(host[16:], t_ref[16:])
    with pytest.raises(Exception):
        acc.call('no_such_kernel', 1, ())


@pytest.mark.parametrize('name,options', [
    ('scene-dev', []),
    ('scene-three-objects', []),
    ('scene-test', []),                               # off by default for this scene: forced on here
    ('scene-dev', ['-DCLR_GUARD_LO=0x1p-7f']),        # most paths restart with the generic arithmetic
    ('scene-three-objects', ['-DCLR_GUARD_LO=0x1p-7f', '-DCLR_GUARD_HI=0x1p+2f']),
])
@pytest.mark.parametrize('regenerate', [False, True])
def test_speculative_arithmetic_and_path_restart_are_exact(name, options, regenerate):
    """The short division / square-root sequences with operand-range speculation
    (hip_templates/clr_exact.inl) give the oracle's bits, both when no path restarts and when
    a tightened range makes most of them restart."""
    from oracle import oracle
    from raytracer_amd.hip_compiler import PARITY_OPTIONS
    r = Renderer(load_scene(name, (64, 48)), seed=5, speculate=True, build_options=PARITY_OPTIONS + options)
    assert 'clr_segment_generic' in r.program_code
    r.regenerate = regenerate
    osc = oracle.OracleScene(r)
    tape = r.make_tape(0, 1)
    _, dump = r.render_tape(tape, debug=True)
    oimg, odump, _ = osc.render(tape, debug=True)
    assert np.array_equal(bits(dump), bits(odump))
    assert np.array_equal(bits(r.img.get()), bits(oimg))
    tape = r.make_tape(1, 9)
    r.render_tape(tape)
    oimg, _, _ = osc.render(tape, img=oimg)
    assert np.array_equal(bits(r.img.get()), bits(oimg))


def test_occupancy_tuning_changes_neither_results_nor_state():
    """Renderer.tune_occupancy rebuilds the program with different __launch_bounds__ hints and
    keeps the fastest; image accumulator, resident tape and random stream stay as they were."""
    from oracle import oracle
    r = Renderer(load_scene('scene-three-objects', (64, 48)), seed=9)
    tape = r.make_tape(0, 4)
    r.upload_tape(tape)
    r.render_resident(0, 2)
    before = r.img.get().copy()
    state = r.rng.get_state()[1].copy()
    best = r.tune_occupancy(candidates=(None, 3, 4), passes=1, rounds=1)
    assert best in (None, 3, 4) and set(r.occupancy_timings) == {None, 3, 4}
    assert np.array_equal(r.rng.get_state()[1], state)
    assert np.array_equal(bits(r.img.get()), bits(before))
    r.render_resident(2, 2)
    oimg, _, _ = oracle.OracleScene(r).render(tape)
    assert np.array_equal(bits(r.img.get()), bits(oimg))


@pytest.mark.parametrize('low_norm', ['min', None])
def test_device_tone_mapping_matches_host_pipeline(low_norm):
    """Renderer.get_display_image (tonemap.hip, binary64 on the device) against
    imgutils.Image._to_24bit on the same accumulator, with and without earlier sums (--append)."""
    from raytracer_amd.imgutils import Image, ColorEncodingSettings
    r = Renderer(load_scene('scene-dev', (96, 64)), seed=4)
    r.render_samples(0, 8)
    settings = ColorEncodingSettings()
    settings.low_normalization = np.min if low_norm == 'min' else None
    sums = r.get_image()
    for base in (None, sums * 0.5 + 0.01):
        want = Image(data=base, settings=settings)._to_24bit(sums.copy())
        got = r.get_display_image(settings, base=base)
        assert got.shape == want.shape and got.dtype == np.uint8
        diff = np.abs(got.astype(int) - want.astype(int))
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-5
        assert want.std() > 10                                   # a real picture, not a flat field
    # flares on a rendered frame (the light is over-exposed), then a reduction the device lacks
    settings.flares = True
    want = Image(settings=settings)._to_24bit(sums.copy())
    got = r.get_display_image(settings)
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-5
    settings.flares = False
    assert (want != Image(settings=settings)._to_24bit(sums.copy())).mean() > 0.01   # the effect is visible
    settings.brightness_reference = np.median
    with pytest.raises(NotImplementedError):
        r.get_display_image(settings)


def test_device_tone_mapping_against_the_reference_fixture():
    """tonemap.hip against what the reference's own Image._to_24bit returned (tests/golden/
    host-io.npz, made by tools/make_golden.py host-io) for five settings: default, other gamma and
    brightness, no floor, with earlier sums (--append), both, and two cases with the flare effect
    (imgutils.py:121-148; filters repeated term by term).  The device evaluates the same
    binary64 pipeline with the scaling folded into one factor: at most one count of difference
    on at most 1e-4 of the values."""
    from test_host_api import tonemap_cases
    from raytracer_amd.accelerator import Accelerator
    r = Renderer(load_scene('scene-dev', (64, 48)), seed=4)
    for name, settings, base, sums, want in tonemap_cases():
        assert sums.shape == r.img_shape + (3,)
        per_ray = Accelerator.host_vec3_array(sums[r.image_order[:, 0], r.image_order[:, 1]])
        r.img.set(per_ray)
        got = r.get_display_image(settings, base=base)
        assert got.shape == want.shape and got.dtype == np.uint8
        diff = np.abs(got.astype(int) - want.astype(int))
        assert diff.max() <= 1 and (diff > 0).mean() <= 1e-4, (name, diff.max(), (diff > 0).mean())


def test_box_light_and_two_lights_bit_exact():
    """A Parallelepiped as a second next-event light (the reference has the light
    interface for Sphere only, so this is checked against the oracle and by its effect)."""
    from oracle import oracle
    from raytracer_amd.objects import Parallelepiped, Sphere
    from raytracer_amd.scenes.default_scenes import BoxScene

    def build(with_box_light):
        s = BoxScene()
        s.image.size = (64, 48)
        s.get_object('light').bidirectional_light = True
        s.add_object(Sphere((0.3, 0.2, 0.5), 0.5), 'white')
        panel = s.add_object(Parallelepiped((-1.5, 0.5, 1.2), (0.8, 0, 0), (0, 0.6, 0), (0, 0, 0.05)), 'light', 'panel')
        panel.bidirectional_light = with_box_light
        return s

    r = Renderer(build(True), seed=11)
    assert len(r.bidirectional_light_ids) == 2
    tape = r.make_tape(0, 12)
    light_ids = tape[1][:, 0, 4, 3].view(np.int32)
    assert len(set(light_ids.tolist())) == 2                     # both lights get sampled
    _, dump = r.render_tape(tuple(x[:1] if x is not None else None for x in tape), debug=True)
    osc = oracle.OracleScene(r)
    oimg, odump, _ = osc.render(tuple(x[:1] if x is not None else None for x in tape), debug=True)
    assert np.array_equal(bits(dump), bits(odump))
    r.img.fill(0)
    r.render_tape(tape)
    oimg, _, _ = osc.render(tape)
    assert np.array_equal(bits(r.img.get()), bits(oimg))
    # same expected image with and without next-event sampling of the panel (unbiased either way)
    plain = Renderer(build(False), seed=11)
    plain.render_samples(0, 384)
    r.img.fill(0)
    r.render_samples(12, 384)
    a, b = r.get_image().mean(), plain.get_image().mean()
    assert abs(a - b) < 0.2 * b, (a, b)
    # per-pixel random numbers: the box is sampled on the device (clr_box_light_point) - HIP equals
    # the oracle bit for bit, and the converged mean agrees with the tape mode's
    p = Renderer(build(True), seed=11, rng_mode='philox')
    assert p.has_box_lights
    posc = oracle.OracleScene(p)
    ptape = p.make_tape(0, 1)
    _, dump = p.render_tape(ptape, debug=True)
    oimg, odump, _ = posc.render(ptape, debug=True)
    assert np.array_equal(bits(dump), bits(odump))
    ptape = p.make_tape(1, 8)
    p.render_tape(ptape, first_pass=1)
    oimg, _, _ = posc.render(ptape, img=oimg, first_pass=1)
    assert np.array_equal(bits(p.img.get()), bits(oimg))
    p.img.fill(0)
    p.render_samples(9, 384)
    c = p.get_image().mean()
    assert abs(c - b) < 0.2 * b, (c, b)


def test_cylinder_cone_and_polyhedron_lights():
    """SURVEY section 8 row f4, the rest: Cylinder, Cone and plane-bounded polyhedra as next-event
    lights (tape mode: the light point comes from the host, objects/convex.py).  The reference has
    no such lights, so: HIP equals the oracle bit for bit on the same tape, every light gets
    sampled, and the converged image mean equals the one without next-event sampling of those
    bodies (the estimator is unbiased either way)."""
    from oracle import oracle
    from raytracer_amd.objects import Cylinder, Cone, Tetrahedron, Sphere
    from raytracer_amd.scenes.default_scenes import BoxScene

    def build(sampled):
        s = BoxScene()
        s.image.size = (64, 48)
        s.get_object('light').bidirectional_light = True
        s.add_object(Sphere((0.3, 0.2, 0.5), 0.5), 'white')
        for tracer in (Cylinder((-1.6, 0.8, 0.0), (0, 0, 1), 0.9, 0.15), Cone((1.4, 1.2, 1.8), (0.2, 0, -1), 0.5, 0.3),
                       Tetrahedron((-0.8, -1.2, 0.4), 0.4)):
            s.add_object(tracer, 'light').bidirectional_light = sampled
        return s

    r = Renderer(build(True), seed=21)
    assert len(r.bidirectional_light_ids) == 4
    tape = r.make_tape(0, 24)
    assert len(set(tape[1][:, 0, 4, 3].view(np.int32).tolist())) == 4          # all four get sampled
    one = tuple(x[:1] if x is not None else None for x in tape)
    _, dump = r.render_tape(one, debug=True)
    osc = oracle.OracleScene(r)
    oimg, odump, _ = osc.render(one, debug=True)
    assert np.array_equal(bits(dump), bits(odump))
    r.img.fill(0)
    r.render_tape(tape)
    oimg, _, _ = osc.render(tape)
    assert np.array_equal(bits(r.img.get()), bits(oimg))
    plain = Renderer(build(False), seed=21)
    plain.render_samples(0, 512)
    r.img.fill(0)
    r.render_samples(24, 512)
    a, b = r.get_image().mean(), plain.get_image().mean()
    assert abs(a - b) < 0.15 * b, (a, b)
    with pytest.raises(NotImplementedError):                       # the device samples spheres and boxes only
        Renderer(build(True), seed=21, rng_mode='philox', device=False)


def test_pooled_and_per_lane_triangle_tests_agree_on_a_larger_mesh():
    """Size-independent property of the octree traversal at 90 000 triangles, 640x360: dealing
    the triangle tests of a wave out over its lanes changes no bit of the image, with lanes in
    lock-step or regenerating their paths (lanes of a wave then sit in different segments)."""
    images = []
    for layout in ('flat', 'flat-lanes'):
        for regenerate in (False, True):
            r = Renderer(load_scene('scene-mesh:300x150:' + layout, (640, 360)), seed=6)
            r.regenerate = regenerate
            r.render_tape(r.make_tape(0, 3))
            images.append(r.img.get())
    assert images[0][:, :3].mean() > 0
    for img in images[1:]:
        assert np.array_equal(bits(img), bits(images[0]))


@pytest.mark.parametrize('name', ['scene-dev', 'scene-three-objects', 'scene-dev-spectral'])
def test_per_pixel_random_numbers_match_oracle_and_tape_mode(name):
    """rng_mode='philox' (hip_templates/clr_rng.inl; no counterpart in the reference): HIP equals
    the oracle bit for bit, path regeneration changes nothing, launches can be split anywhere,
    and the converged image agrees with the reference-style tape mode."""
    from oracle import oracle
    r = Renderer(load_scene(name, (64, 48)), seed=3, rng_mode='philox')
    osc = oracle.OracleScene(r)
    tape = r.make_tape(0, 6)
    assert tape[1] is None
    one = tuple(x[:1] if x is not None else None for x in tape)
    _, dump = r.render_tape(one, debug=True)
    oimg, odump, _ = osc.render(one, debug=True)
    assert np.array_equal(bits(dump), bits(odump))
    assert np.array_equal(bits(r.img.get()), bits(oimg))
    rest = tuple(x[1:] if x is not None else None for x in tape)
    r.regenerate = True
    r.render_tape(rest, first_pass=1)
    oimg, _, _ = osc.render(rest, img=oimg, first_pass=1)
    assert np.array_equal(bits(r.img.get()), bits(oimg))
    # one launch for all six passes gives the same sums
    whole = Renderer(load_scene(name, (64, 48)), seed=3, rng_mode='philox')
    whole.render_tape(whole.make_tape(0, 6))
    assert np.array_equal(bits(whole.img.get()), bits(oimg))
    # statistics: 256 passes per mode, image means within a few per cent
    a = Renderer(load_scene(name, (64, 48)), seed=5, rng_mode='philox')
    b = Renderer(load_scene(name, (64, 48)), seed=5)
    a.render_samples(0, 256)
    b.render_samples(0, 256)
    ma, mb = a.get_image().mean(), b.get_image().mean()
    assert abs(ma - mb) < 0.1 * mb, (ma, mb)


def test_random_number_mode_of_program_and_launch_must_agree():
    r = Renderer(load_scene('scene-dev', (32, 16)), seed=0)
    tape = r.make_tape(0, 1)
    r.rng_mode = 'philox'                       # launch arguments now claim the other mode
    with pytest.raises(Exception) as err:
        r.render_tape(tape)
    assert 'random number mode' in str(err.value)


@pytest.mark.parametrize('name,rng_mode', [('scene-dev', 'tape'), ('scene-three-objects', 'tape'),
                                           ('scene-test', 'tape'), ('scene-dev', 'philox')])
def test_pass_split_mode_adds_in_the_same_order(name, rng_mode):
    """Small frames: the passes of a pixel are dealt out to several lanes that log their addends,
    and a second kernel sums the logs in pass order.  Same bits as the one-lane walk and as the
    oracle, for pass counts that do and do not divide by the number of groups, with and without
    path regeneration, on top of a non-zero image."""
    from oracle import oracle
    size = (40, 24)
    one = Renderer(load_scene(name, size), seed=4, rng_mode=rng_mode, split_passes=False)
    split = Renderer(load_scene(name, size), seed=4, rng_mode=rng_mode)
    assert split.split_passes and not one.split_passes
    osc = oracle.OracleScene(one)
    # start from negative zeros: the log leaves out (+0, +0, +0) addends except for the position
    # of the first, which is what turns a -0 sum into +0
    oimg = np.full((one.n_pixels, 4), -0.0, dtype=np.float32)
    for r in (one, split):
        r.img = r.acc.to_device(oimg.copy())
    first = 0
    for n, regenerate in ((7, False), (1, True), (12, True)):
        tape = one.make_tape(first, n)
        for r in (one, split):
            r.regenerate = regenerate
            r.render_tape(tape, first_pass=first)
        oimg, _, _ = osc.render(tape, img=oimg, first_pass=first)
        first += n
        assert np.array_equal(bits(one.img.get()), bits(oimg))
        assert np.array_equal(bits(split.img.get()), bits(oimg))
    assert split._split_args(12)['pass_groups'] == 12
    # the debug dump of a split program is that of the one-lane walk
    tape = one.make_tape(first, 1)
    assert np.array_equal(bits(one.render_tape(tape, debug=True, first_pass=first)[1]),
                          bits(split.render_tape(tape, debug=True, first_pass=first)[1]))


def test_pass_split_arguments_must_match_the_program():
    r = Renderer(load_scene('scene-dev', (1920, 1080)), seed=0)
    assert not r.split_passes
    tape = r.make_tape(0, 1)
    r._split_args = lambda n, debug=False, static_log=False: dict(pass_log=r.img.ptr, pass_groups=2)
    with pytest.raises(Exception) as err:
        r.render_tape(tape)
    assert 'pass-split' in str(err.value)
    small = Renderer(load_scene('scene-dev', (32, 16)), seed=0)
    small._split_args = lambda n, debug=False, static_log=False: {}
    with pytest.raises(Exception) as err:
        small.render_tape(small.make_tape(0, 1))
    assert 'pass_log' in str(err.value)


def test_triangle_box_tree_gives_the_same_image_as_the_reference_tree():
    """overlap='sat' changes which leaves list a face, not what a ray hits: 14 400 triangles,
    flattened layout, pooled tests - first hits and a 4-pass image equal the reference-rule tree's
    and the oracle's (which walks the reference stream of the reference-rule tree) bit for bit."""
    from oracle import oracle
    sat = Renderer(load_scene('scene-mesh:120x60:flat:sat', (96, 64)), seed=4)
    ref = Renderer(load_scene('scene-mesh:120x60:stream:reference', (96, 64)), seed=4)
    knot = [o for o in sat.scene.objects if o.name == 'knot'][0].tracer
    knot_ref = [o for o in ref.scene.objects if o.name == 'knot'][0].tracer
    assert knot._native['leaf_records'] < 0.3 * knot_ref._native['leaf_records']
    osc = oracle.OracleScene(ref)
    tape = sat.make_tape(0, 4)
    one = tuple(x[:1] if x is not None else None for x in tape)
    ids, t, n = sat.first_hit(one)
    ids_o, t_o, n_o = osc.first_hit(one)
    assert np.array_equal(ids, ids_o) and np.array_equal(bits(t), bits(t_o)) and np.array_equal(bits(n), bits(n_o))
    sat.render_tape(tape)
    oimg, _, _ = osc.render(tape)
    assert np.array_equal(bits(sat.img.get()), bits(oimg))


def test_one_million_triangles_pooled_per_lane_and_both_trees_agree():
    """BASELINE config C4 at its real size (1 M triangles, 1920 x 1080): size-independent
    properties.  First-hit ids, distances and normals of the full frame are bit-identical between
    the pooled and the per-lane triangle tests and between the triangle-box tree and the
    reference-rule tree, and with the next leaf found through neighbour links (ropes) instead of the
    reference's descent, or with node records that take the descent down two levels per access; a
    2-pass image is bit-identical between pooled, per-lane, ropes and two-level records."""
    results, images = [], []
    for layout, overlap in (('flat', 'sat'), ('flat-lanes', 'sat'), ('flat', 'reference'), ('flat-ropes', 'sat'),
                            ('flat-two', 'sat')):
        r = Renderer(load_scene('scene-mesh:1000x500:%s:%s' % (layout, overlap), (1920, 1080)), seed=9)
        tape = r.make_tape(0, 2)
        results.append(r.first_hit(tuple(x[:1] if x is not None else None for x in tape)))
        if overlap == 'sat':
            r.render_tape(tape)
            images.append(r.img.get())
        del r
    ids0, t0, n0 = results[0]
    knot = ids0[:, 0] == np.bincount(ids0[:, 0]).argmax()      # may be a wall; just need both classes
    assert 0 < knot.mean() < 1 and len(np.unique(ids0[:, 1])) > 10000    # thousands of distinct faces seen
    for ids, t, n in results[1:]:
        assert np.array_equal(ids, ids0) and np.array_equal(bits(t), bits(t0)) and np.array_equal(bits(n), bits(n0))
    assert images[0][:, :3].mean() > 0 and len(images) == 4
    for img in images[1:]:
        assert np.array_equal(bits(images[0]), bits(img))


def test_one_million_triangles_through_the_mesh_reader_against_the_oracle(tmp_path):
    """BASELINE config C4 at its real size, fed the way SURVEY section 8d states it: the generated
    1000 x 500 knot is written as an OFF file, parsed by the native reader (csrc/mesh_io.cpp, the
    counterpart of /root/reference/mesh_formats.py:14-40), filed in the triangle-box tree and rendered
    at 1920 x 1080 as bench.py does (pass-split program, two steps in flight).  On every 257th ray, and
    then on every ray of the frame, the
    first-hit ids, distances and normals and the sums of two 2-pass steps equal, bit for bit, what
    the oracle finds walking the reference's own tree (bounding-sphere rule) in the reference's
    stream layout (/root/reference/cl_templates/objects/Octree.cl:32-168)."""
    from oracle import oracle
    from raytracer_amd.scenes.library import scene_mesh, torus_knot_mesh, mesh_through_off_file
    vertices, faces = torus_knot_mesh(1000, 500)
    v, f, seconds = mesh_through_off_file(vertices, faces, str(tmp_path / 'knot.off'))
    print('parsed %d vertices, %d faces of OFF text in %.2f s' % (len(v), len(f), seconds))
    assert v.dtype == np.float64 and np.array_equal(v.astype(np.float32), vertices) and np.array_equal(f, faces)
    r = Renderer(scene_mesh(mesh=(v, f), layout='flat', overlap='sat'), seed=9,
                 overlap_parts='auto', steps_in_flight='auto')
    assert r.img_shape == (1080, 1920) and r.split_passes and r.steps_in_flight == 2
    ref = Renderer(scene_mesh(mesh=(v, f), layout='stream', overlap='reference'), seed=9, device=False)
    osc = oracle.OracleScene(ref)
    tape = r.make_tape(0, 4)
    one = tuple(x[:1] if x is not None else None for x in tape)
    pixels = np.arange(0, r.n_pixels, 257)
    ids, t, normal = r.first_hit(one)
    oids, ot, onormal = osc.first_hit(one, pixels=pixels)
    knot = [i + 1 for i, o in enumerate(r.scene.objects) if o.name == 'knot'][0]
    on_knot = oids[:, 0] == knot
    assert 0.05 < on_knot.mean() < 0.9 and len(np.unique(oids[on_knot, 1])) > 500
    assert np.array_equal(ids[pixels], oids)
    assert np.array_equal(bits(t[pixels]), bits(ot)) and np.array_equal(bits(normal[pixels]), bits(onormal))
    r.upload_tape(tape)
    for step in range(2):
        r.render_resident(2 * step, 2, timed=False)
    oimg, _, _ = osc.render(tape, pixels=pixels)
    img = r.img.get()
    assert np.isfinite(img).all() and img[:, :3].mean() > 0
    assert np.array_equal(bits(img[pixels]), bits(oimg))
    # ... and the whole frame: first hits of all 2 073 600 rays and the sums of the four passes
    oids, ot, onormal = osc.first_hit(one)
    assert np.array_equal(ids, oids) and np.array_equal(bits(t), bits(ot)) and np.array_equal(bits(normal), bits(onormal))
    oimg, _, _ = osc.render(tape)
    assert np.array_equal(bits(img), bits(oimg))


def test_several_light_points_per_segment():
    """SURVEY section 8 row f4, the rest: `Renderer(light_samples=k)` estimates the direct light of a
    diffuse hit with k points of the segment's light (one visibility sweep each) instead of one.
    The reference has k = 1 only, so: HIP equals the oracle bit for bit (tape and per-pixel
    numbers, one-lane, pass-split and regenerating programs, debug dump), k = 1 leaves the
    program as it was, and the converged mean equals the k = 1 mean.  (What it buys depends on the
    scene: the light point is one noise source among several; on the box scenes here the frame
    noise at equal passes moves by a few per cent only, soft shadows of large lights gain most.)"""
    from oracle import oracle
    from raytracer_amd.objects import Parallelepiped, Sphere
    from raytracer_amd.scenes.default_scenes import BoxScene

    def build():
        s = BoxScene()
        s.image.size = (64, 48)
        s.get_object('light').bidirectional_light = True
        s.add_object(Sphere((0.3, 0.2, 0.5), 0.5), 'white')
        s.add_object(Parallelepiped((-1.5, 0.5, 1.2), (0.8, 0, 0), (0, 0.6, 0), (0, 0, 0.05)), 'light',
                     'panel').bidirectional_light = True
        return s

    assert 'vec3 estimate' not in Renderer(build(), seed=5, device=False).program_code
    for mode in ('tape', 'philox'):
        r = Renderer(build(), seed=5, rng_mode=mode, light_samples=3)
        osc = oracle.OracleScene(r)
        one = r.make_tape(0, 1)
        _, dump = r.render_tape(one, debug=True)
        oimg, odump, _ = osc.render(one, debug=True)
        assert np.array_equal(bits(dump), bits(odump)), mode
        tape = r.make_tape(1, 9)
        if mode == 'tape':
            assert tape[1].shape[2:] == (9, 4)
        r.render_tape(tape, first_pass=1)
        oimg, _, _ = osc.render(tape, img=oimg, first_pass=1)
        assert np.array_equal(bits(r.img.get()), bits(oimg)), mode
        assert oimg[:, :3].max() > 0
        # the other program forms add the same numbers
        for kw in (dict(split_passes=True), dict(regenerate=not r.regenerate)):
            other = Renderer(build(), seed=5, rng_mode=mode, light_samples=3, **kw)
            other.render_tape(one)
            other.render_tape(tape, first_pass=1)
            assert np.array_equal(bits(other.img.get()), bits(oimg)), (mode, kw)

    # statistics, per-pixel numbers: the estimator stays unbiased (same converged mean)
    def rendering(k, seed):
        p = Renderer(build(), seed=seed, rng_mode='philox', light_samples=k)
        p.render_samples(0, 256)
        return p.get_image() / 256

    one, eight = rendering(1, 29), rendering(8, 30)
    assert abs(eight.mean() - one.mean()) < 0.02 * one.mean(), (eight.mean(), one.mean())


def test_find_non_negative_and_device_memcpy():
    """Accelerator.find_non_negative (reference accelerator.py:196-222: the stable compaction behind
    the culler) and device_memcpy (:224-229) through the C ABI, against numpy: sizes around the tile
    (4096) and block (256) boundaries, all kept, none kept, a million elements, and the same
    context reused with a smaller input after a larger one."""
    from raytracer_amd.accelerator import Accelerator
    acc = Accelerator(0)
    rng = np.random.RandomState(8)
    for n in (1, 63, 64, 255, 256, 257, 4095, 4096, 4097, 12289, 1 << 20, (1 << 20) + 3, 1000):
        for keep in (0.5, 0.02, 1.0, 0.0):
            values = rng.randint(0, 1 << 30, size=n).astype(np.int32)
            values[rng.random_sample(n) >= keep] = -1
            if keep == 0.5:
                values[rng.randint(n)] = 0                      # zero counts as non-negative
            src = acc.to_device(values)
            dst = acc.new_array((n,), np.int32)
            dst.fill(-7)
            count = acc.find_non_negative(src, dst, n)
            want = values[values >= 0]
            assert count == want.size, (n, keep, count, want.size)
            got = dst.get()
            assert np.array_equal(got[:count], want), (n, keep)
            assert (got[count:] == -7).all()                    # nothing written behind the result
    # only the first n elements take part
    values = np.arange(-5, 5000, dtype=np.int32)
    src, dst = acc.to_device(values), acc.new_array((values.size,), np.int32)
    assert acc.find_non_negative(src, dst, 100) == 95
    assert np.array_equal(dst.get()[:95], np.arange(0, 95))
    assert acc.find_non_negative(src, dst, 0) == 0
    with pytest.raises(TypeError):
        acc.find_non_negative(acc.new_array((4,), np.float32), dst, 4)
    # device_memcpy: whole array, or the first n 32-bit values
    copy = acc.new_array((values.size,), np.int32)
    copy.fill(0)
    acc.device_memcpy(copy, src, 10)
    assert np.array_equal(copy.get()[:10], values[:10]) and (copy.get()[10:] == 0).all()
    acc.device_memcpy(copy, src)
    assert np.array_equal(copy.get(), values)


@pytest.mark.parametrize('rng_mode', ['tape', 'philox'])
def test_launch_chains_on_slices_of_the_frame_change_no_pixel(rng_mode):
    """Renderer(overlap_parts=k): every launch goes out as k launches on contiguous slices of the
    ray order, each slice on a stream of its own (clr_stream_create), un-timed launches not waited
    for.  Same image bit for bit as one launch per step, also for a tile of the frame (ray_range)
    and when reads on the default stream follow directly."""
    scene = load_scene('scene-dev', (320, 240))
    one = Renderer(scene, seed=9, rng_mode=rng_mode, split_passes=False)
    tape = one.make_tape(0, 12)
    one.upload_tape(tape)
    for first in (0, 4, 8):
        one.render_resident(first, 4)
    want = one.img.get()
    for parts, ray_range in ((3, None), (2, (256 * 40, 256 * 100))):      # ray_range = (first ray, count)
        r = Renderer(load_scene('scene-dev', (320, 240)), seed=9, rng_mode=rng_mode, split_passes=False,
                     overlap_parts=parts, ray_range=ray_range)
        assert len(r.part_bounds()) == parts + 1
        r.upload_tape(tape)
        for first in (0, 4, 8):
            assert r.render_resident(first, 4, timed=False) is None
        got = r.img.get()                                      # a copy on the default stream: waits for the slices
        lo, hi = (ray_range[0], ray_range[0] + ray_range[1]) if ray_range else (0, one.n_pixels)
        assert np.array_equal(bits(got), bits(want[lo:hi])), (parts, ray_range)
        assert len(r._part_streams) == parts
        # timed launches report the time of the whole step
        r.img.fill(0)
        ms = r.render_resident(0, 4)
        assert ms > 0
    # a pass-split program: every slice has its own part of the addend log
    r = Renderer(load_scene('scene-dev', (320, 240)), seed=9, rng_mode=rng_mode, split_passes=True, overlap_parts=3)
    r.upload_tape(tape)
    for first in (0, 4, 8):
        r.render_resident(first, 4, timed=False)
    assert np.array_equal(bits(r.img.get()), bits(want))
    # ... and with several launches in flight: launch k + 1 logs into another buffer on another
    # stream while launch k runs, the sums are added in launch order (clr_render_args.reduce_after)
    for parts, in_flight, steps in ((2, 3, ((0, 2), (2, 2), (4, 1), (5, 3), (8, 4))), (1, 2, ((0, 4), (4, 4), (8, 4)))):
        r = Renderer(load_scene('scene-dev', (320, 240)), seed=9, rng_mode=rng_mode, split_passes=True,
                     overlap_parts=parts, steps_in_flight=in_flight)
        r.upload_tape(tape)
        for first, count in steps:
            r.render_resident(first, count, timed=False)
        assert np.array_equal(bits(r.img.get()), bits(want)), (parts, in_flight)
        assert len(r._part_streams) == parts * in_flight and len(r._pass_logs) == in_flight
        r.img.fill(0)
        assert r.render_resident(0, 4) > 0
    # a one-lane program has no log: the option changes nothing
    r = Renderer(load_scene('scene-dev', (320, 240)), seed=9, rng_mode=rng_mode, split_passes=False,
                 overlap_parts=2, steps_in_flight=3)
    r.upload_tape(tape)
    for first in (0, 4, 8):
        r.render_resident(first, 4, timed=False)
    assert np.array_equal(bits(r.img.get()), bits(want)) and len(r._part_streams) == 2
    # a frame too small to cut stays on one stream
    small = Renderer(load_scene('scene-dev', (64, 48)), seed=9, overlap_parts=4)
    assert small.part_bounds() == [0, 64 * 48]


def test_sphere_miss_test_agrees_with_the_full_test_on_grazing_rays():
    """Sphere.can_hit_expression (static.hip, clr_unit_sphere_may_be_hit) drops a ray before the
    normalisation only where the full test cannot find a hit.  Rays aimed at the silhouette of
    three spheres (one of them scaled to an ellipsoid) within 0, +-1e-7 ... +-1e-3 radii of the
    tangent, and rays that leave at right angles to the line to the centre, +- the same: first-hit
    object, sub-object, distance and normal equal the oracle's full test bit for bit."""
    from oracle import oracle
    from raytracer_amd.objects import Sphere
    from raytracer_amd.scenes.default_scenes import BoxScene
    s = BoxScene()
    s.image.size = (256, 96)
    egg = Sphere((0.4, 1.0, 0.6), 0.3)
    egg.linear_transform(scaling=(1, 1, 2))
    bodies = [(np.array((-1.0, 0.5, 1.0)), 0.7, Sphere((-1.0, 0.5, 1.0), 0.7)),
              (np.array((1.5, 2.0, 2.5)), 0.05, Sphere((1.5, 2.0, 2.5), 0.05)),
              (np.array((0.4, 1.0, 0.6)), 0.3, egg)]
    for _, _, tracer in bodies:
        s.add_object(tracer, 'white')
    r = Renderer(s, seed=2, split_passes=False)
    origin = np.array(s.camera_position, dtype=np.float64)
    rng = np.random.RandomState(77)
    offsets = np.concatenate([[0.0], np.ravel([[d, -d] for d in (1e-7, 3e-7, 1e-6, 1e-5, 1e-4, 1e-3)])])
    rays = []
    while len(rays) < r.n_pixels:
        centre, radius, _ = bodies[len(rays) % 3]
        to_centre = centre - origin
        dist = np.linalg.norm(to_centre)
        axis = to_centre / dist
        side = np.cross(axis, rng.normal(size=3))
        side /= np.linalg.norm(side)
        delta = offsets[rng.randint(len(offsets))]
        if len(rays) % 2:
            # tangent to the sphere (for the ellipsoid: near its silhouette), moved by delta radii
            sin_a = min(radius * (1.0 + delta) / dist, 1.0)
            direction = axis * np.sqrt(1.0 - sin_a ** 2) + side * sin_a
        else:
            # at right angles to the line to the centre, tilted by delta: heading away or not
            direction = side + axis * delta
        rays.append(direction / np.linalg.norm(direction))
    cam = np.zeros((r.n_pixels, 4), dtype=np.float32)
    cam[:, :3] = np.array(rays)[:r.n_pixels]
    r.host_cam = cam
    r.cam.set(cam)
    tape = r.make_tape(0, 1)
    tape[0][0, 0:3, :] = 0.0
    tape[0][0, 0:3, 0:3] = np.eye(3)                  # no jitter: the crafted directions as they are
    tape[0][0, 3, :] = 0.0
    tape[0][0, 4, 0:3] = origin
    ids, t, normal = r.first_hit(tape)
    oids, ot, onormal = oracle.OracleScene(r).first_hit(tape)
    assert np.array_equal(ids, oids)
    assert np.array_equal(bits(t), bits(ot)) and np.array_equal(bits(normal), bits(onormal))
    tracers = [tracer for _, _, tracer in bodies]
    hit = np.isin(oids[:, 0], [i + 1 for i, obj in enumerate(s.objects) if obj.tracer in tracers])
    assert 0.15 < hit.mean() < 0.6, hit.mean()         # the family straddles the silhouettes
    # and the rest of the path: a few passes of the full estimator on the same rays
    more = r.make_tape(1, 3)
    r.render_tape(more)
    oimg, _, _ = oracle.OracleScene(r).render(more)
    assert np.array_equal(bits(r.img.get()), bits(oimg))


def test_ball_cull_of_plane_bounded_bodies_agrees_with_the_full_test():
    """FixedConvexIntersection.can_hit_expression (static.hip, clr_ball_may_be_hit): a body bounded
    by planes is dropped before the ray is normalised only where the reference's component tests
    cannot find a hit.  Three bodies (a dodecahedron, a tetrahedron turned and stretched, a small
    cube) and rays aimed at the silhouette of each one's culling ball within 0, +-1e-7 ... +-1e-2
    radii of the tangent, at the silhouette of the ball that just holds the body, into the body, and
    rays that leave at right angles to the line to the centre, +- the same: first-hit object, face, distance
    and normal equal the oracle's full test bit for bit; so do a few passes of the estimator."""
    import re
    from oracle import oracle
    from raytracer_amd.objects import Dodecahedron, Tetrahedron, Cube
    from raytracer_amd.objects.convex import CULL_BALL_FACTOR, CULL_BALL_SLACK
    from raytracer_amd.scenes.default_scenes import BoxScene
    s = BoxScene()
    s.image.size = (256, 96)
    turned = Tetrahedron((0.6, 1.2, 0.8), 0.4)
    turned.rotate(axis=(1, 2, 3), deg=37)
    turned.linear_transform(scaling=(1.0, 1.3, 0.8))
    bodies = [Dodecahedron((-1.0, 0.5, 1.0), 0.6), turned, Cube((1.5, 2.0, 2.2), 0.1)]
    for tracer in bodies:
        assert 'clr_ball_may_be_hit' in tracer.can_hit_expression
        s.add_object(tracer, 'white')
    r = Renderer(s, seed=2, split_passes=False)
    assert r.program_code.count('clr_ball_may_be_hit(') >= 7      # the definition, trace and shadow step of three classes
    origin = np.array(s.camera_position, dtype=np.float64)
    rng = np.random.RandomState(78)
    offsets = np.concatenate([[0.0], np.ravel([[d, -d] for d in (1e-7, 3e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2)])])
    rays = []
    while len(rays) < r.n_pixels:
        tracer = bodies[len(rays) % 3]
        rho = np.sqrt(float(re.search(r', ([0-9.e+-]+)f, isec_dist', tracer.can_hit_expression).group(1)))
        to_world = tracer.global_to_tracer_coordinate_transform()
        linear = np.asarray(to_world.linear, dtype=np.float64)
        centre = np.ravel(np.asarray(to_world.translation, dtype=np.float64))
        # a world direction whose local image has length 1: scale the aim so that the LOCAL ray is
        # tangent to the local ball (the bodies are stretched: work in tracer coordinates)
        inverse = np.linalg.inv(linear)
        o_local = inverse.dot(origin - centre)
        dist = np.linalg.norm(o_local)
        axis = -o_local / dist
        side = np.cross(axis, rng.normal(size=3))
        side /= np.linalg.norm(side)
        delta = offsets[rng.randint(len(offsets))]
        kind = (len(rays) // 3) % 4
        if kind == 3:      # into the body, anywhere up to the ball that holds it
            sin_a = min(rho / np.sqrt(CULL_BALL_SLACK) / CULL_BALL_FACTOR * rng.uniform(0.0, 1.0) / dist, 1.0)
            local = axis * np.sqrt(1.0 - sin_a ** 2) + side * sin_a
        elif kind == 0:    # tangent to the culling ball, moved by delta radii
            sin_a = min(rho * (1.0 + delta) / dist, 1.0)
            local = axis * np.sqrt(1.0 - sin_a ** 2) + side * sin_a
        elif kind == 1:    # tangent to the ball that just holds the body
            sin_a = min(rho / np.sqrt(CULL_BALL_SLACK) / CULL_BALL_FACTOR * (1.0 + delta) / dist, 1.0)
            local = axis * np.sqrt(1.0 - sin_a ** 2) + side * sin_a
        else:              # at right angles to the line to the centre, tilted by delta
            local = side + axis * delta
        direction = linear.dot(local)
        rays.append(direction / np.linalg.norm(direction))
    cam = np.zeros((r.n_pixels, 4), dtype=np.float32)
    cam[:, :3] = np.array(rays)[:r.n_pixels]
    r.host_cam = cam
    r.cam.set(cam)
    tape = r.make_tape(0, 1)
    tape[0][0, 0:3, :] = 0.0
    tape[0][0, 0:3, 0:3] = np.eye(3)                  # no jitter: the crafted directions as they are
    tape[0][0, 3, :] = 0.0
    tape[0][0, 4, 0:3] = origin
    ids, t, normal = r.first_hit(tape)
    oids, ot, onormal = oracle.OracleScene(r).first_hit(tape)
    assert np.array_equal(ids, oids)
    assert np.array_equal(bits(t), bits(ot)) and np.array_equal(bits(normal), bits(onormal))
    hit = np.isin(oids[:, 0], [i + 1 for i, obj in enumerate(s.objects) if obj.tracer in bodies])
    assert 0.1 < hit.mean() < 0.3, hit.mean()          # a quarter of the rays are aimed into the bodies
    more = r.make_tape(1, 3)
    r.render_tape(more)
    oimg, _, _ = oracle.OracleScene(r).render(more)
    assert np.array_equal(bits(r.img.get()), bits(oimg))


def test_plane_distance_test_agrees_with_the_full_test_on_near_ties():
    """HalfSpace.can_hit_expression, second part (static.hip, clr_plane_before): a plane is dropped
    before the ray is normalised only where the full test cannot put it below the distance that
    counts.  Two parallel tilted planes whose offsets differ by 0, +-1, +-2, +-4 ... units in the
    last place, in either object order: which of the two a ray hits first (and at which distance)
    equals the oracle's full test bit for bit over a frame of rays, and so does a rendered image
    (the same planes between the hit points and the light)."""
    from oracle import oracle
    from raytracer_amd.objects import HalfSpace, Sphere
    from raytracer_amd.scenes.default_scenes import BoxScene
    base = np.float32(2.0)
    for ulps in (0, 1, -1, 2, -4, 16, -64):
        other = float(np.nextafter(base, np.float32(np.inf if ulps > 0 else -np.inf)) if abs(ulps) == 1
                      else base + np.float32(ulps) * np.spacing(base))
        for swap in (False, True):
            s = BoxScene()
            s.image.size = (128, 64)
            s.get_object('light').bidirectional_light = True
            offsets = (other, float(base)) if swap else (float(base), other)
            pair = [s.add_object(HalfSpace((0.3, 1.0, 0.2), h), 'white') for h in offsets]
            s.add_object(Sphere((0.0, -3.0, 0.8), 0.4), 'white')
            r = Renderer(s, seed=6, split_passes=False)
            pair_ids = [s.objects.index(obj) + 1 for obj in pair]      # the renderer groups objects by class
            tape = r.make_tape(0, 3)
            ids, t, normal = r.first_hit(tape)
            osc = oracle.OracleScene(r)
            oids, ot, onormal = osc.first_hit(tape)
            assert np.array_equal(ids, oids), (ulps, swap)
            assert np.array_equal(bits(t), bits(ot)) and np.array_equal(bits(normal), bits(onormal)), (ulps, swap)
            planes = np.isin(oids[:, 0], pair_ids)
            assert planes.mean() > 0.3, (planes.mean(), np.unique(oids[:, 0], return_counts=True))
            if abs(ulps) >= 16:       # far enough apart for one plane to win everywhere
                assert len(set(oids[planes, 0].tolist())) == 1
            r.render_tape(tape)
            oimg, _, _ = osc.render(tape)
            assert np.array_equal(bits(r.img.get()), bits(oimg)), (ulps, swap)


def signed_zero_scene():
    """Axis-aligned bodies around a ray origin in the middle of the room, off their centre planes,
    and a frame of rays with one component exactly +0 or -0 (tests below and in test_golden-style
    CPU rehearsal)."""
    from raytracer_amd.objects import Parallelepiped, Cube, Cylinder
    from raytracer_amd.scenes.default_scenes import BoxScene
    s = BoxScene()
    s.image.size = (128, 64)
    s.get_object('light').bidirectional_light = True
    origin = np.array((0.2, 0.2, 1.9))
    s.add_object(Cube((1.3, 1.2, 2.0), 0.8), 'glass')
    s.add_object(Cube((-1.1, -1.3, 2.0), 0.8), 'white')
    s.add_object(Parallelepiped((-1.6, 1.0, 1.6), (0.8, 0, 0), (0, 0.7, 0), (0, 0, 0.8)), 'white')
    s.add_object(Parallelepiped((0.9, -1.7, 1.6), (0.8, 0, 0), (0, 0.7, 0), (0, 0, 0.8)), 'mirror')
    s.add_object(Parallelepiped((-0.1, 1.5, 1.5), (0.7, 0, 0), (0, 0.5, 0), (0, 0, 0.9)), 'white')
    s.add_object(Parallelepiped((-0.1, -2.0, 1.5), (0.7, 0, 0), (0, 0.5, 0), (0, 0, 0.9)), 'glass')
    s.add_object(Cylinder((1.6, -0.2, 1.5), (0, 0, 1), 0.9, 0.3), 'mirror')
    s.add_object(Cylinder((-1.5, 0.1, 1.5), (0, 0, 1), 0.9, 0.3), 'white')
    bodies = (Parallelepiped, Cube, Cylinder)
    rng = np.random.RandomState(5)
    n = s.image.size[0] * s.image.size[1]
    cam = np.zeros((n, 4), dtype=np.float32)
    for k in range(n):
        d = rng.normal(size=3)
        zero_axis = (0, 2, 2, 2)[k % 4]
        d[zero_axis] = 0.0
        d /= np.linalg.norm(d)
        cam[k, :3] = d
        if k % 2:
            cam[k, zero_axis] = np.float32(-0.0)
    return s, origin, cam, bodies


def fixed_camera_tape(r, n_passes, origin):
    """A tape whose passes use the camera rays as they are (no jitter, no depth of field) from `origin`."""
    tape = r.make_tape(0, n_passes)
    tape[0][:, 0:3, :] = 0.0
    tape[0][:, 0:3, 0:3] = np.eye(3)
    tape[0][:, 3, :] = 0.0
    tape[0][:, 4, 0:3] = origin
    return tape


def test_rays_with_signed_zero_components_parallel_to_box_faces():
    """Rays exactly parallel to the faces of axis-aligned bodies, with the parallel component +0 or
    -0: the slope of such a face's plane is a signed zero and the reference divides by it
    (LayerComponent.cl:3-24, HalfSpaceComponent.cl:3-15: interval ends of +-inf, so that the sign of
    the zero decides between hit and miss).  The speculative arithmetic leaves out products with
    the zero components of a face normal (clr_dot_lit), which alone would give the zero another
    sign; it never gets that far, because a direction with a zero component fails the operand guard
    (its square is tracked, clr_make_direction) and the path is walked again with the generic
    arithmetic.  First hits, per-segment state and a rendered image equal the oracle's bit for bit."""
    from oracle import oracle
    s, origin, cam, bodies = signed_zero_scene()
    r = Renderer(s, seed=3, split_passes=False)
    assert r.speculate and r.bake_parameters
    assert np.signbit(cam[1::2]).any() and (cam[:, :3] == 0).sum() >= r.n_pixels
    r.host_cam = cam
    r.cam.set(cam)
    tape = fixed_camera_tape(r, 4, origin)
    one = tuple(x[:1] if x is not None else None for x in tape)
    osc = oracle.OracleScene(r)
    ids, t, normal = r.first_hit(one)
    oids, ot, onormal = osc.first_hit(one)
    assert np.array_equal(ids, oids)
    assert np.array_equal(bits(t), bits(ot)) and np.array_equal(bits(normal), bits(onormal))
    body_ids = [i + 1 for i, o in enumerate(s.objects) if isinstance(o.tracer, bodies)]
    on_body = np.isin(oids[:, 0], body_ids)
    assert 0.03 < on_body.mean() < 0.9
    _, dump = r.render_tape(one, debug=True)
    oimg, odump, _ = osc.render(one, debug=True)
    assert np.array_equal(bits(dump), bits(odump))
    r.img.fill(0)
    r.render_tape(tape)
    oimg, _, _ = osc.render(tape)
    assert np.array_equal(bits(r.img.get()), bits(oimg))


@pytest.mark.parametrize('name,size', [('scene-dev', (96, 64)), ('scene-three-objects', (96, 64)),
                                       ('scene-dev-spectral', (64, 48)), ('scene-test', (64, 40)), ('scene-dev', (30, 22))])
def test_sorted_paths_change_no_pixel(name, size):
    """Renderer(sorted_paths=True) (main.hip, sorted form: per-pixel random numbers, the paths of a
    block change lanes before every segment - north_star's ray compaction against divergence, the
    role of the reference's per-bounce compaction, renderer.py:415-431).  A segment is a pure
    function of (pixel, pass, segment) and a pixel's sums are added in the order of its one live
    path, so the image equals the oracle's bit for bit: whole frames (also one that does not fill
    its last block), several launches, a tile of the frame, launch chains, the speculative
    arithmetic with most paths restarting."""
    from oracle import oracle
    from raytracer_amd.hip_compiler import PARITY_OPTIONS
    r = Renderer(load_scene(name, size), seed=3, rng_mode='philox', split_passes=False, sorted_paths=True)
    assert 'clr_path_words' in r.program_code and r.regenerate
    osc = oracle.OracleScene(r)
    tape = r.make_tape(0, 9)
    r.render_tape(tuple(x[:4] if x is not None else None for x in tape))
    r.render_tape(tuple(x[4:] if x is not None else None for x in tape), first_pass=4)
    oimg, _, _ = osc.render(tape)
    img = r.img.get()
    assert np.isfinite(img).all() and img[:, :3].max() > 0
    assert np.array_equal(bits(img), bits(oimg))
    # the debug dump of such a program comes from the unsorted walk
    one = tuple(x[:1] if x is not None else None for x in tape)
    _, dump = r.render_tape(one, debug=True)
    _, odump, _ = osc.render(one, debug=True)
    assert np.array_equal(bits(dump), bits(odump))
    if name != 'scene-dev' or size != (96, 64):
        return
    # a tile made of bands of the frame, as two launch chains
    from raytracer_amd.multi_gpu import Shard
    shard = Shard('tiles', 1, 2, r.n_pixels, band_blocks=2)
    part = Renderer(load_scene(name, size), seed=3, rng_mode='philox', split_passes=False, sorted_paths=True,
                    overlap_parts=2, **shard.renderer_arguments())
    part.n_pixels >= 8192 or part.part_bounds()
    part.upload_tape(tape)
    part.render_resident(0, 9, timed=False)
    assert np.array_equal(bits(part.img.get()), bits(oimg[part.ray_index]))
    # most paths leave the range of the short sequences and start again
    tight = Renderer(load_scene(name, size), seed=3, rng_mode='philox', split_passes=False, sorted_paths=True,
                     speculate=True, build_options=PARITY_OPTIONS + ['-DCLR_GUARD_LO=0x1p-7f'])
    tight.render_tape(tape)
    assert np.array_equal(bits(tight.img.get()), bits(oimg))
