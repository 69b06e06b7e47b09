"""Render controller: packs the scene for the device, generates the random tape and drives
the fused render kernel.

Host-side mirror of the reference's ``Renderer`` (/root/reference/renderer.py:12-518) with
the same public surface (``Renderer(scene, args)``, ``render_sample(i)``, ``get_image()``,
``rays_per_sample()``, ``acc``).  What the reference does with ~28 blocking kernel launches
per bounce, this class does with one launch per *batch of passes*
(``render_samples(first, count)``): the per-pass camera sample and the per-bounce random
numbers (which the reference draws on the host and broadcasts to every pixel) are drawn in
the same order, stored in a tape and replayed by the device.
"""
import itertools
import math
import os
import numpy as np

from . import geometry as camera
from . import geometry as utils
from .hip_compiler import Compiler, PARITY_OPTIONS
from .accelerator import Accelerator, RenderArgs

PASS_RECORD = 6      # float4 per pass in the tape
BOUNCE_RECORD = 5    # float4 per path segment in the tape (+ 2 per further light point, light_samples > 1)
DEBUG_RECORD = 6     # float4 per (segment, pixel) in the debug dump


# lanes that fill an MI355X: 256 CUs x 4 SIMDs x 6 waves x 64
FULL_GPU_LANES = 256 * 4 * 6 * 64
# lanes a pass-split launch aims for (several rounds of waves even out the tail)
SPLIT_LANES = 4 * FULL_GPU_LANES


def pass_groups_for(n_pixels, n_passes, at_least=1):
    """Lanes per pixel of a pass-split launch: enough for SPLIT_LANES lanes (and `at_least`), at
    most one group per pass, and no more groups than ceil(n_passes / passes per group) can use."""
    if n_passes <= 0:
        return 1
    groups = max(1, min(n_passes, max(at_least, -(-SPLIT_LANES // n_pixels))))
    per_group = -(-n_passes // groups)
    return -(-n_passes // per_group)


def unpermute_rays(per_ray, image_order, img_shape):
    """(n, C) values in ray order -> (H, W, C) image; image_order maps ray -> (row, col)."""
    out = np.zeros(tuple(img_shape) + per_ray.shape[1:], dtype=per_ray.dtype)
    out[image_order[:, 0], image_order[:, 1]] = per_ray
    return out


LOG_BYTES_LIMIT = 1 << 33       # addend log of a pass-split launch: 8 GB at most


def log_passes_limit(n_pixels, max_bounces, limit=LOG_BYTES_LIMIT):
    """Passes whose addend log (1 + 6 max_bounces floats per pass and pixel) fits in `limit` bytes."""
    return max(1, limit // ((1 + 6 * max_bounces) * n_pixels * 4))


SPILL_MARGIN = 0.03               # what a build with many more spilled registers has to win by
SPILL_MARGIN_PER_REGISTER = 0.004  # ... and one with a few more: this much per register
TUNE_STEPS = 3                    # launches back to back per timing sample (the shipped launch pattern)


def pick_occupancy_hint(timings, scratch):
    """The fastest candidate among those that spill the fewest registers, unless one that spills
    more beats it by its margin: SPILL_MARGIN_PER_REGISTER for every register it spills more, at
    most SPILL_MARGIN.  (Two spilled registers at eight waves on scene-dev: 3.5 % faster for 1.2 x
    the algorithmic HBM traffic, profiles/r03_dev_h8_clr_render.json; dozens of spilled registers
    move several times the algorithmic traffic for noise-level gains.)"""
    least = min(scratch.values())
    lean = min((h for h in timings if scratch[h] == least), key=timings.get)
    best = lean
    for h in sorted(timings, key=timings.get):
        margin = min(SPILL_MARGIN, SPILL_MARGIN_PER_REGISTER * (scratch[h] - least))
        if timings[h] <= (1.0 - margin) * timings[lean]:
            best = h
            break
    return best


class Renderer:

    def __init__(self, scene, args=None, seed=None, device=True, build_options=None,
                 device_index=0, bake_parameters=None, ray_range=None,
                 min_waves_per_simd=None, speculate=None, rng_mode='tape', split_passes=None,
                 regenerate=None, light_samples=1, overlap_parts=1, steps_in_flight=1, outline=None, ray_bands=None, variants=(), sorted_paths=None):
        self.scene = scene
        # development: names of code-generation variants under measurement; a template asks with
        # {% if 'name' in renderer.variants %} (tools/bench_variants.py times them side by side)
        self.variants = frozenset(variants) | frozenset(filter(None, os.environ.get('CLR_VARIANTS', '').split(',')))
        # Launch chains on slices of the frame: with k > 1 every launch of
        # render_resident goes out as k launches, one per contiguous slice of the ray order, each
        # slice on a stream of its own.  A slice's launches stay in order, those of different
        # slices overlap, so the blocks of one fill the GPU while the last blocks of another
        # drain.  No pixel changes (a pixel belongs to one slice).  1 = everything on the caller's
        # stream, as the reference's single queue.
        self._overlap_parts = overlap_parts
        self._part_streams = {}          # (step slot, slice) -> stream
        # Pass-split programs only: up to d launches of render_resident log their addends at the
        # same time (d logs, d streams per slice); the sums are still added in launch order on the
        # slice's stream, so no pixel changes.  Small frames are bound by their longest path:
        # launch k + 1 fills the lanes that the last paths of launch k leave idle.
        # 'auto' for both: what measured best - two slices for a one-lane program,
        # one slice and two steps in flight for a pass-split one.
        self._steps_in_flight = steps_in_flight
        self._launch_counter = 0
        # next-event estimation with k points of the segment's light instead of one (no
        # counterpart in the reference, k = 1 is its estimator).  The light is still picked once per
        # segment; its contribution is the mean over the k points, each with its own visibility test.
        if not 1 <= int(light_samples) <= 16:
            raise ValueError("light_samples must be in 1..16")
        self.light_samples = int(light_samples)
        self.bounce_record = BOUNCE_RECORD + 2 * (self.light_samples - 1)
        self.light_weight_literal = float(np.float32(1.0 / self.light_samples)).hex() + 'f'
        # tile partitioning: (first ray, count) in ray order; a contiguous range of ray
        # indices is a set of 4-pixel-wide column blocks of the frame (get_image_order)
        self.ray_range = ray_range
        # interleaved tile partitioning: (g, N, band) = every N-th band of `band` consecutive rays,
        # starting with band g (multi_gpu.py; evens out tiles that differ in cost)
        self.ray_bands = ray_bands
        if ray_range is not None and ray_bands is not None:
            raise ValueError("ray_range and ray_bands exclude each other")
        self.shader = scene.shader(scene)
        self.rng = np.random.RandomState(seed)
        # 'tape': the reference's random numbers, drawn on the host in its order and shared by
        # all pixels of a pass (parity mode).  'philox': every pixel draws its own numbers on the
        # device (hip_templates/clr_rng.inl); camera jitter and wavelength stay per pass.
        if rng_mode not in ('tape', 'philox'):
            raise ValueError("rng_mode must be 'tape' or 'philox'")
        self.rng_mode = rng_mode
        # passes striped over several GPUs (multi_gpu.py): local pass j has the global index
        # first_pass + j * pass_stride in the per-pixel random number counter
        self.pass_stride = 1
        # unseeded (like the reference's global numpy generator): a fresh key per process, so that
        # `clray.py --append` runs add independent samples
        key = int.from_bytes(os.urandom(8), 'little') if seed is None else int(seed)
        self.rng_seed = key * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019 & (2**64 - 1)

        self._group_objects()
        # Long shape tests (ray marchers, mesh traversals: Tracer.outline_candidate) as functions of
        # their own instead of inlined into both object sweeps.  outline: False / True (every
        # candidate) / 'marchers' / 'meshes'; None = the default policy.
        if outline is None:
            outline = self.default_outline(scene)
        self.outline = outline
        for obj in scene.objects:
            tracer = obj.tracer
            is_mesh = tracer.has_data()
            tracer.outlined = bool(tracer.outline_candidate and (
                outline is True or (outline == 'marchers' and not is_mesh) or (outline == 'meshes' and is_mesh)))
        # development switch of the mesh traversal (objects/meshes.py): a module global of counters
        self.profile_blocks = os.environ.get('CLR_BLOCK_PROFILE') == '1'     # development, tools/block_costs.py
        self.profile_mesh_phases = any(getattr(obj.tracer, 'profile_phases', False) for obj in scene.objects)
        self._init_lights()
        self._collect_tracer_data()
        self._init_misc()
        self._init_camera_and_image()
        self._init_materials()

        # exact division / square root by operand-range speculation with path restart
        # (hip_templates/clr_exact.inl); False = the compiler's generic expansions everywhere
        # default: on when every shape of the scene is a quadric / convex body - their steps are
        # dominated by the divisions and square roots of the ray set-up.  Marching and mesh
        # shapes gain nothing and their kernels are register-bound: the second copy of the
        # segment code costs them a wave of occupancy (scene-test: 239 -> 262 VGPRs).
        if speculate is None:
            speculate = all(getattr(obj.tracer, 'speculation_friendly', False) for obj in scene.objects)
        self.speculate = speculate
        # Pass-split mode for small frames.  One lane per pixel gives a 320 x 240 frame 1200 waves
        # for 1024 SIMDs that each want 4-8; the program then deals the passes of a pixel out to
        # several lanes, which log their addends, and a second kernel sums the logs in pass order
        # (hip_templates/main.hip; same additions in the same order, same bits).  Measured:
        # scene-test 320 x 240 32 -> 275 Msamples/s, scene-dev 320 x 240 4.1 -> 11.8 Gsamples/s,
        # 800 x 600 (the reference's default size) 16.8 -> 19.2; at 1080p the log costs a light
        # kernel 11 %, so the default is: on up to 1.5 x the lanes that fill the GPU.
        # Scenes with triangle meshes are split at every size, into at least 4 groups: their waves
        # are long and uneven (a ray that enters the octree costs many times one that misses it), and
        # shorter waves even that out - scene-mesh 1080p 338 -> 385 Msamples/s, scene-test 1080p
        # 0.68 -> 1.04 Gsamples/s.  (Ray-marching scenes lose: scene-marchers 4K 1.48 -> 1.29.)
        self.has_mesh = any(obj.tracer.has_data() for obj in scene.objects)
        if split_passes is None:
            split_passes = 2 * self.n_pixels <= 3 * FULL_GPU_LANES or self.has_mesh
        self.split_passes = bool(split_passes)
        # lanes per pixel a pass-split launch uses at least (more on small frames, pass_groups_for)
        self.min_pass_groups = 4 if self.has_mesh else 1
        self._pass_logs = {}
        # optional occupancy hint for the render kernel (second __launch_bounds__ argument)
        self.min_waves_per_simd = min_waves_per_simd
        # small scenes: bake the per-object parameters into the program text as literals
        self.bake_parameters = len(scene.objects) <= 48 if bake_parameters is None else bake_parameters
        # path regeneration (main.hip): identical results; no gain while all pixels share their
        # random numbers (lock-step then keeps pass and segment in scalar registers and reads the
        # tape through scalar loads), +26..44 % once every pixel draws its own
        self._regenerate = (self.rng_mode == 'philox') if regenerate is None else bool(regenerate)
        # Sorted form of the regenerating kernel (main.hip): the paths of a block change lanes before
        # every segment - primary rays together, the others by the octant of their direction - so
        # that the lanes of a wave do like work.  For per-pixel random numbers, where the paths of
        # neighbouring pixels go their own ways (with the broadcast tape they stay a bundle).
        if sorted_paths is None:
            sorted_paths = self.default_sorted_paths()
        self.sorted_paths = bool(sorted_paths)
        if self.sorted_paths and (self.split_passes or not self._regenerate or len(scene.objects) >= 1 << 16
                                  or scene.max_bounces > 15):
            raise ValueError("sorted_paths needs the regenerating one-lane-per-pixel program, < 65536 objects "
                             "and at most 15 segments per path")
        self.acc = None
        self.program_code = Compiler(self).make_program()
        self.build_options = list(build_options if build_options is not None else PARITY_OPTIONS)
        extra = getattr(args, 'cl_build_options', None)
        if extra:
            self.build_options += extra.split()
        if device:
            self.connect(device_index)

    @staticmethod
    def default_outline(scene):
        return False

    def default_sorted_paths(self):
        return False

    @property
    def regenerate(self):
        return self._regenerate

    @regenerate.setter
    def regenerate(self, value):
        """The two forms of clr_render are two programs: switching regenerates and rebuilds."""
        if bool(value) == self._regenerate:
            return
        if not value:
            self.sorted_paths = False
        self._regenerate = bool(value)
        self.program_code = Compiler(self).make_program()
        if self.acc is not None:
            self._build()

    # ------------------------------------------------------------------ scene preparation
    def _group_objects(self):
        """Sort objects by step-function name (stable) and number them from 1
        (reference renderer.py:51-74)."""
        key = lambda obj: obj.tracer.tracer_kernel_name
        self.scene.objects.sort(key=key)
        self.object_groups = []
        offset = 0
        for _, members in itertools.groupby(self.scene.objects, key=key):
            members = list(members)
            self.object_groups.append((members[0].tracer, len(members), offset))
            offset += len(members)
        self.root_object_id = 0
        for i, obj in enumerate(self.scene.objects):
            if self.scene.root_object is obj:
                self.root_object_id = i + 1

    def _init_lights(self):
        self.bidirectional_light_ids = [i for i, obj in enumerate(self.scene.objects)
                                        if obj.bidirectional_light]
        self.bidirectional = len(self.bidirectional_light_ids) > 0
        self.n_lights = len(self.bidirectional_light_ids)
        # RGB material tables (5 float4 + 5 float per object and the ambient medium) are copied to
        # LDS by every block when they are small: 10 kB at most
        self.materials_in_lds = bool(self.shader.rgb) and 5 * (len(self.scene.objects) + 1) <= 512
        # per-pixel random numbers: the light points are drawn on the device, from this table: per
        # light 8 floats (centre, R or 0 for a box, area, minimum sampling distance, object id, float
        # offset of the box's faces), then per box 6 face records of 16 floats (clr_rng.inl)
        headers = np.zeros((self.n_lights, 8), dtype=np.float32)
        faces = []
        if self.rng_mode == 'philox':
            for row, light_id in zip(headers, self.bidirectional_light_ids):
                tracer = self.scene.objects[light_id].tracer
                center, min_dist = tracer.center_and_min_sampling_distance()
                row[0:3], row[4], row[5] = center, tracer.surface_area(), min_dist
                row[6:7] = np.array([light_id + 1], dtype=np.int32).view(np.float32)
                if hasattr(tracer, 'R') and hasattr(tracer, '_check_no_distortion'):
                    tracer._check_no_distortion()
                    row[3] = tracer.R
                elif hasattr(tracer, '_faces'):
                    offset = headers.size + 16 * len(faces)
                    row[7:8] = np.array([offset], dtype=np.int32).view(np.float32)
                    box = tracer._faces()
                    shares = np.cumsum([f[4] for f in box]) / sum(f[4] for f in box)
                    for (start, u, v, outward, _), share in zip(box, shares):
                        faces.append(np.concatenate([start, [share], u, [0], v, [0], outward, [0]]))
                else:
                    raise NotImplementedError("rng_mode='philox' samples sphere and box lights only")
        self.has_box_lights = len(faces) > 0
        self.light_table = np.concatenate([headers.ravel()] + [np.asarray(f, dtype=np.float32) for f in faces])

    def _collect_tracer_data(self):
        """Pack per-object parameters and mesh data (role of reference renderer.py:121-241).

        Layout: `param_table` holds one float4 slot per declared parameter of an object
        (float3 in xyz, float or bit-cast int in x) followed by four slots with the rows
        and the translation of the INVERSE of global_to_tracer_coordinate_transform().
        Objects of one group are contiguous with a common stride, so a step function finds
        its slots at group_base + i * group_stride.  Mesh data go to `vector_data` (float4)
        and `integer_data` (int32) with per-object offsets in `data_offsets`.
        """
        slots, vectors, integers, data_offsets = [], [], [], []
        n_vec = n_int = 0
        self.object_param_base = []
        for obj in self.scene.objects:
            tracer = obj.tracer
            data_offsets += [n_vec, n_int]
            if tracer.has_data():
                data = tracer.get_data()
                vec = np.asarray(data['vector'], dtype=np.float32)
                if vec.ndim != 2 or vec.shape[1] not in (3, 4):
                    raise RuntimeError('invalid vector data shape')
                vectors.append(vec if vec.shape[1] == 4 else Accelerator.host_vec3_array(vec))
                n_vec += vec.shape[0]
                ints = np.ravel(data['integer']).astype(np.int32)
                integers.append(ints)
                n_int += ints.size
            self.object_param_base.append(len(slots))
            slots.extend(tracer.parameter_slots())
            tracer.param_slots = len(slots) - self.object_param_base[-1]

        self.param_table = np.array(slots, dtype=np.float32).reshape(-1, 4)
        self.n_param_slots = self.param_table.shape[0]
        self.data_offsets = np.array(data_offsets, dtype=np.int32)
        self.vector_data = (np.vstack(vectors) if vectors
                            else np.zeros((1, 4), dtype=np.float32))
        self.integer_data = (np.concatenate(integers) if integers else np.zeros(1, dtype=np.int32))

        self.group_param_base, self.group_param_stride = [], []
        for tracer, count, offset in self.object_groups:
            strides = {self.scene.objects[offset + i].tracer.param_slots for i in range(count)}
            assert len(strides) == 1, "objects of one group must share a parameter layout"
            self.group_param_base.append(self.object_param_base[offset])
            self.group_param_stride.append(strides.pop())

    def affine_skip_mask(self, object_index):
        """Which terms of an object's inverse-affine transform the speculative arithmetic may
        leave out (clr_exact.inl): bit 3*row+col = the coefficient is zero, bit 9+row = the
        translation of that row is non-zero (only then may the position row be shortened).
        Rows of zeros are left alone.  Zero unless the parameters are baked into the program."""
        if not (self.bake_parameters and self.speculate):
            return 0
        end = self.object_param_base[object_index] + self.scene.objects[object_index].tracer.param_slots
        rows = self.param_table[end - 4:end, :3]
        mask = 0
        for r in range(3):
            zeros = rows[r] == 0.0
            if zeros.all():
                continue
            for c in range(3):
                if zeros[c]:
                    mask |= 1 << (3 * r + c)
            if rows[3][r] != 0.0:
                mask |= 1 << (9 + r)
        if self.shares_world_direction and self._identity_linear(object_index):
            mask |= 1 << 12
        return mask

    def _identity_linear(self, object_index):
        end = self.object_param_base[object_index] + self.scene.objects[object_index].tracer.param_slots
        return bool(np.array_equal(self.param_table[end - 4:end - 1, :3], np.eye(3, dtype=np.float32)))

    @property
    def shares_world_direction(self):
        """Objects whose inverse transform has an identity linear part see the world ray itself as
        their local ray: its normalisation (the reference's per-object first step) is the same
        computation for all of them and is done once per sweep when at least two such objects
        exist (bit 12 of the affine skip mask; speculative arithmetic with baked parameters only,
        where the products with the literal 1 and 0 coefficients fold away)."""
        if not (self.bake_parameters and self.speculate):
            return False
        return sum(self._identity_linear(i) for i in range(len(self.scene.objects))) >= 2

    def param_table_literals(self):
        """The parameter table as exact C++17 hexadecimal float literals, row by row."""
        return [[float(v).hex() + 'f' for v in row] for row in self.param_table]

    def _init_misc(self):
        # consumes the RNG exactly like the reference (renderer.py:84-85)
        self.qdirs = camera.quasi_random_direction_sample(self.scene.samples_per_pixel)
        self.qdirs = self.rng.permutation(self.qdirs)

    def _init_camera_and_image(self):
        scene = self.scene
        cam = scene.get_camera_rays()
        self.rotmat = scene.get_camera_rotmat()
        fovx_rad = scene.camera_fov / 180.0 * np.pi
        self.pixel_angle = fovx_rad / scene.image.size[0]
        self.img_shape = tuple(scene.image.size[::-1])          # (H, W)
        self.n_pixels = self.img_shape[0] * self.img_shape[1]
        self.image_order = self.full_image_order = self.get_image_order()
        self.ray_index = None                   # global index of every local ray (None: the whole frame)
        if self.ray_range is not None:
            first, count = self.ray_range
            assert 0 <= first and first + count <= self.n_pixels and count > 0
            self.ray_index = np.arange(first, first + count)
        elif self.ray_bands is not None:
            g, n, band = self.ray_bands
            assert 0 <= g < n and band > 0 and g * band < self.n_pixels
            starts = np.arange(g * band, self.n_pixels, n * band)
            self.ray_index = np.concatenate([np.arange(b, min(b + band, self.n_pixels)) for b in starts])
        if self.ray_index is not None:
            self.image_order = self.image_order[self.ray_index]
            self.n_pixels = len(self.ray_index)
        self.host_cam = Accelerator.host_vec3_array(
            cam[self.image_order[:, 0], self.image_order[:, 1], 0:3])

    def _init_materials(self):
        if self.shader.rgb:
            self.material_colors, self.material_scalars = self.shader.host_material_tables()
        else:
            self.shader.prepare_host_tables()

    def get_image_order(self):
        """Ray index -> (row, col): 8-row x 4-column blocks, column blocks outermost
        (reference renderer.py:435-456; its `width` is the image height)."""
        rows, cols = self.img_shape
        block_r, block_c = 8, 4
        if rows % block_r == 0 and cols % block_c == 0:
            k = np.arange(rows * cols)
            in_r = k % block_r
            in_c = (k // block_r) % block_c
            blk_r = (k // (block_r * block_c)) % (rows // block_r)
            blk_c = k // (block_r * block_c * (rows // block_r))
            return np.column_stack((in_r + blk_r * block_r, in_c + blk_c * block_c))
        print("WARNING: image size should be (%d*x, %d*y)" % (block_c, block_r))
        r, c = np.meshgrid(np.arange(rows), np.arange(cols), indexing='ij')
        return np.column_stack((r.ravel(), c.ravel()))

    # ------------------------------------------------------------------ random tape
    def new_camera_sample(self):
        """Sub-pixel jitter and depth-of-field offset of one pass (reference renderer.py:251-291)."""
        scene = self.scene
        sx = np.float32(self.rng.rand())
        sy = np.float32(self.rng.rand())
        if scene.tent_filter:
            def tent(x):
                x *= 2
                return np.sqrt(x) - 1 if x < 1 else 1 - np.sqrt(2 - x)
            sx, sy = tent(sx), tent(sy)
        overlap = 0.0
        thetax = (sx - 0.5) * self.pixel_angle * (1.0 + overlap)
        thetay = (sy - 0.5) * self.pixel_angle * (1.0 + overlap)
        dofx, dofy = camera.random_dof_sample(self.rng)
        dof_pos = (dofx * self.rotmat[:, 0] + dofy * self.rotmat[:, 1]) * scene.camera_dof_fstop
        tilt = camera.rotmat_tilt_camera(thetax, thetay)
        mat4 = np.zeros((4, 4))
        mat4[0:3, 0:3] = np.dot(np.dot(self.rotmat, tilt), self.rotmat.transpose())
        mat4[3, 0:3] = dof_pos
        mat4[3, 3] = scene.camera_sharp_distance
        return np.asarray(scene.camera_position) + dof_pos, mat4

    def get_light_point(self):
        ids = self.bidirectional_light_ids
        light_id = ids[self.rng.randint(len(ids))]
        tracer = self.scene.objects[light_id].tracer
        return ((light_id, tracer.surface_area()) + tracer.random_surface_point_and_normal(self.rng)
                + tracer.center_and_min_sampling_distance())

    def make_tape(self, first_pass, n_passes, segments=None):
        """Draw the random numbers of `n_passes` passes in the reference's order (SURVEY
        Appendix B) and lay them out for the device.

        The reference stops drawing for a pass once no ray of the frame is alive
        (renderer.py:318-320, 430), which only happens at very small frame sizes; this
        generator always draws all max_bounces segments unless `segments[j]` gives the number
        of segments the reference executed in pass j (used to replay recorded tapes)."""
        scene = self.scene
        max_bounces = scene.max_bounces
        passes = np.zeros((n_passes, PASS_RECORD, 4), dtype=np.float32)
        per_pixel = self.rng_mode == 'philox'        # then only the per-pass part is drawn here
        bounces = None if per_pixel else np.zeros((n_passes, max_bounces, self.bounce_record, 4), dtype=np.float32)
        rows = None
        if not self.shader.rgb:
            rows = np.zeros((n_passes, self.shader.host_mat_y.shape[0]), dtype=np.float32)
        for j in range(n_passes):
            cam_origin, mat4 = self.new_camera_sample()
            passes[j, 0:4, :] = mat4
            passes[j, 4, 0:3] = cam_origin
            passes[j, 4, 3] = 1.0
            passes[j, 5, 0:3] = 1.0
            if rows is not None:
                inv_pdf, rows[j], cmask = self.shader.sample_wavelength(self.rng.random_sample())
                passes[j, 4, 3] = inv_pdf
                passes[j, 5, 0:3] = cmask
            for b in range(0 if per_pixel else (max_bounces if segments is None else int(segments[j]))):
                is_last = b == max_bounces - 1
                rec = bounces[j, b]
                if self.bidirectional:
                    light_id0, area, point, normal, center, min_dist = self.get_light_point()
                    rec[2, 0:3], rec[2, 3] = point, area
                    rec[3, 0:3], rec[3, 3] = normal, min_dist
                    rec[4, 0:3] = center
                    rec[4, 3:4] = np.array([0 if is_last else light_id0 + 1], dtype=np.int32).view(np.float32)
                    tracer = scene.objects[light_id0].tracer
                    for extra in range(self.light_samples - 1):
                        point, normal = tracer.random_surface_point_and_normal(self.rng)
                        rec[BOUNCE_RECORD + 2 * extra, 0:3] = point
                        rec[BOUNCE_RECORD + 2 * extra + 1, 0:3] = normal
                if scene.quasirandom and b == 1:
                    rec[0, 0:3] = self.qdirs[first_pass + j, :]
                else:
                    rec[0, 0:3] = utils.normalize(self.rng.normal(0, 1, (3,)))
                rec[0, 3] = np.float32(self.rng.rand())
                rec[1, 0:3] = self.rng.normal(0, 1, (3,))
                if not is_last:
                    rec[1, 3] = np.float32(self.rng.rand())
        return passes, bounces, rows

    # ------------------------------------------------------------------ device
    def connect(self, device_index=0):
        """Create the device context, upload the scene and build the program."""
        acc = self.acc = Accelerator(device_index)
        self.cam = acc.to_device(self.host_cam)
        self.img = acc.new_vec3_array((self.n_pixels,))
        self.dev_vector_data = acc.to_device(self.vector_data)
        self.dev_integer_data = acc.to_device(self.integer_data)
        self._build()
        self._tape_capacity = 0
        self.dev_rows = None

    def _build(self, keep_previous=False):
        acc = self.acc
        self.prog = acc.build_program(self.program_code, self.build_options, keep_previous=keep_previous)
        if self.n_param_slots and not self.bake_parameters:
            acc.set_global('clr_params', self.param_table)
        acc.set_global('clr_data_offsets', self.data_offsets)
        if self.rng_mode == 'philox' and self.n_lights:
            acc.set_global('clr_lights', self.light_table)
        if self.shader.rgb:
            acc.set_global('clr_material_colors', self.material_colors)
            acc.set_global('clr_material_scalars', self.material_scalars)

    def tune_occupancy(self, candidates=(None, 4, 5, 6, 7, 8), passes=16, rounds=None, verbose=False):
        """Pick the occupancy hint of the render kernel (second __launch_bounds__ argument) by
        timing: the register allocation of these scene-specialised kernels lands on either side
        of a waves-per-SIMD boundary more or less by chance, and one more resident wave or a few
        spills move the run time by 5-20 % (profiles/r01c_occupancy.log).  Results do not depend
        on the hint.  Builds the program once per candidate and leaves the fastest one in place;
        the image accumulator and the random stream of the renderer are not touched."""
        saved_rng, saved_img, saved_tape = self.rng, self.img, (
            getattr(self, 'dev_pass_tape', None), getattr(self, 'dev_bounce_tape', None),
            self.dev_rows, self._tape_capacity, getattr(self, '_resident_passes', 0))
        self.rng = np.random.RandomState(20240229)
        tape = self.make_tape(0, passes)
        self.rng = saved_rng
        self.img = self.acc.new_vec3_array((self.n_pixels,))
        saved_first = getattr(self, '_resident_first_pass', 0)
        self.upload_tape(tape)
        # all candidates stay loaded and are timed in interleaved rounds, so that clock and
        # thermal drift hits them alike (differences of 3-5 % decide here)
        previous = self.acc.prog
        built, meta = {}, {}
        for hint in candidates:
            self.min_waves_per_simd = hint
            code = Compiler(self).make_program()
            self.program_code = code
            self._build(keep_previous=True)
            built[hint] = (self.acc.prog, code)
            meta[hint] = self.acc.last_metadata.get('clr_render', {})
            warm_ms = self.render_tape(tape)[0]                          # warm-up
        self.acc.drop_program(previous)
        samples = {hint: [] for hint in candidates}
        # launches that take a quarter of a second or more: one step per sample, three rounds, and
        # the launch pattern stays as it is (the tuning should not take minutes)
        heavy = warm_ms >= 250.0
        steps = 1 if heavy else TUNE_STEPS
        if rounds is None:
            # candidates often lie within 2-3 % of each other, the spread of single launches:
            # about 0.3 s of launches per candidate, 7 to 31 rounds (medians decide)
            rounds = 3 if heavy else int(min(31, max(7, 300.0 / max(steps * warm_ms, 1e-3))))

        def sample():
            # `steps` steps back to back in the launch pattern that will be used (launch chains
            # on slices, steps in flight): what one more wave or a few spills are worth shows
            # there, where the launches overlap, more than in a launch that runs alone
            self.acc.timer_begin()
            for _ in range(steps):
                self.render_resident(0, passes, timed=False)
            self.join_parts()
            return self.acc.timer_end() / steps
        for _ in range(rounds):
            for hint in candidates:
                self.acc.use_program(built[hint][0])
                samples[hint].append(sample())
        timings = {hint: float(np.median(v)) for hint, v in samples.items()}
        # vector registers each build spills to scratch memory (code object metadata): a build that
        # spills more than the leanest one moves registers through HBM inside the segment loop (a
        # round-2 build of scene-dev with dozens of them: 7 x the algorithmic HBM writes), which
        # hurts any neighbour on the GPU; such a build has to win by a margin that grows with the
        # number of registers it spills (pick_occupancy_hint)
        scratch = {hint: int(meta[hint].get('.vgpr_spill_count', 0)) for hint in candidates}
        if verbose:
            print('occupancy hints, median ms:', timings, 'spilled vector registers:', scratch)
        best = pick_occupancy_hint(timings, scratch)
        self.min_waves_per_simd = best
        self.prog, self.program_code = built[best]
        self.acc.use_program(self.prog)
        for hint, (prog, _) in built.items():
            if hint != best:
                self.acc.drop_program(prog)
        # one-lane programs with the automatic launch pattern: two launch chains or three, by the
        # same timing (scene-dev at eight waves: three are 1 % faster; a third chain has to win by 0.5 %)
        if self._overlap_parts == 'auto' and not self.split_passes and self.n_pixels >= 3 * 4096 and rounds > 1 and not heavy:
            parts = {}
            for k in (2, 3):
                self._auto_parts = k
                sample()
            for _ in range(rounds):
                for k in (2, 3):
                    self._auto_parts = k
                    parts.setdefault(k, []).append(sample())
            medians = {k: float(np.median(v)) for k, v in parts.items()}
            self._auto_parts = 3 if medians[3] < 0.995 * medians[2] else 2
            self.chain_timings = medians
            if verbose:
                print('launch chains, median ms per step:', medians)
        # pass-split programs with the automatic launch pattern: two steps in flight or three (each
        # with an addend log of its own), timed over twice as many steps so that three can overlap;
        # the launch counter restarts with the pattern (a step's sums are added after those of the
        # step before it, found by its slot)
        if self._steps_in_flight == 'auto' and self.split_passes and rounds > 1 and not heavy:
            flights = {}

            def sample_flight(k):
                self.acc.finish()
                self._auto_in_flight, self._launch_counter = k, 0
                self.acc.timer_begin()
                for _ in range(2 * TUNE_STEPS):
                    self.render_resident(0, passes, timed=False)
                self.join_parts()
                return self.acc.timer_end() / (2 * TUNE_STEPS)
            for k in (2, 3):
                sample_flight(k)
            for _ in range(rounds):
                for k in (2, 3):
                    flights.setdefault(k, []).append(sample_flight(k))
            medians = {k: float(np.median(v)) for k, v in flights.items()}
            self.acc.finish()
            self._auto_in_flight, self._launch_counter = (3 if medians[3] < 0.98 * medians[2] else 2), 0
            self.flight_timings = medians
            if verbose:
                print('steps in flight, median ms per step:', medians)
        self.acc.finish()
        self.img = saved_img
        self.dev_pass_tape, self.dev_bounce_tape, self.dev_rows, self._tape_capacity, self._resident_passes = saved_tape
        self._resident_first_pass = saved_first
        self.occupancy_timings = timings
        self.occupancy_scratch = scratch
        self.kernel_metadata = meta[best]
        return best

    def _upload_tape(self, tape):
        passes, bounces, rows = tape
        acc = self.acc
        n = passes.shape[0]
        if n > self._tape_capacity or (bounces is not None and getattr(self, 'dev_bounce_tape', None) is None) \
                or (rows is not None and self.dev_rows is None):
            n = max(n, self._tape_capacity)
            self.dev_pass_tape = acc.new_array((n,) + passes.shape[1:], np.float32)
            self.dev_bounce_tape = acc.new_array((n,) + bounces.shape[1:], np.float32) if bounces is not None else None
            self.dev_rows = (acc.new_array((n, rows.shape[1]), np.float32) if rows is not None else None)
            self._tape_capacity = n
            n = passes.shape[0]
        self.dev_pass_tape[:n].set(passes)
        if bounces is not None:
            self.dev_bounce_tape[:n].set(bounces)
        if rows is not None:
            self.dev_rows[:n].set(rows)

    def use_image_buffer(self, device_ptr):
        """Accumulate into caller-owned device memory (n_pixels x float4), e.g. a torch
        tensor that takes part in an all-reduce."""
        from .accelerator import DeviceArray
        self.img = DeviceArray(self.acc, (self.n_pixels, 4), np.float32, ptr=int(device_ptr))

    def upload_tape(self, tape, first_pass=0):
        """Make `tape` resident on the device; later launches pick pass ranges out of it.
        first_pass: global index of the tape's first pass (per-pixel random numbers count passes
        globally)."""
        self._tape_capacity = 0
        self._upload_tape(tape)
        self._resident_passes = tape[0].shape[0]
        self._resident_first_pass = int(first_pass)

    def render_resident(self, first, count, stream=None, timed=True, static_log=False):
        """Render passes [first, first + count) of the resident tape (no host transfer).
        static_log: through clr_render_passes_static_log, the entry point that allocates nothing
        (pass-split programs; the caller keeps `count` within the library's static log)."""
        assert 0 <= first and first + count <= self._resident_passes
        if self.split_passes and not static_log:
            # keep the addend log below LOG_BYTES_LIMIT: longer ranges go as several launches
            most = self.max_passes_per_launch()
            if count > most:
                ms = [self.render_resident(f, min(most, first + count - f), stream, timed)
                      for f in range(first, first + count, most)]
                return sum(ms) if timed else None
        rows = self.dev_rows[first:first + count].ptr if self.dev_rows is not None else None

        def launch(lo, hi, on_stream, timed_launch):
            rng = self._rng_args(self._resident_first_pass + first * self.pass_stride)
            rng['first_ray'] += lo
            args = RenderArgs(
                img=self.img.ptr + 16 * lo, cam=self.cam.ptr + 16 * lo,
                pass_tape=self.dev_pass_tape[first:first + count].ptr,
                bounce_tape=self.dev_bounce_tape[first:first + count].ptr if self.dev_bounce_tape is not None else None,
                material_rows=rows,
                vector_data=self.dev_vector_data.ptr, integer_data=self.dev_integer_data.ptr,
                n_pixels=hi - lo, n_passes=count, regenerate=int(self.regenerate),
                debug_dump=None, **split, **rng)
            if split and lo and not static_log:
                # a slice's part of the addend log: the log is (pass, float, pixel) with the slice's
                # own pixel count as the stride, so the slices' logs lie one after the other
                args.pass_log += 4 * count * (1 + 6 * self.scene.max_bounces) * lo
            if after is not None:
                args.reduce_after = after
            return self.acc.render_passes(args, stream=on_stream, timed=timed_launch, static_log=static_log)

        after = None
        chained = stream is None and not static_log
        in_flight = self.steps_in_flight if chained and self.split_passes else 1
        if chained:
            slot = self._launch_counter % in_flight
            self._launch_counter += 1
        else:
            # a launch on a stream of the caller's is not ordered against the slices' streams: it
            # gets an addend log of its own instead of the one a chained launch may still be using
            slot = 'caller'
        split = self._split_args(count, static_log=static_log, slot=slot)
        parts = self.part_bounds() if chained else [0, self.n_pixels]
        if len(parts) == 2 and in_flight == 1:
            return launch(0, self.n_pixels, stream, timed)
        # pass-split program with several steps in flight: launch k + 1 logs into another buffer on
        # another stream while launch k is still running; its sums are added after launch k's
        # (clr_render_args.reduce_after).  One stream per (slot, slice).
        n_slices = len(parts) - 1

        def stream_of(step_slot, p):
            if (step_slot, p) not in self._part_streams:
                self._part_streams[(step_slot, p)] = self.acc.new_stream()
            return self._part_streams[(step_slot, p)]
        if timed:
            self.acc.timer_begin()
        for p in range(n_slices):
            if in_flight > 1:
                after = stream_of((slot - 1) % in_flight, p)
            launch(parts[p], parts[p + 1], stream_of(slot, p), False)
        if timed:
            self.join_parts()
            ms = self.acc.timer_end()
            self.acc._record('clr_render', ms * 1e-3, ms * 1e-3)
            return ms
        return None

    @property
    def overlap_parts(self):
        if self._overlap_parts == 'auto':
            # (tune_occupancy may have timed two chains against three: _auto_parts)
            return 1 if self.split_passes else getattr(self, '_auto_parts', 2)
        return max(1, int(self._overlap_parts))

    @overlap_parts.setter
    def overlap_parts(self, value):
        self._overlap_parts = value

    @property
    def steps_in_flight(self):
        if self._steps_in_flight == 'auto':
            # (tune_occupancy may have timed two steps in flight against three: _auto_in_flight)
            return getattr(self, '_auto_in_flight', 2) if self.split_passes else 1
        return max(1, int(self._steps_in_flight))

    @steps_in_flight.setter
    def steps_in_flight(self, value):
        self._steps_in_flight = value

    def part_bounds(self):
        """Ray indices at which the frame is cut into overlap_parts slices (multiples of the block size)."""
        k = min(self.overlap_parts, max(1, self.n_pixels // 4096))
        cuts = [((self.n_pixels * p // k + 255) // 256) * 256 for p in range(k)] + [self.n_pixels]
        cuts[0] = 0
        return cuts

    def join_parts(self, stream=None):
        """`stream` (None: the default stream) waits for the launches on the slices' streams.  Work
        on the default stream does so by itself (ordinary HIP streams); this is for other streams."""
        for part_stream in self._part_streams.values():
            self.acc.stream_wait(stream, part_stream)

    def max_passes_per_launch(self):
        """Passes one launch of a pass-split program may cover so that its addend log stays below
        LOG_BYTES_LIMIT (None for a one-lane-per-pixel program: no log, no limit)."""
        if not self.split_passes:
            return None
        return log_passes_limit(self.n_pixels, self.scene.max_bounces)

    def _split_args(self, n_passes, debug=False, static_log=False, slot=0):
        """The pass-split fields of clr_render_args (empty for a one-lane-per-pixel program).
        slot: which of the steps_in_flight addend logs to use."""
        if not self.split_passes:
            return {}
        groups = 1 if debug else pass_groups_for(self.n_pixels, n_passes, self.min_pass_groups)
        if static_log:
            return dict(pass_log=None, pass_groups=groups)
        floats = 1 + 6 * self.scene.max_bounces
        log = self._pass_logs.get(slot)
        if log is None or log.shape[0] < n_passes:
            if log is not None:
                self.acc.finish()          # a launch in flight may still use it
            self._pass_logs[slot] = log = None          # release before growing
            self._pass_logs[slot] = log = self.acc.new_array((max(n_passes, 1), floats, self.n_pixels), np.float32)
        return dict(pass_log=log.ptr, pass_groups=groups)

    def _rng_args(self, first_pass):
        """The per-pixel random number fields of clr_render_args."""
        bands = {}
        if self.ray_bands is not None:
            g, n, band = self.ray_bands
            bands = dict(ray_band=int(band), ray_band_stride=int(n * band), ray_band_base=int(g * band))
        return dict(per_pixel_rng=int(self.rng_mode == 'philox'),
                    rng_seed_lo=self.rng_seed & 0xffffffff, rng_seed_hi=(self.rng_seed >> 32) & 0xffffffff,
                    first_pass=int(first_pass), first_ray=self.ray_range[0] if self.ray_range else 0,
                    pass_stride=int(self.pass_stride), **bands)

    def render_tape(self, tape, debug=False, stream=None, timed=True, first_pass=0, static_log=False):
        """Render the passes described by `tape` and add them to the image accumulator.
        Returns (kernel milliseconds, debug dump or None)."""
        n = tape[0].shape[0]
        most = self.max_passes_per_launch()
        if most is not None and n > most and not debug and not static_log:
            # the CLI's batches of 64 passes would need a 13 GB log at 1080p: several launches
            total = 0.0
            for f in range(0, n, most):
                part = tuple(x[f:f + most] if x is not None else None for x in tape)
                ms, _ = self.render_tape(part, stream=stream, timed=timed,
                                         first_pass=first_pass + f * self.pass_stride)
                total += ms
            return total, None
        self._upload_tape(tape)
        dump = None
        if debug:
            dump = self.acc.new_array((self.scene.max_bounces, self.n_pixels, DEBUG_RECORD, 4),
                                      np.float32, zeros=True)
        args = RenderArgs(
            img=self.img.ptr, cam=self.cam.ptr, pass_tape=self.dev_pass_tape.ptr,
            bounce_tape=self.dev_bounce_tape.ptr if self.dev_bounce_tape is not None else None,
            material_rows=self.dev_rows.ptr if self.dev_rows is not None else None,
            vector_data=self.dev_vector_data.ptr, integer_data=self.dev_integer_data.ptr,
            n_pixels=self.n_pixels, n_passes=n, regenerate=int(self.regenerate),
            debug_dump=dump.ptr if dump is not None else None, **self._split_args(n, debug, static_log),
            **self._rng_args(first_pass))
        ms = self.acc.render_passes(args, stream=stream, timed=timed, static_log=static_log)
        return ms, (dump.get() if dump is not None else None)

    def render_samples(self, first_pass, n_passes):
        """Render passes [first_pass, first_pass + n_passes) with one kernel launch."""
        ms, _ = self.render_tape(self.make_tape(first_pass, n_passes), first_pass=first_pass)
        return ms

    def render_sample(self, sample_index):
        """One sample pass (reference renderer.py:293-324); returns the last path index."""
        self.render_samples(sample_index, 1)
        return self.scene.max_bounces - 1

    def first_hit(self, tape=None):
        """Object / sub-object ids, hit distance and normal of the primary rays, image-shaped."""
        if tape is None:
            tape = self.make_tape(0, 1)
        self._upload_tape(tape)
        acc = self.acc
        ids = acc.new_array((self.n_pixels, 2), np.int32, zeros=True)
        t = acc.new_array((self.n_pixels,), np.float32, zeros=True)
        normal = acc.new_array((self.n_pixels, 4), np.float32, zeros=True)
        import ctypes
        from .accelerator import _check
        _check(acc.lib.clr_first_hit(acc.prog, self.cam.ptr, self.dev_pass_tape.ptr,
                                     self.dev_vector_data.ptr, self.dev_integer_data.ptr,
                                     self.n_pixels, ids.ptr, t.ptr, normal.ptr, None))
        acc.finish()
        return ids.get(), t.get(), normal.get()[:, :3]

    # ------------------------------------------------------------------ results
    def unpermute(self, per_ray):
        """(n_pixels, C) in ray order -> (H, W, C) image."""
        return unpermute_rays(per_ray, self.image_order, self.img_shape)

    def get_image(self):
        """Sum over all passes so far, (H, W, 3) (reference renderer.py:24-28)."""
        return self.unpermute(self.img.get().astype(np.float32)[:, 0:3]).astype(np.float64)

    def get_display_image(self, settings=None, base=None):
        """8-bit (H, W, 3) image of the sums so far, tone-mapped on the device (tonemap.py);
        `base` = earlier sums to add; flares included.  Raises NotImplementedError for settings
        the device path does not cover (reduction functions other than min / mean): use
        imgutils.Image then."""
        from .tonemap import DeviceToneMapper
        if getattr(self, '_tone_mapper', None) is None:
            self._tone_mapper = DeviceToneMapper(self)
            self._tone_base = None
        if base is not self._tone_base:
            self._tone_mapper.set_base(base)
            self._tone_base = base
        return self._tone_mapper.to_24bit(settings if settings is not None else self.scene.image)

    def rays_per_sample(self):
        return self.n_pixels

    @property
    def warp_size(self):
        return 64

    @property
    def log_warp_size(self):
        return int(round(math.log(self.warp_size, 2)))
