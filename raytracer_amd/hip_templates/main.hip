{#- Program skeleton: one scene-specialised HIP translation unit, compiled with hiprtc.
    Counterpart of the reference's main.cl (/root/reference/cl_templates/main.cl:2-30) plus
    the launch schedule of renderer.py:293-433, which here runs on the device: each lane
    owns one pixel and walks its sample passes and path segments in registers. -#}
{% include 'static.hip' %}
### if renderer.profile_mesh_phases
__device__ unsigned long long clr_mesh_phases[64 * 16];
### endif
### if renderer.profile_blocks
// development (tools/block_costs.py): cycles of the slowest wave of every block of clr_render
__device__ unsigned long long clr_block_cycles[65536];
### endif

// ---- scene tables, filled by the host after the module is loaded
// clr_params, per object: declared parameters, then 4 rows of the inverse affine
### if renderer.bake_parameters
// baked into the program: with the object loops unrolled every parameter becomes an immediate
__constant__ const float4 clr_params[{{ [renderer.n_param_slots, 1]|max }}] = {
### for row in renderer.param_table_literals()
    { {{ row|join(', ') }} },
### endfor
};
### else
__constant__ float4 clr_params[{{ [renderer.n_param_slots, 1]|max }}];
### endif
__constant__ int clr_data_offsets[{{ [2*n_objects, 2]|max }}];              // per object: offset into vector_data, integer_data
### if shader.rgb
__constant__ float4 clr_material_colors[{{ 5*(n_objects+1) }}];
__constant__ float clr_material_scalars[{{ 5*(n_objects+1) }}];
### if renderer.materials_in_lds
// block-local copy: a stage reads the properties of the surface it is on by object id, one
// dependent gather per stage; from LDS that is ~60 cycles each instead of a trip to L2
static __shared__ float4 clr_lds_material_colors[{{ 5*(n_objects+1) }}];
static __shared__ float clr_lds_material_scalars[{{ 5*(n_objects+1) }}];
static __shared__ float4 clr_lds_material_ratios[{{ 5*(n_objects+1) }}];      // colour / stage probability
### endif
### endif

// 1: every pixel draws its own random numbers (clr_rng.inl), 0: they come from the bounce tape;
// read by the host library to check the launch arguments against the program
__constant__ int clr_rng_mode = {{ 1 if renderer.rng_mode == 'philox' else 0 }};
// 1: pass-split program (clr_render writes a log of addends per pass, clr_reduce_log sums it);
// read by the host library like clr_rng_mode
__constant__ int clr_split_mode = {{ 1 if renderer.split_passes else 0 }};
// 1: regenerating form of clr_render, 0: lock-step form; read by the host library, which checks
// the `regenerate` launch argument against it
__constant__ int clr_regen_mode = {{ 1 if renderer.regenerate else 0 }};
#define CLR_LOG_FLOATS {{ 1 + 6*renderer.scene.max_bounces }}   // per (pass, pixel): header word, then <= 2 addends of 3 floats per segment
__constant__ int clr_log_floats = CLR_LOG_FLOATS;              // read by the host library (size check of the static log)
### if renderer.rng_mode == 'philox'
// next-event lights: per light (centre, R or 0 for a box), (area, minimum sampling distance, object id
// bits, float offset of the box's face records as int bits); then the face records (clr_rng.inl)
__constant__ float4 clr_lights[{{ [renderer.light_table.size // 4, 2]|max }}];
### endif

### if renderer.bidirectional
template<class CLR_G>
CLR_FN float clr_light_visibility(const vec3 pos, const vec3 normal, const uint whichobject,
    const uint which_subobject, const uint inside, const int diffusions_left,
    const vec3 dest, const int light_id,
    const float4 * __restrict__ vector_data, const int * __restrict__ integer_data, CLR_G &clr_g);
template<class CLR_G>
CLR_FN float clr_light_visibility_along(const vec3 pos, const uint whichobject,
    const uint which_subobject, const uint inside, const clr_direction &light_dir, const int light_id,
    const float4 * __restrict__ vector_data, const int * __restrict__ integer_data, CLR_G &clr_g);
### endif

### if renderer.light_samples > 1 and renderer.rng_mode == 'philox'
// point `sample` >= 1 on light `idx` of clr_lights from the pixel's own stream (clr_draw_bounce
// below draws point 0 the same way)
CLR_FN void clr_draw_light_point(const uint seed_lo, const uint seed_hi, const uint pixel, const uint pass,
                                 const uint segment, const int sample, const int idx, vec3 &point, vec3 &normal) {
    float light_dir[3], light_uniforms[3];
    clr_draw_light_extra(seed_lo, seed_hi, pixel, pass, segment, (uint)sample, light_dir, light_uniforms);
    const float4 L0 = clr_lights[2*idx], L1 = clr_lights[2*idx + 1];
    normal = make_vec3(light_dir[0], light_dir[1], light_dir[2]);
    point = make_vec3(L0.x + light_dir[0] * L0.w, L0.y + light_dir[1] * L0.w, L0.z + light_dir[2] * L0.w);
    ### if renderer.has_box_lights
    if (!(L0.w > 0.0f)) {
        float p[3], nrm[3];
        clr_box_light_point((const float *)clr_lights + __float_as_int(L1.w), light_uniforms[2],
                            light_uniforms[0], light_uniforms[1], p, nrm);
        point = make_vec3(p[0], p[1], p[2]);
        normal = make_vec3(nrm[0], nrm[1], nrm[2]);
    }
    ### endif
}
### endif

{% include 'shader.hip' %}

// ---- shape functions
### for k in functions.declarations
{{ k }}
### endfor

### for k in functions.definitions
{{ k }}
### endfor

// ---- per-class steps
### for k in functions.kernels
{{ k }}
### endfor

#define CLR_MAX_BOUNCES {{ renderer.scene.max_bounces }}
#define CLR_MAX_RAY_LENGTH {{ renderer.scene.max_ray_length|float }}f
#define CLR_N_OBJECTS {{ n_objects }}
#define CLR_PASS_STRIDE 6
#define CLR_BOUNCE_STRIDE {{ renderer.bounce_record }}
#define CLR_MATERIAL_ROW {{ 10*(n_objects+1) }}

// closest hit over every object of the scene (renderer.py:338-344)
template<class CLR_G>
CLR_FN void clr_closest_hit(const vec3 pos, const vec3 ray, float &isec_dist, uint &whichobject,
    uint &which_subobject, const uint last_whichobject, const uint last_which_subobject,
    const uint inside, const float4 * __restrict__ vector_data, const int * __restrict__ integer_data,
    CLR_G &clr_g)
{
    // the normalised world ray, shared by the objects whose local ray it is (clr_direction_of)
### if renderer.shares_world_direction
    const clr_direction world_dir = clr_make_direction(ray, clr_g);
### else
    const clr_direction world_dir = {};
### endif
### for tracer, count, offset in renderer.object_groups
### set group = loop.index0
### if renderer.bake_parameters
    ### for i in range(count)
    {{ tracer.tracer_kernel_name }}(pos, ray, isec_dist, whichobject, which_subobject,
        last_whichobject, last_which_subobject, inside, vector_data, integer_data,
        {{ offset+1+i }}, clr_params + {{ renderer.group_param_base[group] + i*renderer.group_param_stride[group] }},
        {{ renderer.affine_skip_mask(offset+i) }}, world_dir, clr_g);
    ### endfor
### else
    #pragma unroll {{ [count, 8]|min }}
    for (int i = 0; i < {{ count }}; ++i)
        {{ tracer.tracer_kernel_name }}(pos, ray, isec_dist, whichobject, which_subobject,
            last_whichobject, last_which_subobject, inside, vector_data, integer_data,
            {{ offset+1 }} + i, clr_params + {{ renderer.group_param_base[group] }} + i*{{ renderer.group_param_stride[group] }}, 0, world_dir, clr_g);
### endif
### endfor
}

// normal of the object that was hit (renderer.py:346-351)
template<class CLR_G>
CLR_FN vec3 clr_surface_normal(const vec3 pos, const vec3 ray, const uint whichobject,
    const uint which_subobject, const uint inside,
    const float4 * __restrict__ vector_data, const int * __restrict__ integer_data, CLR_G &clr_g)
{
### if renderer.bake_parameters and n_objects <= 24
    // one case per object: its parameters are literals, so e.g. the normal of a wall is a constant,
    // and a wave runs only the cases of the objects its lanes are on (no gather from the table)
### for tracer, count, offset in renderer.object_groups
### set group = loop.index0
    ### for i in range(count)
    if (whichobject == {{ offset+1+i }})
        return {{ tracer.normal_kernel_name }}(pos, ray, which_subobject, inside, vector_data, integer_data,
            {{ offset+1+i }}, clr_params + {{ renderer.group_param_base[group] + i*renderer.group_param_stride[group] }}, clr_g);
    ### endfor
### endfor
### else
### for tracer, count, offset in renderer.object_groups
    {% if not loop.first %}else {% endif %}if (whichobject >= {{ offset+1 }} && whichobject < {{ offset+1+count }})
        return {{ tracer.normal_kernel_name }}(pos, ray, which_subobject, inside, vector_data, integer_data,
            (int)whichobject, clr_params + {{ renderer.group_param_base[loop.index0] }} + (whichobject - {{ offset+1 }})*{{ renderer.group_param_stride[loop.index0] }}, clr_g);
### endfor
### endif
    return make_vec3(0.0f, 0.0f, 0.0f);
}

### if renderer.bidirectional
// visibility of the light point from the hit point (renderer.py:373-382, tracer.cl:143-221):
// the sweep itself, along light_dir = clr_make_direction(light point - pos)
template<class CLR_G>
CLR_FN float clr_light_visibility_along(const vec3 pos, const uint whichobject,
    const uint which_subobject, const uint inside, const clr_direction &light_dir, const int light_id,
    const float4 * __restrict__ vector_data, const int * __restrict__ integer_data, CLR_G &clr_g)
{
    clr_guard_upper(light_dir, 0.0f, clr_g);
    const float isec_dist = light_dir.len;
    const vec3 ray = light_dir.unit;
    float shadow_mask = 1.0f;
### if renderer.shares_world_direction
    const clr_direction world_dir = clr_make_direction(ray, clr_g);
### else
    const clr_direction world_dir = {};
### endif
### for tracer, count, offset in renderer.object_groups
### set group = loop.index0
### if renderer.bake_parameters
    ### for i in range(count)
    // (a ray that an earlier object has blocked stays blocked: the remaining tests cannot change it)
    if ({{ offset+1+i }} != light_id && shadow_mask != 0.0f)
        {{ tracer.shadow_kernel_name }}(pos, ray, isec_dist, whichobject, which_subobject, inside,
            shadow_mask, vector_data, integer_data,
            {{ offset+1+i }}, clr_params + {{ renderer.group_param_base[group] + i*renderer.group_param_stride[group] }},
            {{ renderer.affine_skip_mask(offset+i) }}, world_dir, clr_g);
    ### endfor
### else
    #pragma unroll {{ [count, 8]|min }}
    for (int i = 0; i < {{ count }}; ++i)
        if ({{ offset+1 }} + i != light_id && shadow_mask != 0.0f)
            {{ tracer.shadow_kernel_name }}(pos, ray, isec_dist, whichobject, which_subobject, inside,
                shadow_mask, vector_data, integer_data,
                {{ offset+1 }} + i, clr_params + {{ renderer.group_param_base[group] }} + i*{{ renderer.group_param_stride[group] }}, 0, world_dir, clr_g);
### endif
### endfor
    return shadow_mask;
}

// the reference's form: every ray of the segment, with the two early exits of tracer.cl:143-221
template<class CLR_G>
CLR_FN float clr_light_visibility(const vec3 pos, const vec3 normal, const uint whichobject,
    const uint which_subobject, const uint inside, const int diffusions_left,
    const vec3 dest, const int light_id,
    const float4 * __restrict__ vector_data, const int * __restrict__ integer_data, CLR_G &clr_g)
{
    if (diffusions_left < 1) return 0.0f;
    const vec3 to_light = dest - pos;
    if (dot(normal, to_light) < 0.0f) return 0.0f;
    const clr_direction light_dir = clr_make_direction(to_light, clr_g);
    return clr_light_visibility_along(pos, whichobject, which_subobject, inside, light_dir, light_id,
                                      vector_data, integer_data, clr_g);
}
### endif

CLR_FN BounceSample clr_load_bounce(const float4 * __restrict__ rec) {
    BounceSample B;
    const float4 a = rec[0], b = rec[1], c = rec[2], d = rec[3], e = rec[4];
    B.rvec = make_vec3(a); B.p = a.w;
### if renderer.rng_mode != 'philox'
    B.gauss = make_vec3(b);
### endif
    B.russian = b.w;
    B.light_point = make_vec3(c); B.light_area = c.w;
    B.light_normal = make_vec3(d); B.min_light_sampling_distance = d.w;
    B.light_center = make_vec3(e); B.light_id = __float_as_int(e.w);
### if renderer.light_samples > 1 and renderer.rng_mode != 'philox'
    B.more_lights = rec + 5;
### endif
    return B;
}

### if renderer.rng_mode == 'philox'
// the record of one segment from the pixel's own stream (same fields as the tape record); the
// light sample is drawn only where the segment can use it, the gaussian vector inside clr_shade
CLR_FN BounceSample clr_draw_bounce(const uint seed_lo, const uint seed_hi, const uint pixel,
                                    const uint pass, const uint segment, const bool with_light) {
    BounceSample B;
    float rvec[3];
    clr_draw_block0(seed_lo, seed_hi, pixel, pass, segment, rvec, &B.p, &B.russian);
    B.rvec = make_vec3(rvec[0], rvec[1], rvec[2]);
    B.seed_lo = seed_lo; B.seed_hi = seed_hi; B.pixel = pixel; B.pass = pass; B.segment = segment;
    B.light_point = make_vec3(0.0f, 0.0f, 0.0f); B.light_area = 0.0f;
    B.light_normal = make_vec3(0.0f, 0.0f, 0.0f); B.min_light_sampling_distance = 0.0f;
    B.light_center = make_vec3(0.0f, 0.0f, 0.0f); B.light_id = 0;
    ### if renderer.light_samples > 1
    B.light_index = 0;
    ### endif
    ### if renderer.n_lights
    if (with_light) {
        float light_u, light_dir[3], light_uniforms[3];
        clr_draw_light(seed_lo, seed_hi, pixel, pass, segment, &light_u, light_dir, light_uniforms);
        int idx = (int)(light_u * {{ renderer.n_lights|float }}f);
        if (idx > {{ renderer.n_lights - 1 }}) idx = {{ renderer.n_lights - 1 }};
        const float4 L0 = clr_lights[2*idx], L1 = clr_lights[2*idx + 1];
        B.light_normal = make_vec3(light_dir[0], light_dir[1], light_dir[2]);
        B.light_point = make_vec3(L0.x + light_dir[0] * L0.w, L0.y + light_dir[1] * L0.w, L0.z + light_dir[2] * L0.w);
        ### if renderer.has_box_lights
        if (!(L0.w > 0.0f)) {
            float point[3], normal[3];
            clr_box_light_point((const float *)clr_lights + __float_as_int(L1.w), light_uniforms[2],
                                light_uniforms[0], light_uniforms[1], point, normal);
            B.light_point = make_vec3(point[0], point[1], point[2]);
            B.light_normal = make_vec3(normal[0], normal[1], normal[2]);
        }
        ### endif
        B.light_area = L1.x;
        B.min_light_sampling_distance = L1.y;
        B.light_center = make_vec3(L0);
        B.light_id = __float_as_int(L1.z);
        ### if renderer.light_samples > 1
        B.light_index = idx;
        ### endif
    }
    ### endif
    return B;
}
### endif

// primary ray of one pass (static.cl:109-131): rows of the jitter rotation in rec[0..2],
// rec[3] = (dof offset, sharp distance), rec[4] = (camera origin, initial path colour scale)
CLR_FN void clr_start_path(PathState &s, const float4 * __restrict__ rec, const vec3 base_ray) {
    // the five records are fetched together and the depth-of-field branch is a select: as a branch
    // it splits the loads into three dependent round trips per pass (measured again in round 3 with
    // the records in scalar registers: the branch is 4 % slower on scene-dev, 1.4 % on
    // scene-three-objects, profiles/r03_ab_shading.log)
    const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4];
    const vec3 rot_ray = make_vec3(dot(make_vec3(r0), base_ray), dot(make_vec3(r1), base_ray),
                                   dot(make_vec3(r2), base_ray));
    const float sharp_distance = r3.w;
    const vec3 focused = fast_normalize(sharp_distance * rot_ray - make_vec3(r3));
    s.ray.x = sharp_distance > 0.0f ? focused.x : rot_ray.x;
    s.ray.y = sharp_distance > 0.0f ? focused.y : rot_ray.y;
    s.ray.z = sharp_distance > 0.0f ? focused.z : rot_ray.z;
    s.pos = make_vec3(r4);
    s.ray_color = COLOR_ONE * r4.w;
    s.whichobject = 0;
    s.which_subobject = 0;
    s.inside = {{ renderer.root_object_id }};
    s.diffusions_left = {{ renderer.scene.min_bounces }};
    s.suppress_emission = 0;
}

#define CLR_DBG_STRIDE 6   // float4 records per (segment, pixel) in the debug dump

// One path segment: closest hit, normal, light visibility, shading, roulette
// (renderer.py:326-433).  Returns false when the path ends.  clr_g selects the arithmetic of
// the two object sweeps: speculative short sequences or the generic ones (clr_exact.inl).
template<class CLR_G>
CLR_FN bool clr_segment(PathState &s, const BounceSample &B, const bool is_last, const vec3 cmask,
    clr_acc &img, const float * __restrict__ material_row,
    const float4 * __restrict__ vector_data, const int * __restrict__ integer_data,
    float4 * __restrict__ dbg, CLR_G &clr_g)
{
    float isec_dist = CLR_MAX_RAY_LENGTH;
    const uint last_whichobject = s.whichobject, last_which_subobject = s.which_subobject;
    uint whichobject = 0, which_subobject = last_which_subobject;
    clr_closest_hit(s.pos, s.ray, isec_dist, whichobject, which_subobject,
                    last_whichobject, last_which_subobject, s.inside, vector_data, integer_data, clr_g);
    if (whichobject == 0) {
        // open scene: the reference leaves this case undefined; end the path
        if (dbg) { dbg[0] = make_float4(0.0f, 0.0f, __int_as_float((int)s.inside), __int_as_float(s.diffusions_left));
                   dbg[1] = make_float4(isec_dist, 0.0f, 0.0f, 0.0f); }
        return false;
    }
    s.whichobject = whichobject;
    s.which_subobject = which_subobject;
    s.pos += isec_dist * s.ray;
    const vec3 normal = clr_surface_normal(s.pos, s.ray, whichobject, which_subobject, s.inside,
                                           vector_data, integer_data, clr_g);
    float shadow_mask = 0.0f;
### if renderer.bidirectional
    // with a debug dump the visibility of the light point is evaluated for every ray, as the
    // reference does; otherwise inside the diffuse stage, for the lanes that use it (shader.hip)
    if (dbg && !is_last)
        shadow_mask = clr_light_visibility(s.pos, normal, whichobject, which_subobject, s.inside,
            s.diffusions_left, B.light_point, B.light_id, vector_data, integer_data, clr_g);
### endif
    const bool alive = clr_shade(s, normal, isec_dist, shadow_mask, dbg == nullptr, B, is_last, cmask, img,
                                 vector_data, integer_data, clr_g
### if not shader.rgb
        , material_row
### endif
        );
    if (dbg) {
        dbg[0] = make_float4(__int_as_float((int)whichobject), __int_as_float((int)which_subobject),
                             __int_as_float((int)s.inside), __int_as_float(s.diffusions_left));
        dbg[1] = make_float4(isec_dist, alive ? 1.0f : 0.0f, shadow_mask, __int_as_float(s.suppress_emission));
        // state after shading: position (moved back on volume scattering), surface id (0 then)
        dbg[2] = make_float4(s.pos.x, s.pos.y, s.pos.z, __int_as_float((int)s.whichobject));
        dbg[3] = make_float4(s.ray.x, s.ray.y, s.ray.z, 0.0f);
        dbg[4] = make_float4(normal.x, normal.y, normal.z, 0.0f);
        const vec3 c = color_to_rgb(s.ray_color, make_vec3(1.0f, 1.0f, 1.0f));
        dbg[5] = make_float4(c.x, c.y, c.z, 0.0f);
    }
    return alive;
}

### if renderer.speculate
// the replay of a segment with the generic arithmetic: rare, kept out of the render loop
static __device__ __attribute__((noinline)) bool clr_segment_generic(PathState *s, const BounceSample *B,
    const bool is_last, const vec3 cmask, clr_acc *img, const float * __restrict__ material_row,
    const float4 * __restrict__ vector_data, const int * __restrict__ integer_data, float4 * __restrict__ dbg)
{
    return clr_segment(*s, *B, is_last, cmask, *img, material_row, vector_data, integer_data, dbg, clr_g);
}
### endif

// ---------------------------------------------------------------------------------------
// Render kernel.  Lane k owns pixel k (blocked pixel order) and accumulates its radiance in
// registers over `n_passes` sample passes; img[k] is read once and written once.  Two program
// forms (clr_regen_mode; per-pixel results are identical, a lane visits its passes and segments
// in the same order in both):
//   lock-step (default with the broadcast tape): the lanes of a wave start each pass together, so
//       pass and segment numbers are wave-uniform: they live in scalar registers, and the tape
//       records of a pass / a segment arrive through scalar loads into scalar registers instead
//       of 20 + 20 vector registers per lane.  Finished paths idle until the slowest path of the
//       wave ends (with the broadcast tape all lanes take the same stochastic decisions and
//       finish together).
//   regenerating (default with per-pixel random numbers): a lane starts its next pass as soon as
//       its path ends, so every lane of the wave always executes a live segment.
// dbg (optional, n_passes == 1): CLR_DBG_STRIDE float4 per (segment, pixel).
// Pass-split programs (clr_split_mode, small frames): n_padded x pass_groups lanes, n_padded =
// n_pixels rounded up to whole waves, so that a wave never straddles two groups; a lane walks a
// range of the passes of its pixel and logs its addends instead of summing them (clr_acc), img
// is left to clr_reduce_log.
// ---------------------------------------------------------------------------------------
{%- macro run_segment(pass, segment) %}
            const bool is_last = {{ segment }} == CLR_MAX_BOUNCES - 1;
### if renderer.rng_mode == 'philox'
            // the light sample feeds the visibility test and the diffuse stage, both of which
            // need a diffuse bounce to be left and are skipped on the last segment
            const BounceSample B = clr_draw_bounce(rng_seed_lo, rng_seed_hi, global_ray,
                                                   (uint)(first_pass + {{ pass }} * pass_stride), (uint){{ segment }},
                                                   !is_last && s.diffusions_left > 0);
### else
            const BounceSample B = clr_load_bounce(bounce_tape + ((size_t){{ pass }} * CLR_MAX_BOUNCES + {{ segment }}) * CLR_BOUNCE_STRIDE);
### endif
            float4 *d = dbg ? dbg + ((size_t){{ segment }} * n_pixels + k) * CLR_DBG_STRIDE : nullptr;
            const float *material_row = material_rows + (size_t){{ pass }} * CLR_MATERIAL_ROW;
{%- endmacro %}
{#- Sorted form of clr_render (per-pixel random numbers, one lane per pixel): the paths of a block
    change lanes before every segment so that the lanes of a wave do like work.  No counterpart in
    the reference, whose compaction (renderer.py:415-431) only closes the gaps that dead rays leave;
    same results, because a segment is a pure function of (pixel, pass, segment) and a pixel's sums
    are added in the order of its one live path. -#}
{%- macro sorted_render_body() %}
    {
        // Every path carries its pixel (slot in the block), pass and segment.  Before each segment the
        // 256 paths of the block are sorted (stable counting sort over 16 keys: ballots inside a wave,
        // a 64-entry table across the four waves) by what the segment will do: primary rays first -
        // neighbouring pixels, the same walls and objects - then the other rays by the octant of their
        // direction, which decides the walls a ray inside a room can reach, then lanes without a
        // path, so that whole waves fall out at the end of a launch.  The state travels through LDS.
        static_assert(CLR_MAX_BOUNCES <= 15, "4 bits of segment number");
        __shared__ float clr_pixel_rays[3 * 256];
### if renderer.speculate
        __shared__ float clr_pixel_sums_at_start[3 * 256];
### endif
        __shared__ uint clr_path_words[{{ 14 if shader.rgb else 12 }} * 256];
        __shared__ int clr_sort_table[64];
        const int tid = threadIdx.x, wave = tid >> 6;
        const unsigned long long lanes_before = (1ull << (tid & 63)) - 1ull;
        clr_pixel_rays[tid] = base_ray.x; clr_pixel_rays[256 + tid] = base_ray.y; clr_pixel_rays[512 + tid] = base_ray.z;
        clr_pixel_sums[tid] = img0.x; clr_pixel_sums[256 + tid] = img0.y; clr_pixel_sums[512 + tid] = img0.z;
        if (tid < 64) clr_sort_table[tid] = 0;
        PathState s;
        s.pos = s.ray = make_vec3(0.0f, 0.0f, 0.0f); s.ray_color = COLOR_ONE;
        s.whichobject = s.which_subobject = s.inside = 0; s.diffusions_left = s.suppress_emission = 0;
        int slot = tid, pass = in_image ? 0 : n_passes, segment = 0;
        bool active = false, generic_path = false;
        __syncthreads();
        for (;;) {
            // a finished path makes way for the next pass of the same pixel
            if (!active && pass < n_passes) {
                clr_start_path(s, pass_tape + (size_t)pass * CLR_PASS_STRIDE,
                               make_vec3(clr_pixel_rays[slot], clr_pixel_rays[256 + slot], clr_pixel_rays[512 + slot]));
                segment = 0;
                active = true;
                generic_path = false;
### if renderer.speculate
                clr_pixel_sums_at_start[slot] = clr_pixel_sums[slot];
                clr_pixel_sums_at_start[256 + slot] = clr_pixel_sums[256 + slot];
                clr_pixel_sums_at_start[512 + slot] = clr_pixel_sums[512 + slot];
### endif
            }
            // ---- sort: destination = paths with a smaller key + paths with this key in earlier waves
            // + such paths in earlier lanes of this wave
            const int key = !active ? 15 : segment == 0 ? 0
                : 1 + (s.ray.x < 0.0f ? 1 : 0) + (s.ray.y < 0.0f ? 2 : 0) + (s.ray.z < 0.0f ? 4 : 0);
            unsigned long long same_key = ~0ull;
            #pragma unroll
            for (int bit = 0; bit < 4; ++bit) {
                const unsigned long long plane = __ballot((key >> bit) & 1);
                same_key &= ((key >> bit) & 1) ? plane : ~plane;
            }
            const int rank_in_wave = __popcll(same_key & lanes_before);
            if (rank_in_wave == 0) clr_sort_table[key * 4 + wave] = __popcll(same_key);
            __syncthreads();
            if (tid < 64) {
                // exclusive prefix sums of the 64 counts (key-major, wave-minor), in place
                const int count = clr_sort_table[tid];
                int sum = count;
                #pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int below = __shfl_up(sum, d);
                    if (tid >= d) sum += below;
                }
                clr_sort_table[tid] = sum - count;
            }
            __syncthreads();
            const int to = clr_sort_table[key * 4 + wave] + rank_in_wave;
            const bool none_active = clr_sort_table[15 * 4] == 0;      // the lanes without a path start at 0: all of them
            {
                uint *w = clr_path_words + to;
                w[0 * 256] = __float_as_uint(s.pos.x); w[1 * 256] = __float_as_uint(s.pos.y); w[2 * 256] = __float_as_uint(s.pos.z);
                w[3 * 256] = __float_as_uint(s.ray.x); w[4 * 256] = __float_as_uint(s.ray.y); w[5 * 256] = __float_as_uint(s.ray.z);
                w[6 * 256] = s.whichobject | (s.inside << 16);
                w[7 * 256] = s.which_subobject;
                w[8 * 256] = (uint)(s.diffusions_left & 0xff) | ((uint)s.suppress_emission << 8);
                w[9 * 256] = (uint)slot | ((uint)segment << 8) | (generic_path ? 1u << 12 : 0u) | (active ? 1u << 13 : 0u);
                w[10 * 256] = (uint)pass;
### if shader.rgb
                w[11 * 256] = __float_as_uint(s.ray_color.x); w[12 * 256] = __float_as_uint(s.ray_color.y); w[13 * 256] = __float_as_uint(s.ray_color.z);
### else
                w[11 * 256] = __float_as_uint(s.ray_color);
### endif
            }
            __syncthreads();
            if (none_active) break;
            {
                const uint *w = clr_path_words + tid;
                s.pos = make_vec3(__uint_as_float(w[0 * 256]), __uint_as_float(w[1 * 256]), __uint_as_float(w[2 * 256]));
                s.ray = make_vec3(__uint_as_float(w[3 * 256]), __uint_as_float(w[4 * 256]), __uint_as_float(w[5 * 256]));
                s.whichobject = w[6 * 256] & 0xffffu; s.inside = w[6 * 256] >> 16;
                s.which_subobject = w[7 * 256];
                s.diffusions_left = (int)(w[8 * 256] & 0xffu); s.suppress_emission = (int)w[8 * 256] >> 8;
                slot = (int)(w[9 * 256] & 0xffu); segment = (int)((w[9 * 256] >> 8) & 0xfu);
                generic_path = (w[9 * 256] >> 12) & 1u; active = (w[9 * 256] >> 13) & 1u;
                pass = (int)w[10 * 256];
### if shader.rgb
                s.ray_color = make_vec3(__uint_as_float(w[11 * 256]), __uint_as_float(w[12 * 256]), __uint_as_float(w[13 * 256]));
### else
                s.ray_color = __uint_as_float(w[11 * 256]);
### endif
            }
            // every wave clears its own column of the table (the entries only it writes) for the next round
            if ((tid & 63) < 16) clr_sort_table[(tid & 63) * 4 + wave] = 0;
            if (active) {
                clr_acc acc;
                acc.sum = clr_pixel_sums + slot;
                const int k = blockIdx.x * blockDim.x + slot;          // the pixel of this path
### if renderer.rng_mode == 'philox'
                const int local_ray = first_ray + k;
                const uint global_ray = ray_band > 0
                    ? (uint)(ray_band_base + (local_ray / ray_band) * ray_band_stride + local_ray % ray_band) : (uint)local_ray;
### endif
                const vec3 cmask = make_vec3(pass_tape[(size_t)pass * CLR_PASS_STRIDE + 5]);
                {{ run_segment('pass', 'segment') }}
### if renderer.speculate
                bool alive;
                bool again = false;
                if (!generic_path) {
                    clr_guard_spec spec;
                    clr_guard_reset(spec);
                    alive = clr_segment(s, B, is_last, cmask, acc, material_row, vector_data, integer_data, d, spec);
                    if (!clr_guard_ok(spec)) {
                        // operands left the range of the short sequences: the pixel's sums go back to
                        // the start of the pass and the path is walked again with the generic arithmetic
                        acc.set(make_vec3(clr_pixel_sums_at_start[slot], clr_pixel_sums_at_start[256 + slot], clr_pixel_sums_at_start[512 + slot]));
                        clr_start_path(s, pass_tape + (size_t)pass * CLR_PASS_STRIDE,
                                       make_vec3(clr_pixel_rays[slot], clr_pixel_rays[256 + slot], clr_pixel_rays[512 + slot]));
                        segment = 0;
                        generic_path = true;
                        again = true;
                    }
                }
                else {
                    PathState s1 = s;
                    clr_acc acc1 = acc;
                    BounceSample B1 = B;
                    alive = clr_segment_generic(&s1, &B1, is_last, cmask, &acc1, material_row, vector_data, integer_data, d);
                    s = s1;
                }
                if (!again) {
                    segment++;
                    if (!alive) { active = false; pass++; }
                }
### else
                const bool alive = clr_segment(s, B, is_last, cmask, acc, material_row, vector_data, integer_data, d, clr_g);
                segment++;
                if (!alive) { active = false; pass++; }
### endif
            }
        }
        if (in_image) img[k] = make_float4(clr_pixel_sums[tid], clr_pixel_sums[256 + tid], clr_pixel_sums[512 + tid], img0.w);
### if renderer.profile_blocks
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 65536)
            atomicMax(&clr_block_cycles[blockIdx.x], (unsigned long long)((long long)__builtin_readcyclecounter() - clr_block_begin));
### endif
        return;
    }
{%- endmacro %}
// The kernel exists twice: clr_render, and clr_render_debug for launches with a debug dump.  In
// clr_render the dump pointer is a null constant, so the stores of the dump and the up-front
// light visibility sweep that the dump needs (clr_segment) are not part of its code at all.
{%- macro render_kernel(kernel_name, debug) %}
extern "C" __global__ void __launch_bounds__(256{{ ', %d' % renderer.min_waves_per_simd if renderer.min_waves_per_simd else '' }})
{{ kernel_name }}(float4 * __restrict__ img, const float4 * __restrict__ cam,
           const float4 * __restrict__ pass_tape, const float4 * __restrict__ bounce_tape,
           const float * __restrict__ material_rows,
           const float4 * __restrict__ vector_data, const int * __restrict__ integer_data,
           const int n_pixels, const int n_passes, const int regen, float4 * __restrict__ dbg_argument,
           const uint rng_seed_lo, const uint rng_seed_hi, const int first_pass, const int first_ray,
           float * __restrict__ pass_log, const int pass_groups, const int pass_stride,
           const int ray_band, const int ray_band_stride, const int ray_band_base)
{
    float4 * __restrict__ const dbg = {{ 'dbg_argument' if debug else 'nullptr' }};
### if renderer.profile_blocks
    const long long clr_block_begin = (long long)__builtin_readcyclecounter();
### endif
### if renderer.split_passes
    // pass-split mode: lane (k, group) walks the passes [group * per_group, ...) of pixel k and
    // logs what it would have added to the pixel (layout: clr_acc, shader.hip)
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_padded = (n_pixels + 63) & ~63;
    const int k = lane % n_padded;
    const int group = __builtin_amdgcn_readfirstlane(lane / n_padded);      // the same for a whole wave
    const bool in_image = group < pass_groups && k < n_pixels;
    const int per_group = (n_passes + pass_groups - 1) / pass_groups;
    const int pass_begin = group < pass_groups ? min(group * per_group, n_passes) : n_passes;
    const int pass_end = group < pass_groups ? min(pass_begin + per_group, n_passes) : n_passes;
    const int kk = in_image ? k : 0;
    const vec3 base_ray = make_vec3(cam[kk]);
### else
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_image = k < n_pixels;
    const int pass_begin = 0, pass_end = n_passes;
    const int kk = in_image ? k : 0;
    const vec3 base_ray = make_vec3(cam[kk]);
    const float4 img0 = img[kk];
### endif
### if shader.rgb and renderer.materials_in_lds
    for (int i = threadIdx.x; i < {{ 5*(n_objects+1) }}; i += blockDim.x) {
        // .w = the probability of the stage that the colour belongs to (COLOR2PROB, shader.hip): a
        // function of the material alone, divided once here instead of in every segment
        float4 c = clr_material_colors[i];
        c.w = COLOR2PROB(make_vec3(c));
        clr_lds_material_colors[i] = c;
        // what a specular stage multiplies the path's colour by when it is chosen (never read
        // where the probability is zero: such a stage is not chosen)
        const vec3 ratio = make_vec3(c) / c.w;
        clr_lds_material_ratios[i] = make_float4(ratio.x, ratio.y, ratio.z, 0.0f);
        clr_lds_material_scalars[i] = clr_material_scalars[i];
    }
    __syncthreads();
### endif
### if renderer.rng_mode == 'philox'
    // the ray's index in the whole frame keys its random numbers.  A tile is a contiguous range of
    // rays (first_ray = index of its first ray) or, ray_band > 0, every N-th band of ray_band rays:
    // local ray L = first_ray + k is ray L % ray_band of band L / ray_band of this tile, and the
    // tile's bands start ray_band_stride rays apart from ray_band_base on (multi_gpu.py)
    const int local_ray = first_ray + k;
    const uint global_ray = ray_band > 0
        ? (uint)(ray_band_base + (local_ray / ray_band) * ray_band_stride + local_ray % ray_band) : (uint)local_ray;
### endif
### if renderer.sorted_paths
    __shared__ float clr_pixel_sums[3 * 256];
### if not debug
    {{ sorted_render_body() }}
### endif
### endif
    clr_acc acc;
### if renderer.split_passes
    acc.stride = (size_t)n_pixels;
    acc.begin_pass(pass_log, 0, kk);
### else
### if renderer.sorted_paths
    acc.sum = clr_pixel_sums + threadIdx.x;
### endif
    acc.set(make_vec3(img0.x, img0.y, img0.z));
### endif

    PathState s;
    vec3 cmask = make_vec3(1.0f, 1.0f, 1.0f);
### if renderer.speculate and not renderer.split_passes
    vec3 acc_at_pass_start = acc.get();
### endif

### if not renderer.regenerate
    // ---- lock-step form
    for (int pass = pass_begin; pass < pass_end; ++pass) {
        const float4 *rec = pass_tape + (size_t)pass * CLR_PASS_STRIDE;
        clr_start_path(s, rec, base_ray);
        cmask = make_vec3(rec[5]);
        bool active = in_image;
    ### if renderer.split_passes
        acc.begin_pass(pass_log, pass, kk);
    ### endif
    ### if renderer.speculate
        ### if not renderer.split_passes
        acc_at_pass_start = acc.get();
        ### endif
        // a lane whose operands left the range of the speculative arithmetic abandons the pass
        // and walks it again below with the generic arithmetic
        bool redo = false;
    ### endif
        for (int segment = 0; segment < CLR_MAX_BOUNCES; ++segment) {
            if (!__any(active)) break;
            if (active) {
                {{ run_segment('pass', 'segment') }}
    ### if renderer.speculate
                clr_guard_spec spec;
                clr_guard_reset(spec);
                bool alive = clr_segment(s, B, is_last, cmask, acc, material_row, vector_data, integer_data, d, spec);
                if (!clr_guard_ok(spec)) { redo = true; alive = false; }
    ### else
                const bool alive = clr_segment(s, B, is_last, cmask, acc, material_row, vector_data, integer_data, d, clr_g);
    ### endif
                active = alive;
            }
        }
    ### if renderer.speculate
        if (__any(redo)) {                      // rare (none on scene-dev, 2e-6 per sample on scene-three-objects)
            if (redo) {
        ### if renderer.split_passes
                acc.restart_pass();
        ### else
                acc.set(acc_at_pass_start);
        ### endif
                clr_start_path(s, rec, base_ray);
                for (int segment = 0; segment < CLR_MAX_BOUNCES; ++segment) {
                    {{ run_segment('pass', 'segment') }}
                    // only copies of the state escape to the callee
                    PathState s1 = s;
                    clr_acc acc1 = acc;
                    BounceSample B1 = B;
                    const bool alive = clr_segment_generic(&s1, &B1, is_last, cmask, &acc1, material_row, vector_data, integer_data, d);
                    s = s1;
                    acc = acc1;
                    if (!alive) break;
                }
            }
        }
    ### endif
    ### if renderer.split_passes
        if (in_image) acc.end_pass();
    ### endif
    }
### else
    // ---- regenerating form: every lane is its own little state machine
    int pass = in_image ? pass_begin : pass_end;
    int segment = 0;
    bool active = false;
    ### if renderer.speculate
    bool generic_path = false;
    ### endif
    for (;;) {
        if (!active && pass < pass_end) {
            const float4 *rec = pass_tape + (size_t)pass * CLR_PASS_STRIDE;
            clr_start_path(s, rec, base_ray);
            cmask = make_vec3(rec[5]);
            segment = 0;
            active = true;
    ### if renderer.split_passes
            acc.begin_pass(pass_log, pass, kk);
    ### endif
    ### if renderer.speculate
        ### if not renderer.split_passes
            acc_at_pass_start = acc.get();
        ### endif
            generic_path = false;
    ### endif
        }
        if (!__any(active)) break;
        if (active) {
            {{ run_segment('pass', 'segment') }}
    ### if renderer.speculate
            // Speculative arithmetic first.  If an operand leaves the range in which it is exact,
            // the lane takes the radiance it had at the start of the pass, starts the path of
            // this pass again and walks it with the generic arithmetic (rare; out of line).
            bool alive;
            if (!generic_path) {
                clr_guard_spec spec;
                clr_guard_reset(spec);
                alive = clr_segment(s, B, is_last, cmask, acc, material_row, vector_data, integer_data, d, spec);
                if (!clr_guard_ok(spec)) {
        ### if renderer.split_passes
                    acc.restart_pass();
        ### else
                    acc.set(acc_at_pass_start);
        ### endif
                    clr_start_path(s, pass_tape + (size_t)pass * CLR_PASS_STRIDE, base_ray);
                    segment = 0;
                    generic_path = true;
                    continue;
                }
            }
            else {
                // only copies of the state escape to the callee
                PathState s1 = s;
                clr_acc acc1 = acc;
                BounceSample B1 = B;
                alive = clr_segment_generic(&s1, &B1, is_last, cmask, &acc1, material_row, vector_data, integer_data, d);
                s = s1;
                acc = acc1;
            }
    ### else
            const bool alive = clr_segment(s, B, is_last, cmask, acc, material_row, vector_data, integer_data, d, clr_g);
    ### endif
            segment++;
            if (!alive) {
    ### if renderer.split_passes
                acc.end_pass();
    ### endif
                active = false;
                pass++;
            }
        }
    }
### endif
### if not renderer.split_passes
    if (in_image) { const vec3 total = acc.get(); img[k] = make_float4(total.x, total.y, total.z, img0.w); }
### endif
### if renderer.profile_blocks
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 65536)
        atomicMax(&clr_block_cycles[blockIdx.x], (unsigned long long)((long long)__builtin_readcyclecounter() - clr_block_begin));
### endif
}
{%- endmacro %}
{{ render_kernel('clr_render', false) }}
{{ render_kernel('clr_render_debug', true) }}

### if renderer.split_passes
// Second step of pass-split mode: the logged addends of every pass, added to the pixel in pass
// order and, inside a pass, in the order in which the path produced them.
extern "C" __global__ void __launch_bounds__(256)
clr_reduce_log(float4 * __restrict__ img, const float * __restrict__ pass_log,
               const int n_pixels, const int n_passes)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pixels) return;
    const float4 img0 = img[k];
    vec3 acc = make_vec3(img0.x, img0.y, img0.z);
    for (int pass = 0; pass < n_passes; ++pass) {
        const float *header = pass_log + (size_t)pass * CLR_LOG_FLOATS * n_pixels + k;
        const float *entries = header - k + n_pixels + 3 * (size_t)k;
        const int word = __float_as_int(*header);
        const int n = word & 0xff, first_plus_zero = word >> 8;
        for (int i = 0; i <= n; ++i) {
            if (first_plus_zero == i + 1) acc += make_vec3(0.0f, 0.0f, 0.0f);
            if (i < n) {
                const float *e = entries + (size_t)i * 3 * n_pixels;
                acc += make_vec3(e[0], e[1], e[2]);
            }
        }
    }
    img[k] = make_float4(acc.x, acc.y, acc.z, img0.w);
}
### endif

// First-hit AOV for the primary rays of pass record 0: object id, sub-object, distance, normal.
extern "C" __global__ void __launch_bounds__(256)
clr_first_hit(const float4 * __restrict__ cam, const float4 * __restrict__ pass_tape,
              const float4 * __restrict__ vector_data, const int * __restrict__ integer_data,
              const int n_pixels, int * __restrict__ out_ids, float * __restrict__ out_t,
              float4 * __restrict__ out_normal)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pixels) return;
    PathState s;
    clr_start_path(s, pass_tape, make_vec3(cam[k]));
    float isec_dist = CLR_MAX_RAY_LENGTH;
    uint whichobject = 0, which_subobject = 0;
    clr_closest_hit(s.pos, s.ray, isec_dist, whichobject, which_subobject, 0, 0, s.inside,
                    vector_data, integer_data, clr_g);
    vec3 normal = make_vec3(0.0f, 0.0f, 0.0f);
    if (whichobject != 0)
        normal = clr_surface_normal(s.pos + isec_dist * s.ray, s.ray, whichobject, which_subobject,
                                    s.inside, vector_data, integer_data, clr_g);
    out_ids[2*k] = (int)whichobject;
    out_ids[2*k+1] = (int)which_subobject;
    out_t[k] = isec_dist;
    out_normal[k] = make_float4(normal.x, normal.y, normal.z, 0.0f);
}
