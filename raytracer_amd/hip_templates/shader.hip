{#- Material model and the per-segment shading pipeline, fused into one __device__ function.
    Counterpart of the reference's shader.cl (/root/reference/cl_templates/shader.cl:2-434),
    where each stage (volumetric, emission, reflection, refraction, diffuse) and the culler are
    separate kernels that communicate through global buffers.  The stage protocol is kept:
    one shared sample p in [0,1) is consumed stage by stage; the stage that "fires" sets
    p = -1, commits the new direction and multiplies the path colour by pipeline_color. -#}

### for item in shader.get_material_property_offsets()
#define MAT_{{item[0]}} {{item[1]}}
### endfor

### if shader.rgb
typedef vec3 color_t;
### if renderer.materials_in_lds
#define COLOR(prop, id) make_vec3(clr_lds_material_colors[(prop) + (id)])
#define SCALAR(prop, id) clr_lds_material_scalars[(prop) + (id)]
// COLOR2PROB of the same entry, computed when the table was copied to LDS (main.hip)
#define STAGE_PROB(prop, id) clr_lds_material_colors[(prop) + (id)].w
#define STAGE_RATIO(prop, id) make_vec3(clr_lds_material_ratios[(prop) + (id)])
### else
#define COLOR(prop, id) make_vec3(clr_material_colors[(prop) + (id)])
#define SCALAR(prop, id) clr_material_scalars[(prop) + (id)]
#define STAGE_PROB(prop, id) COLOR2PROB(COLOR(prop, id))
#define STAGE_RATIO(prop, id) (COLOR(prop, id) / COLOR2PROB(COLOR(prop, id)))
### endif
#define COLOR2PROB(color) (dot(color, make_vec3(1.0f, 1.0f, 1.0f))/3.0f)
#define COLOR_ONE make_vec3(1.0f, 1.0f, 1.0f)
CLR_FN vec3 color_to_rgb(vec3 c, vec3 cmask) { return c; }
### else
typedef float color_t;
#define COLOR(prop, id) material_scalars[(prop) + (id)]
#define COLOR2PROB(color) (color)
#define STAGE_PROB(prop, id) COLOR(prop, id)
#define SCALAR(prop, id) material_scalars[(prop) + (id)]
#define COLOR_ONE 1.0f
CLR_FN vec3 color_to_rgb(float c, vec3 cmask) { return c * cmask; }
### endif

#define CLR_PI_F 3.14159274101257324f

// Where a path's radiance goes.  Normally a running sum in registers.  In pass-split mode
// (small frames: several lanes per pixel, each with a range of passes; main.hip) the addends are
// written to a log in the order of the additions, and clr_reduce_log adds them to the pixel in
// pass order afterwards - the same additions in the same order, hence the same bits.
// Most addends are (+0, +0, +0) (the emission of a surface that does not emit) and are not logged:
// adding +0 changes a sum only while it is -0, and after the first such addition no component can
// become -0 again (x + y = -0 needs x = y = -0), so the log keeps just the position of the first
// one.  Per (pass, pixel): one header word, n | (1 + entries before the first +0 addend, 0 = none)
// << 8, in a plane of n_pixels words, then the entries, 3 floats each, in planes of 3 n_pixels.
struct clr_acc {
### if renderer.split_passes
    float *header;        // this lane's header word of the current pass
    float *entries;       // its entry 0; entry i is 3 i stride floats further on
    size_t stride;        // n_pixels
    int n, first_plus_zero;
    CLR_FN_MEMBER void operator+=(const vec3 a) {
        if ((__float_as_int(a.x) | __float_as_int(a.y) | __float_as_int(a.z)) == 0) {
            if (first_plus_zero == 0) first_plus_zero = n + 1;
            return;
        }
        float *e = entries + (size_t)n * 3 * stride;
        e[0] = a.x; e[1] = a.y; e[2] = a.z;
        n++;
    }
    CLR_FN_MEMBER void begin_pass(float *pass_log, const int pass, const int k) {
        header = pass_log + (size_t)pass * CLR_LOG_FLOATS * stride + k;
        entries = header - k + stride + 3 * (size_t)k;
        n = 0;
        first_plus_zero = 0;
    }
    CLR_FN_MEMBER void restart_pass() { n = 0; first_plus_zero = 0; }
    CLR_FN_MEMBER void end_pass() { *header = __int_as_float(n | (first_plus_zero << 8)); }
    CLR_FN_MEMBER vec3 get() const { return make_vec3(0.0f, 0.0f, 0.0f); }
    CLR_FN_MEMBER void set(const vec3) {}
### elif renderer.sorted_paths
    // paths change lanes between segments (main.hip, sorted form): the sum of a pixel stays put in
    // LDS, three planes of 256 floats per block, and whichever lane holds the pixel's path adds to it
    // (one path per pixel at a time, so the additions keep their order)
    float *sum;
    CLR_FN_MEMBER void operator+=(const vec3 a) { sum[0] += a.x; sum[256] += a.y; sum[512] += a.z; }
    CLR_FN_MEMBER vec3 get() const { return make_vec3(sum[0], sum[256], sum[512]); }
    CLR_FN_MEMBER void set(const vec3 a) { sum[0] = a.x; sum[256] = a.y; sum[512] = a.z; }
### else
    vec3 v;
    CLR_FN_MEMBER void operator+=(const vec3 a) { v += a; }
    CLR_FN_MEMBER vec3 get() const { return v; }
    CLR_FN_MEMBER void set(const vec3 a) { v = a; }
### endif
};

struct PathState {
    vec3 pos, ray;
    color_t ray_color;
    uint whichobject, which_subobject;   // surface the path is on (0 = none)
    uint inside;                         // object the path travels in (0 = ambient)
    int diffusions_left;
    int suppress_emission;
};

struct BounceSample {                    // one record of the random tape
    vec3 rvec; float p;                  // unit vector, stage sample
    float russian;                       // Russian-roulette sample
### if renderer.rng_mode == 'philox'
    // the gaussian vector is block 1 of the pixel's stream: drawn where a blurred stage uses it
    uint seed_lo, seed_hi, pixel, pass, segment;
    CLR_FN_MEMBER vec3 gauss_vector() const {
        float g[3];
        clr_draw_gauss(seed_lo, seed_hi, pixel, pass, segment, g);
        return make_vec3(g[0], g[1], g[2]);
    }
### else
    vec3 gauss;                          // gaussian vector
    CLR_FN_MEMBER vec3 gauss_vector() const { return gauss; }
### endif
    vec3 light_point; float light_area;
    vec3 light_normal; float min_light_sampling_distance;
    vec3 light_center; int light_id;     // 0 on the last segment / without lights
### if renderer.light_samples > 1
    // further points of the same light (light_samples > 1): i = 1 .. light_samples - 1
    ### if renderer.rng_mode == 'philox'
    int light_index;                     // row of clr_lights
    CLR_FN_MEMBER void light_sample(const int i, vec3 &point, vec3 &normal) const {
        clr_draw_light_point(seed_lo, seed_hi, pixel, pass, segment, i, light_index, point, normal);
    }
    ### else
    const float4 *more_lights;           // (point, normal) pairs behind the record in the tape
    CLR_FN_MEMBER void light_sample(const int i, vec3 &point, vec3 &normal) const {
        point = make_vec3(more_lights[2*i - 2]);
        normal = make_vec3(more_lights[2*i - 1]);
    }
    ### endif
### endif
};

// length(v) > m as the reference evaluates it (shader.cl:313: correctly rounded square root of the
// left-to-right dot product), decided on the squares where that is safe.  With m2 = RN(m m) and
// eps = 2^-24: d2 > m2 (1 + 2^-20) puts sqrt(d2) above m (1 + 2^-22), more than one unit in the last
// place above m, so the rounded root exceeds m; d2 < m2 (1 - 2^-20) puts it more than one unit
// below.  In between, and where a product is not a normal number (the comparisons then fail),
// the root is taken.
CLR_FN bool clr_farther_than(const vec3 v, const float m) {
    const float d2 = dot(v, v), m2 = m * m;
    const bool normal_range = (m2 > 0x1p-100f) & (m2 < 0x1p+100f);
    const bool farther = normal_range & (d2 > m2 * 1.00000095f);
    const bool nearer = normal_range & (d2 < m2 * 0.99999905f);
    if (farther | nearer) return farther;
    return sqrtf(d2) > m;
}

// Runs the stage pipeline for one path segment that ended at s.pos (surface s.whichobject,
// normal n, segment length last_dist).  Adds radiance to `img`.  Returns false if the path
// is finished (last segment, or culled).
// The visibility of the light point (the second sweep over all objects, main.hip) is needed by
// one term only: the next-event estimate inside the diffuse stage.  It is therefore evaluated
// there, for the lanes that get that far (lazy_visibility; a path that ends on a surface
// without a diffuse part - sky, lamp - or leaves by a mirror reflection never asks for it; same
// operands, same result: no stage before the diffuse one changes position, surface or medium
// without ending the pipeline).  With a debug dump the sweep runs up front as in the reference,
// because the dump records the mask of every ray.
template<class CLR_G>
CLR_FN bool clr_shade(PathState &s, vec3 n, const float last_distance, const float eager_shadow_mask,
                      const bool lazy_visibility,
                      const BounceSample &B, const bool is_last, const vec3 cmask, clr_acc &img,
                      const float4 * __restrict__ vector_data, const int * __restrict__ integer_data, CLR_G &clr_g
### if not shader.rgb
                      , const float * __restrict__ material_scalars
### endif
                      )
{
    float p = B.p;
    vec3 r = s.ray;
    float cur_prob = 0.0f;
    color_t cur_col;
    color_t pipeline_color;

### if renderer.materials_in_lds
    // properties of the medium and of the surface: looked up in LDS where a stage needs them
    #define m_scattering SCALAR(MAT_VOLUME_SCATTERING, s.inside)
    #define m_scattering_blur SCALAR(MAT_VOLUME_SCATTERING_BLUR, s.inside)
    #define m_absorption COLOR(MAT_VOLUME_ABSORPTION, s.inside)
    #define m_emission COLOR(MAT_EMISSION, s.whichobject)
    #define m_reflection COLOR(MAT_REFLECTION, s.whichobject)
    #define m_transparency COLOR(MAT_TRANSPARENCY, s.whichobject)
    #define m_diffuse COLOR(MAT_DIFFUSE, s.whichobject)
    #define m_reflection_blur SCALAR(MAT_REFLECTION_BLUR, s.whichobject)
    #define m_transparency_blur SCALAR(MAT_TRANSPARENCY_BLUR, s.whichobject)
    #define m_ior SCALAR(MAT_IOR, s.whichobject)
    #define m_ior_ambient SCALAR(MAT_IOR, 0)
    #define m_light_emission COLOR(MAT_EMISSION, B.light_id)
### else
    // Properties of the medium and of the surface come from global memory here: fetched together
    // up front, because inside the stages they would be one dependent look-up after another (the
    // stages are sequential by construction; with the tables in LDS that costs less than the
    // registers do).  No stage changes s.whichobject or s.inside before it has used them.
    const float m_scattering = SCALAR(MAT_VOLUME_SCATTERING, s.inside);
    const float m_scattering_blur = SCALAR(MAT_VOLUME_SCATTERING_BLUR, s.inside);
    const color_t m_absorption = COLOR(MAT_VOLUME_ABSORPTION, s.inside);
    const color_t m_emission = COLOR(MAT_EMISSION, s.whichobject);
    const color_t m_reflection = COLOR(MAT_REFLECTION, s.whichobject);
    const color_t m_transparency = COLOR(MAT_TRANSPARENCY, s.whichobject);
    const color_t m_diffuse = COLOR(MAT_DIFFUSE, s.whichobject);
    const float m_reflection_blur = SCALAR(MAT_REFLECTION_BLUR, s.whichobject);
    const float m_transparency_blur = SCALAR(MAT_TRANSPARENCY_BLUR, s.whichobject);
    const float m_ior = SCALAR(MAT_IOR, s.whichobject), m_ior_ambient = SCALAR(MAT_IOR, 0);
    ### if renderer.bidirectional
    const color_t m_light_emission = COLOR(MAT_EMISSION, B.light_id);     // id 0: the ambient entry
    ### endif
### endif

    // ---- volumetric effects (shader.cl:124-170)
    {
        float last_dist = last_distance;
        const float alpha = m_scattering;
        if (alpha > 0) cur_prob = 1.0f - clr_expf(-alpha*last_dist);

        if (p < cur_prob) {
            // scattering inside the medium at distance d < last_dist along the segment
            float d = -clr_logf(1.0f - p) / alpha;
            s.pos -= (last_dist - d) * r;
            last_dist = d;
            s.whichobject = 0;
            const float blur = m_scattering_blur;
            if (blur < 1.0f) r = normalize(B.gauss_vector() + r * clr_tanf(CLR_PI_F*0.5f*(1.0f - blur)));
            else r = B.rvec;
### if renderer.bidirectional
            s.suppress_emission = 0;
### endif
            p = -1.0f;
        }
        else p -= cur_prob;

        // A medium that absorbs nothing (air: most segments of most scenes) has exp(-0 * d) = 1:
        // clr_expf returns exactly 1.0f for +-0 (k = 0, r = 0, the polynomial ends in p * 0 + 1), and
        // last_dist is a finite distance, so the three ~30-instruction evaluations are skipped.
### if shader.rgb
        cur_col = m_absorption;
        pipeline_color = COLOR_ONE;
        if ((cur_col.x != 0.0f) | (cur_col.y != 0.0f) | (cur_col.z != 0.0f))
            pipeline_color = make_vec3(clr_expf(-cur_col.x*last_dist), clr_expf(-cur_col.y*last_dist), clr_expf(-cur_col.z*last_dist));
### else
        pipeline_color = COLOR_ONE;
        {
            const float absorption = m_absorption;
            if (absorption != 0.0f) pipeline_color = clr_expf(-absorption*last_dist);
        }
### endif
        if (p < 0.0f) { s.ray = r; s.ray_color = s.ray_color * pipeline_color; }
    }

    // ---- emission (shader.cl:172-188)
    if (p >= 0.0f) {
### if renderer.bidirectional
        const int suppressed = s.suppress_emission;
        if ((suppressed >= 0 && s.whichobject != (uint)suppressed) || s.whichobject == (uint)(-suppressed))
### endif
        {
            img += color_to_rgb(m_emission * s.ray_color * pipeline_color, cmask);
        }
    }

    if (is_last) return false;

    // ---- specular reflection (shader.cl:190-217)
    if (p >= 0.0f) {
        cur_prob = STAGE_PROB(MAT_REFLECTION, s.whichobject);
        if (p < cur_prob) {
### if shader.rgb
            pipeline_color *= STAGE_RATIO(MAT_REFLECTION, s.whichobject);
### endif
            r -= 2*dot(r, n) * n;
            const float blur = m_reflection_blur;
            if (blur > 0) {
                vec3 gauss_rvec = B.gauss_vector();
                if (dot(n, gauss_rvec) < 0) gauss_rvec = -gauss_rvec;
                r = normalize(gauss_rvec * blur + r * (1.0f - blur));
            }
### if renderer.bidirectional
            s.suppress_emission = 0;
### endif
            p = -1.0f;
            s.ray = r; s.ray_color = s.ray_color * pipeline_color;
        }
        else p -= cur_prob;
    }

    // ---- refraction / transparency (shader.cl:219-288)
    if (p >= 0.0f) {
        cur_prob = STAGE_PROB(MAT_TRANSPARENCY, s.whichobject);
        if (p < cur_prob) {
### if shader.rgb
            pipeline_color *= STAGE_RATIO(MAT_TRANSPARENCY, s.whichobject);
### endif
            float dotp = dot(r, n);
            if (dotp > 0) { n = -n; dotp = -dotp; }

            float nfrac = 0;
            const float ior1 = m_ior;
            const float ior0 = m_ior_ambient;
            if (s.inside == s.whichobject) {         // leaving the object
                nfrac = ior1/ior0;
                s.inside = 0;
            }
            else {                                   // entering it
                nfrac = ior0/ior1;
                s.inside = s.whichobject;
            }

            if (nfrac != 1) {
                const float cos2t = 1 - nfrac*nfrac*(1 - dotp*dotp);
                if (cos2t < 0) {
                    // total internal reflection: undo the inside bookkeeping
                    r -= 2*dotp * n;
                    if (s.inside == s.whichobject) s.inside = 0;
                    else s.inside = s.whichobject;
                }
                else r = normalize(r*nfrac + n*(-dotp*nfrac - sqrtf(cos2t)));
            }

            if (dot(n, r) < 0) n = -n;
            const float blur = m_transparency_blur;
            if (blur > 0) {
                vec3 gauss_rvec = B.gauss_vector();
                if (dot(n, gauss_rvec) < 0) gauss_rvec = -gauss_rvec;
                r = normalize(gauss_rvec * blur + r * (1.0f - blur));
            }
### if renderer.bidirectional
            s.suppress_emission = 0;
### endif
            p = -1.0f;
            s.ray = r; s.ray_color = s.ray_color * pipeline_color;
        }
        else p -= cur_prob;
    }

    // ---- diffuse reflection or absorption (shader.cl:291-353)
    if (p >= 0.0f) {
### if renderer.bidirectional
        s.suppress_emission = 0;
### endif
        cur_col = m_diffuse;
        cur_prob = STAGE_PROB(MAT_DIFFUSE, s.whichobject);

        if (cur_prob > 0.0f && s.diffusions_left > 0) {
            pipeline_color = pipeline_color * cur_col;
            s.diffusions_left -= 1;
### if renderer.bidirectional
            if (B.light_id > 0) {
                // next-event estimation towards the light point of this segment
                if (clr_farther_than(s.pos - B.light_center, B.min_light_sampling_distance)) {
                    s.suppress_emission = B.light_id;
### if renderer.light_samples > 1
                    // mean over light_samples points of the light: the terms are summed first and
                    // reach the image as one addend (the pass-split log holds one per stage)
                    vec3 estimate = make_vec3(0.0f, 0.0f, 0.0f);
                    #pragma unroll 1
                    for (int light_i = 0; light_i < {{ renderer.light_samples }}; ++light_i) {
                        vec3 light_point = B.light_point, light_normal = B.light_normal;
                        if (light_i > 0) B.light_sample(light_i, light_point, light_normal);
### else
                    {
                        const vec3 light_point = B.light_point, light_normal = B.light_normal;
### endif
                        vec3 shadow_ray = light_point - s.pos;
                        if (dot(shadow_ray, n) > 0 && dot(shadow_ray, light_normal) < 0) {
                            // diffusions_left was at least 1 before this stage took one, and
                            // dot(n, shadow_ray) > 0 was just tested: the two early exits of the
                            // reference's visibility kernel do not apply.  The sweep normalises the
                            // vector to the light point; its length and unit vector are, bit for
                            // bit, the distance and direction that the estimate below needs
                            // (same expression, v / sqrt(dot(v, v))), so they are taken from there.
                            float shadow_mask, shadow_dist;
                            if (lazy_visibility) {
                                const clr_direction light_dir = clr_make_direction(shadow_ray, clr_g);
                                shadow_mask = clr_light_visibility_along(s.pos, s.whichobject, s.which_subobject, s.inside,
                                                                         light_dir, B.light_id, vector_data, integer_data, clr_g);
                                shadow_dist = light_dir.len;
                                shadow_ray = light_dir.unit;
                            }
                            else {
### if renderer.light_samples > 1
                                shadow_mask = light_i == 0 ? eager_shadow_mask :
                                    clr_light_visibility(s.pos, n, s.whichobject, s.which_subobject, s.inside, 1,
                                                         light_point, B.light_id, vector_data, integer_data, clr_g);
### else
                                shadow_mask = eager_shadow_mask;
### endif
                                shadow_dist = length(shadow_ray);
                                shadow_ray = fast_normalize(shadow_ray);
                            }
                            const float geom = dot(n, shadow_ray) / CLR_PI_F
                                * dot(-shadow_ray, light_normal) / (shadow_dist*shadow_dist);
### if renderer.light_samples > 1
                            estimate += color_to_rgb(geom * s.ray_color * pipeline_color, cmask)
                                * color_to_rgb(m_light_emission, make_vec3(1.0f, 1.0f, 1.0f))
                                * shadow_mask * B.light_area;
### else
                            img += color_to_rgb(geom * s.ray_color * pipeline_color, cmask)
                                * color_to_rgb(m_light_emission, make_vec3(1.0f, 1.0f, 1.0f))
                                * shadow_mask * B.light_area;
### endif
                        }
                    }
### if renderer.light_samples > 1
                    img += estimate * {{ renderer.light_weight_literal }};
### endif
                }
                else if (s.diffusions_left == 0) {
                    // too close to the light for a stable estimate: allow one more bounce
                    s.diffusions_left = 1;
                    s.suppress_emission = -B.light_id;
                }
            }
### endif
            r = B.rvec;
            if (dot(n, r) < 0) r = -r;
            pipeline_color = pipeline_color * (2.0f * dot(n, r));
        }
        else pipeline_color = pipeline_color * 0.0f;

        p = -1.0f;
        s.ray = r; s.ray_color = s.ray_color * pipeline_color;
    }

    // ---- Russian roulette (shader.cl:371-434 with a group of one ray)
    {
        const float cur_intensity = COLOR2PROB(s.ray_color);
        float survive = cur_intensity;
        float russian_prob = 1.0f;
### if renderer.scene.min_russian_prob
        if (cur_intensity > 0.0f) {
            russian_prob = fminf(fmaxf(cur_intensity, {{ renderer.scene.min_russian_prob|float }}f), 1.0f);
            if (B.russian > russian_prob) survive = 0.0f;
        }
### endif
        if (survive == 0.0f) return false;
### if renderer.scene.min_russian_prob
        if (russian_prob < 1.0f) s.ray_color = s.ray_color / russian_prob;
### endif
    }
    return true;
}
### if renderer.materials_in_lds
### for name in ['scattering', 'scattering_blur', 'absorption', 'emission', 'reflection', 'transparency', 'diffuse', 'reflection_blur', 'transparency_blur', 'ior', 'ior_ambient', 'light_emission']
#undef m_{{ name }}
### endfor
### endif
