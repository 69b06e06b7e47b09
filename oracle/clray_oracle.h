/* clray_oracle.h - CPU restatement of the clray per-pixel path-tracing loop.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle for the HIP back end: a plain-C,
 * data-driven restatement of the reference's algorithm (one generic interpreter instead of
 * scene-specialised generated code).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product (raytracer_amd/) never does.
 *
 * Pinning: checked against golden vectors produced in the development container by the
 * reference's own OpenCL device code compiled for x86-64 (tools/make_golden.py,
 * tests/golden/), see DESIGN.md.
 */
#ifndef CLRAY_ORACLE_H
#define CLRAY_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* object kinds */
enum { ORC_SPHERE = 1, ORC_HALFSPACE = 2, ORC_CONVEX = 3, ORC_TRIMESH = 4, ORC_OCTREE = 5,
       ORC_QJS2 = 6, ORC_IMPLICIT = 7, ORC_DISTFIELD = 8, ORC_QJS_IA = 9 };
/* convex component kinds */
enum { ORC_C_SPHERE = 1, ORC_C_HALFSPACE = 2, ORC_C_LAYER = 3, ORC_C_ZLAYER = 4,
       ORC_C_CYLINDER = 5, ORC_C_ZCYLINDER = 6, ORC_C_CONE = 7, ORC_C_ZCONE = 8 };

typedef struct orc_component {
    int kind;
    int n_subobjects;
    float pos[3];      /* component position inside the intersection */
    float v[3];        /* normal or axis */
    float s;           /* h, R or slope */
} orc_component;

/* hooks for shapes whose arithmetic is generated from user input (implicit surfaces,
 * distance fields): compiled separately and attached by address */
typedef void (*orc_tracer_fn)(const float origin[3], const float ray[3], float old_isec_dist,
                              float *p_new_isec_dist, unsigned *p_subobject, int inside,
                              int origin_self);
typedef void (*orc_normal_fn)(const float pos[3], unsigned subobject, float normal[3]);

typedef struct orc_object {
    int kind;
    int convex;
    int auto_flip_normal;
    float inv[12];            /* rows 0..2 of the inverse linear map, then its translation */
    int comp_first, comp_count;
    int vec_offset, int_offset;   /* into vector_data (float4 units) and integer_data */
    int ip[8];                /* mesh: n_vertices, n_triangles, [octree: root_data_offset, face_data_length], smooth flag in ip[7] */
    float fp[8];              /* octree: root_origin xyz, root_size; qjs2: c[4], ... */
    orc_tracer_fn tracer;
    orc_normal_fn normal;
} orc_object;

typedef struct orc_scene {
    int n_objects;
    const orc_object *objects;          /* object id = index + 1 */
    const orc_component *components;
    const float *vector_data;           /* float4 rows */
    const int *integer_data;
    int rgb;                            /* 1: RGB shading, 0: spectral */
    const float *material_colors;       /* rgb: 5*(n+1) float4 rows */
    const float *material_scalars;      /* rgb: 5*(n+1) floats */
    int max_bounces, min_bounces;
    int root_object_id;
    int bidirectional;
    float min_russian_prob;             /* 0 = roulette off */
    float max_ray_length;
    int light_samples;                  /* points of the segment's light per next-event estimate; 0 = 1 */
} orc_scene;

/* per-pixel random numbers (hip_templates/clr_rng.inl) instead of the bounce tape */
typedef struct orc_rng {
    unsigned seed_lo, seed_hi;
    int first_pass;                     /* global index of the first pass of this call */
    int first_ray;                      /* global ray index of pixel 0 (tile partitioning) */
    int n_lights;
    const float *lights;                /* per light 8 floats: centre xyz, R, area, min distance, id (int bits), 0 */
    int pass_stride;                    /* global index of local pass j = first_pass + j * pass_stride; 0 = 1 */
    const int *pixel_ids;               /* optional: global ray index of pixel k (a subset of the frame); NULL = first_ray + k */
} orc_rng;

#define ORC_PASS_RECORD 6
#define ORC_BOUNCE_RECORD 5             /* + 2 per further light point (light_samples > 1) */
#define ORC_DEBUG_RECORD 6

/* Adds n_passes sample passes to img (n_pixels x float4, running sums).  Tapes as in
 * include/clray_hip.h.  dbg: optional max_bounces x n_pixels x 6 float4 (n_passes == 1).
 * flops: optional, receives the number of algorithmic FP32 operations executed
 * (counting rule: DESIGN.md).  Returns 0. */
/* rng == NULL: the random numbers of every segment come from bounce_tape (shared by all pixels);
 * else bounce_tape is ignored and every pixel draws its own */
int orc_render(const orc_scene *scene, const float *cam, const float *pass_tape,
               const float *bounce_tape, const float *material_rows, int n_pixels, int n_passes,
               float *img, float *dbg, int n_threads, double *flops, const orc_rng *rng);

/* The same passes in the reference's execution structure (renderer.py:293-433): ray state in arrays,
 * one sweep over the live rays per object and per step with a barrier behind each, the culler over
 * groups of warp_size consecutive live rays (shader.cl:371-434) and a stable compaction per bounce.
 * warp_size 1: the sums of orc_render bit for bit.  Tape mode only.  launches: number of sweeps. */
int orc_render_wavefront(const orc_scene *scene, const float *cam, const float *pass_tape,
                         const float *bounce_tape, const float *material_rows, int n_pixels, int n_passes,
                         float *img, int warp_size, int n_threads, double *launches);

int orc_first_hit(const orc_scene *scene, const float *cam, const float *pass_tape, int n_pixels,
                  int *out_ids, float *out_t, float *out_normal /* n x 4 */, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
