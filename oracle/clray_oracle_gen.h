/* clray_oracle_gen.h - interval arithmetic and the two ray-marching loops of the CPU oracle
 * (TEST INFRASTRUCTURE ONLY, see clray_oracle.h).  Included by clray_oracle.c (quaternion
 * Julia set) and by the small C files that oracle.py generates for shapes whose function is
 * user input (ImplicitSurface equations, DistanceField code strings).
 *
 * Intervals are GCC 2-float vectors so that the sympy-printed expressions
 * ("ia_pow4(x) + ia_mul_neg_exact(ia_pow2(x),-5) + 11.8f") compile unchanged:
 * +, * act per lane and broadcast scalars, as float2 does in OpenCL C.
 */
#ifndef CLRAY_ORACLE_GEN_H
#define CLRAY_ORACLE_GEN_H

#include <math.h>

typedef float ia_type __attribute__((vector_size(8)));

/* static.cl:4-24 */
static inline ia_type ia_new(float b, float e) { ia_type r = {b, e}; return r; }
#define ia_new_exact(ex) ia_new(ex, ex)
#define ia_begin(a) ((a)[0])
#define ia_end(a) ((a)[1])
#define ia_len(a) (ia_end(a) - ia_begin(a))
#define ia_contains_zero(a) (ia_begin(a) <= 0.0f && ia_end(a) >= 0.0f)
#define ia_center(a) ((ia_begin(a) + ia_end(a)) * 0.5f)
static inline ia_type ia_sub(ia_type a, ia_type b) { return ia_new(ia_begin(a) - ia_end(b), ia_end(a) - ia_begin(b)); }
static inline ia_type ia_neg(ia_type a) { return ia_new(-ia_end(a), -ia_begin(a)); }
#define ia_mul_pos_exact(a, ex) ((a) * (float)(ex))
#define ia_mul_neg_exact(a, ex) (ia_neg(a) * -(float)(ex))

static inline ia_type ia_abs(ia_type a) {
    if (ia_begin(a) >= 0.0f) return a;
    if (ia_end(a) <= 0.0f) return ia_neg(a);
    return ia_new(0.0f, fmaxf(-ia_begin(a), ia_end(a)));
}

/* static.cl:26-44 */
static inline ia_type ia_mul(ia_type a, ia_type b) {
    if (ia_contains_zero(a)) {
        if (ia_contains_zero(b))
            return ia_new(fminf(ia_begin(a) * ia_end(b), ia_begin(b) * ia_end(a)),
                          fmaxf(ia_begin(a) * ia_begin(b), ia_end(a) * ia_end(b)));
        ia_type tmp = b; b = a; a = tmp;
    }
    if (ia_end(a) < 0.0f) { a = ia_neg(a); b = ia_neg(b); }
    if (ia_end(b) < 0.0f) return ia_new(ia_end(a) * ia_begin(b), ia_begin(a) * ia_end(b));
    else if (ia_contains_zero(b)) return ia_new(ia_end(a) * ia_begin(b), ia_end(a) * ia_end(b));
    else return a * b;
}

/* static.cl:46-61 */
static inline ia_type ia_pow2(ia_type a) { ia_type ab = ia_abs(a); return ab * ab; }
static inline ia_type ia_pow3(ia_type a) { return a * a * a; }
static inline ia_type ia_pow4(ia_type a) { ia_type a2 = ia_pow2(a); return a2 * a2; }

/* objects/bounding_volumes.cl:3-27; 0 = the ray misses the bounding sphere */
static inline int orc_bounding_sphere(const float o[3], const float r[3], float R2, float old_isec_dist,
                                      float *tmin, float *tmax) {
    float dotp = -(r[0] * o[0] + r[1] * o[1] + r[2] * o[2]);
    float psq = o[0] * o[0] + o[1] * o[1] + o[2] * o[2];
    int inside_bnd = psq < R2;
    if (dotp <= 0 && !inside_bnd) return 0;
    const float discr = dotp * dotp - psq + R2;
    if (discr < 0) return 0;
    const float sq = sqrtf(discr);
    *tmin = fmaxf(dotp - sq, 0.0f);
    *tmax = fminf(dotp + sq, old_isec_dist);
    if (*tmax <= *tmin) return 0;
    return 1;
}

typedef struct orc_ia_rayfn {
    /* f over the box (x, y, z); df/dt over the parameter interval; ctx = shape data */
    ia_type (*f)(ia_type x, ia_type y, ia_type z, const void *ctx);
    ia_type (*df)(const ia_type ray[3], const ia_type origin[3], ia_type t, const void *ctx);
    const void *ctx;
} orc_ia_rayfn;

/* objects/ImplicitSurface.cl:6-111 */
static inline void orc_implicit_tracer(const orc_ia_rayfn *fn, const float origin[3], const float ray[3],
        float old_isec_dist, float *p_new_isec_dist, int inside, int origin_self,
        int no_self_intersection, int has_bound, float bound_r2, int MAX_ITER, float TARGET_EPS) {
    if (no_self_intersection && origin_self) return;
    ia_type t;
    if (has_bound) {
        float lo, hi;
        if (!orc_bounding_sphere(origin, ray, bound_r2, old_isec_dist, &lo, &hi)) return;
        t = ia_new(lo, hi);
    } else t = ia_new(0.0f, old_isec_dist);
    const float t_max = ia_end(t);
    const float FRACTION = 0.5f, SELF_MAX_BEGIN_STEP = 0.1f;
    float step;
    int steps_since_subdiv = 0;
    const ia_type r3[3] = {ia_new_exact(ray[0]), ia_new_exact(ray[1]), ia_new_exact(ray[2])};
    const ia_type o3[3] = {ia_new_exact(origin[0]), ia_new_exact(origin[1]), ia_new_exact(origin[2])};
    if (origin_self) ia_end(t) = fminf(ia_end(t), ia_begin(t) + SELF_MAX_BEGIN_STEP);
    for (int i = 0; i < MAX_ITER; i++) {
        if (ia_begin(t) >= t_max) return;
        if (ia_end(t) > t_max) ia_end(t) = t_max;
        const ia_type x = ia_mul(r3[0], t) + o3[0];
        const ia_type y = ia_mul(r3[1], t) + o3[1];
        const ia_type z = ia_mul(r3[2], t) + o3[2];
        const ia_type f = fn->f(x, y, z, fn->ctx);
        step = ia_len(t);
        int need_subdiv = 0;
        if ((inside && ia_end(f) > 0) || (!inside && ia_begin(f) < 0)) {
            need_subdiv = 1;
            if (origin_self) {
                const ia_type df = fn->df(r3, o3, t, fn->ctx);
                if (!ia_contains_zero(df)) {
                    if ((ia_end(df) < 0) == (inside != 0)) need_subdiv = 0;
                }
            }
            if (need_subdiv) {
                if (step < TARGET_EPS || i == MAX_ITER - 1) { *p_new_isec_dist = ia_center(t); return; }
                step *= FRACTION;
                ia_end(t) = ia_begin(t) + step;
                steps_since_subdiv = 0;
                continue;
            }
        }
        steps_since_subdiv++;
        if (steps_since_subdiv > 1) step /= FRACTION;
        t = ia_new(ia_end(t), ia_end(t) + step);
    }
}

/* objects/DistanceField.cl:6-66; dist_fn evaluates the user's distance code at (x, y, z) */
static inline void orc_distance_field_tracer(float (*dist_fn)(float, float, float), const float origin[3],
        const float ray[3], float old_isec_dist, float *p_new_isec_dist, int inside, int origin_self,
        int no_self_intersection, int has_bound, float bound_r2, int MAX_ITER, float EPS) {
    if (no_self_intersection && origin_self) return;
    float t, t_end;
    float sign = 1.0f;
    if (inside) sign = -1.0f;
    if (has_bound) { if (!orc_bounding_sphere(origin, ray, bound_r2, old_isec_dist, &t, &t_end)) return; }
    else { t = 0.0f; t_end = old_isec_dist; }
    float last_dist = 0.0f, derivative = -1.0f;
    for (int i = 0; i < MAX_ITER; i++) {
        if (t > t_end) return;
        const float x = origin[0] + t * ray[0], y = origin[1] + t * ray[1], z = origin[2] + t * ray[2];
        float dist = dist_fn(x, y, z);
        dist *= sign;
        if (dist < EPS) {
            if (origin_self) {
                if (i > 0) derivative = dist - last_dist;
                else derivative = 0.0f;
            }
            if (derivative < 0.0f) {
                if (t < t_end) *p_new_isec_dist = t;
                return;
            }
        }
        t += fmaxf(dist, EPS);
        last_dist = dist;
    }
}

static inline void orc_normalize(float v[3]) {
    const float len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] = v[0] / len; v[1] = v[1] / len; v[2] = v[2] / len;
}

#endif
