// Correctly rounded binary32 division and square root in fewer instructions than the
// compiler's generic expansions, by speculation.  Device code only.  The values returned are
// the IEEE-754 round-to-nearest results, bit for bit what `a / b` and `sqrtf(x)` give (and
// what the CPU oracle computes); tests/test_gpu_exact_math.py checks that on 2^30 operand
// tuples per exponent range.
//
// Division.  The generic expansion is v_div_scale x2, v_rcp, a Newton step on the reciprocal,
// quotient + two residual corrections, v_div_fmas, v_div_fixup.  The two scale steps and the
// fix-up are the identity when
//     both operands are finite and non-zero,  |a| >= 2^-103,  |d| normal and < 2^126,
//     exponent(a) - exponent(d) < 96  and the quotient is not denormal,
// all of which hold for 2^-47 <= |a|, |d| <= 2^47.  What remains is the FMA chain below: its
// first half depends on the divisor only and is shared by all quotients with that divisor
// (ray / |ray| and t / |ray| of every object test).  Scalar instructions throughout: packed FP32
// (v_pk_fma_f32) was measured slower here, written by hand or made by the SLP vectoriser (the
// register pairs cost moves and registers; programs are built with -fno-slp-vectorize).
// Square root.  v_sqrt_f32 is within 1 ulp; its two neighbours are tested with one FMA residual
// each, as in the generic expansion, whose input scaling, un-scaling and class test matter only
// below 2^-96.
//
// Speculation.  The short sequences carry no range tests.  Instead the operands are folded into
// two running bounds (clr_guard_spec) and the caller checks the bounds once per path segment: if
// they left the range the lane starts the path of that pass again and walks it with the generic
// arithmetic (clr_guard_safe; main.hip).  Generated functions are templates over the guard type, so both
// versions come from one source text.  The bounds are kept on the bit patterns with integer
// min / max (the operands tracked are non-negative: squares, lengths, distances, discriminants;
// a negative one or a NaN lands outside the range):
//     lo = smallest of  x^2, y^2, z^2 of every direction,  every t,  every sqrt argument
//     hi = largest of   every length, every t, every square root
// and the range is lo >= 2^-94, hi <= 2^30.  Then components are >= 2^-47 and lengths are in
// [2^-47, 2^30], so quotients component / length and t / length are normal numbers with an
// exponent difference < 96 and numerators >= 2^-103, and sqrt arguments are in [2^-94, 2^60].
#ifndef CLR_GUARD_LO
#define CLR_GUARD_LO 0x1p-94f
#endif
#ifndef CLR_GUARD_HI
#define CLR_GUARD_HI 0x1p+30f
#endif

struct clr_guard_safe {};                // generic arithmetic, nothing to track
struct clr_guard_spec { int lo, hi; };   // short sequences; bit patterns of the bounds

#define CLR_BITS(v) __float_as_int(v)
#define CLR_IMIN(a, b) ((a) < (b) ? (a) : (b))
#define CLR_IMAX(a, b) ((a) > (b) ? (a) : (b))

CLR_FN void clr_guard_reset(const clr_guard_safe &) {}
CLR_FN bool clr_guard_ok(const clr_guard_safe &) { return true; }
CLR_FN void clr_guard_reset(clr_guard_spec &g) { g.lo = CLR_BITS(1.0f); g.hi = CLR_BITS(1.0f); }
CLR_FN bool clr_guard_ok(const clr_guard_spec &g) {
    return g.lo >= CLR_BITS(CLR_GUARD_LO) && g.hi <= CLR_BITS(CLR_GUARD_HI);
}

// ---- the short sequences (valid inside the range only)
CLR_FN float clr_sqrt_short(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float down = __int_as_float(__float_as_int(s) - 1);
    const float up = __int_as_float(__float_as_int(s) + 1);
    const float r_down = __builtin_fmaf(-down, s, x);
    const float r_up = __builtin_fmaf(-up, s, x);
    s = r_down <= 0.0f ? down : s;
    s = r_up > 0.0f ? up : s;
    return s;
}

CLR_FN float clr_reciprocal_short(float d) {
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}

CLR_FN float clr_quotient_short(float a, float d, float y) {
    float q = a * y;
    float r = __builtin_fmaf(-d, q, a);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-d, q, a);
    return __builtin_fmaf(r, y, q);
}

// The one division inside clr_logf (clr_math.inl), f / (2 + f) with f = m - 1, m a binary32 number in
// [sqrt(1/2), sqrt(2)): f is 0 or at least 2^-24 in magnitude and the divisor lies in [1.7, 2.42], so
// the operands are inside the range of the short sequence by construction (all 2^23 values of m are
// checked against the compiler's division in tests/test_gpu_exact_math.py).
CLR_FN float clr_log_quotient(const float a, const float d) { return clr_quotient_short(a, d, clr_reciprocal_short(d)); }

// Square root where no guard is at hand (marching loops): the short sequence inside its range,
// tested on the spot, the generic expansion outside it (and for NaNs).  Same bits either way.
CLR_FN float clr_sqrt_ranged(float x) {
    if (x >= 0x1p-94f && x <= 0x1p+60f) return clr_sqrt_short(x);
    return sqrtf(x);
}

// ---- square root
CLR_FN float clr_sqrt(float x, const clr_guard_safe &) { return sqrtf(x); }
CLR_FN float clr_sqrt(float x, clr_guard_spec &g) {
    const float s = clr_sqrt_short(x);
    g.lo = CLR_IMIN(g.lo, CLR_BITS(x));
    g.hi = CLR_IMAX(g.hi, CLR_BITS(s));
    return s;
}

// ---- a vector, its length and the quotients by that length
struct clr_direction {
    vec3 unit;     // v / |v|
    float len;     // |v|
    float y;       // reciprocal of len after the Newton step (speculative version only)
};

CLR_FN clr_direction clr_make_direction(vec3 v, const clr_guard_safe &) {
    clr_direction r;
    r.len = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    r.unit = make_vec3(v.x / r.len, v.y / r.len, v.z / r.len);
    r.y = 0.0f;
    return r;
}

CLR_FN clr_direction clr_make_direction(vec3 v, clr_guard_spec &g) {
    clr_direction r;
    const float xx = v.x * v.x, yy = v.y * v.y, zz = v.z * v.z;
    r.len = clr_sqrt_short(xx + yy + zz);
    r.y = clr_reciprocal_short(r.len);
    r.unit = make_vec3(clr_quotient_short(v.x, r.len, r.y), clr_quotient_short(v.y, r.len, r.y),
                       clr_quotient_short(v.z, r.len, r.y));
    g.lo = CLR_IMIN(CLR_IMIN(g.lo, CLR_BITS(xx)), CLR_BITS(yy));
    g.lo = CLR_IMIN(g.lo, CLR_BITS(zz));
    return r;
}

// ---- products with a direction vector some of whose components are known to be zero
// (`zero` = clr_zero_mask of that vector: a constant after inlining where the parameters are baked
// into the program, a wave-uniform value otherwise).  For finite v a
// product with a zero is a zero and adding it changes nothing unless the whole sum is a zero
// (whose sign may then differ, which a later sum with a non-zero term, a square or a comparison
// cannot see).  The speculative version leaves those products out; the generic one evaluates
// every term, in the reference's order (x + y) + z.
CLR_FN int clr_zero_mask(const vec3 n) {
    return (n.x == 0.0f ? 1 : 0) | (n.y == 0.0f ? 2 : 0) | (n.z == 0.0f ? 4 : 0);
}
CLR_FN float clr_dot_lit(const vec3 v, const vec3 n, const int, const clr_guard_safe &) {
    return v.x * n.x + v.y * n.y + v.z * n.z;
}
CLR_FN float clr_dot_lit(const vec3 v, const vec3 n, const int zero, clr_guard_spec &) {
    float r = 0.0f;
    bool any = false;
    if (!(zero & 1)) { r = v.x * n.x; any = true; }
    if (!(zero & 2)) { r = any ? r + v.y * n.y : v.y * n.y; any = true; }
    if (!(zero & 4)) { r = any ? r + v.z * n.z : v.z * n.z; }
    return r;
}
// v - s * n (the part of v at right angles to a unit axis n, s = v . n)
CLR_FN vec3 clr_reject_lit(const vec3 v, const float s, const vec3 n, const int, const clr_guard_safe &) {
    return make_vec3(v.x - s * n.x, v.y - s * n.y, v.z - s * n.z);
}
CLR_FN vec3 clr_reject_lit(const vec3 v, const float s, const vec3 n, const int zero, clr_guard_spec &) {
    return make_vec3((zero & 1) ? v.x : v.x - s * n.x, (zero & 2) ? v.y : v.y - s * n.y,
                     (zero & 4) ? v.z : v.z - s * n.z);
}

// An object's local ray direction: normalised here, or - bit 12 of the object's skip mask: its
// local ray IS the world ray (identity linear part) - the normalisation of the world ray that
// the sweep made once for all such objects.  The same computation on the same bits either way.
CLR_FN clr_direction clr_direction_of(vec3 v, const clr_direction &, const int, const clr_guard_safe &g) {
    return clr_make_direction(v, g);
}
CLR_FN clr_direction clr_direction_of(vec3 v, const clr_direction &world, const int skip, clr_guard_spec &g) {
    if (skip & (1 << 12)) return world;
    return clr_make_direction(v, g);
}

// upper bound of a direction's length and of a distance along it (t <= 0: no hit, not divided)
CLR_FN void clr_guard_upper(const clr_direction &, float, const clr_guard_safe &) {}
CLR_FN void clr_guard_upper(const clr_direction &dir, float t, clr_guard_spec &g) {
    g.hi = CLR_IMAX(CLR_IMAX(g.hi, CLR_BITS(dir.len)), CLR_BITS(t));
}

// t / dir.len for t > 0, after clr_guard_upper(dir, t, g)
CLR_FN float clr_over_length(float t, const clr_direction &dir, const clr_guard_safe &) { return t / dir.len; }
CLR_FN float clr_over_length(float t, const clr_direction &dir, clr_guard_spec &g) {
    g.lo = CLR_IMIN(g.lo, CLR_BITS(t));
    return clr_quotient_short(t, dir.len, dir.y);
}

// ---- affine transform with statically known zero coefficients
// m[0..2].xyz = rows of the matrix, m[3].xyz = translation.  `skip` (a literal at the call site,
// Renderer.affine_skip_mask) has bit 3*row+col set where the baked coefficient is zero and bit
// 9+row set where the translation is non-zero.  For finite v a term with a zero coefficient is a
// zero, so leaving it out changes nothing unless the whole sum is a zero (whose sign may then
// differ).  The speculative version therefore shortens
//   * directions: a zero component fails the guard (its square is tracked);
//   * position rows with a non-zero translation: a zero sum is absorbed by the translation.
// The generic version evaluates every term.
CLR_FN float clr_row_dot(const float4 m, const vec3 v, const int skip) {
    float r = 0.0f;
    bool any = false;
    if (!(skip & 1)) { r = m.x * v.x; any = true; }
    if (!(skip & 2)) { r = any ? r + m.y * v.y : m.y * v.y; any = true; }
    if (!(skip & 4)) { r = any ? r + m.z * v.z : m.z * v.z; }
    return r;
}

CLR_FN vec3 clr_to_local_direction(const float4 *m, const vec3 v, const int, const clr_guard_safe &) {
    return make_vec3(m[0].x * v.x + m[0].y * v.y + m[0].z * v.z,
                     m[1].x * v.x + m[1].y * v.y + m[1].z * v.z,
                     m[2].x * v.x + m[2].y * v.y + m[2].z * v.z);
}
CLR_FN vec3 clr_to_local_direction(const float4 *m, const vec3 v, const int skip, clr_guard_spec &) {
    return make_vec3(clr_row_dot(m[0], v, skip & 7), clr_row_dot(m[1], v, (skip >> 3) & 7),
                     clr_row_dot(m[2], v, (skip >> 6) & 7));
}

CLR_FN vec3 clr_to_local_position(const float4 *m, const vec3 v, const int skip, const clr_guard_safe &g) {
    const vec3 r = clr_to_local_direction(m, v, skip, g);
    return make_vec3(r.x + m[3].x, r.y + m[3].y, r.z + m[3].z);
}
CLR_FN vec3 clr_to_local_position(const float4 *m, const vec3 v, const int skip, clr_guard_spec &) {
    return make_vec3(clr_row_dot(m[0], v, (skip & (1 << 9)) ? skip & 7 : 0) + m[3].x,
                     clr_row_dot(m[1], v, (skip & (1 << 10)) ? (skip >> 3) & 7 : 0) + m[3].y,
                     clr_row_dot(m[2], v, (skip & (1 << 11)) ? (skip >> 6) & 7 : 0) + m[3].z);
}
