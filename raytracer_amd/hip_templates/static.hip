{#- Scene-independent device helpers.  Counterpart of the reference's static.cl
    (/root/reference/cl_templates/static.cl:4-101): vector algebra, affine helpers,
    interval arithmetic and quaternion products.  Every arithmetic step is written out
    so that the CPU oracle (oracle/clray_oracle.c) can follow the same order of
    operations; the translation unit is compiled with -ffp-contract=off in parity mode. -#}
typedef unsigned int uint;

struct vec3 { float x, y, z; };
struct vec2 { float x, y; };
struct vec4 { float x, y, z, w; };

#define CLR_FN static __device__ __forceinline__
#define CLR_FN_MEMBER __device__ __forceinline__
// shape tests kept out of line (Tracer.outlined): one copy for both object sweeps
#define CLR_OUTLINED static __device__ __attribute__((noinline))

CLR_FN vec3 make_vec3(float x, float y, float z) { vec3 v; v.x = x; v.y = y; v.z = z; return v; }
CLR_FN vec3 make_vec3(float4 f) { return make_vec3(f.x, f.y, f.z); }
CLR_FN vec2 make_vec2(float x, float y) { vec2 v; v.x = x; v.y = y; return v; }
CLR_FN vec4 make_vec4(float x, float y, float z, float w) { vec4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
CLR_FN vec4 make_vec4(vec3 a, float w) { return make_vec4(a.x, a.y, a.z, w); }

CLR_FN vec3 operator+(vec3 a, vec3 b) { return make_vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
CLR_FN vec3 operator-(vec3 a, vec3 b) { return make_vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
CLR_FN vec3 operator*(vec3 a, vec3 b) { return make_vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
CLR_FN vec3 operator/(vec3 a, vec3 b) { return make_vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
CLR_FN vec3 operator-(vec3 a) { return make_vec3(-a.x, -a.y, -a.z); }
CLR_FN vec3 operator*(vec3 a, float s) { return make_vec3(a.x * s, a.y * s, a.z * s); }
CLR_FN vec3 operator*(float s, vec3 a) { return make_vec3(s * a.x, s * a.y, s * a.z); }
CLR_FN vec3 operator/(vec3 a, float s) { return make_vec3(a.x / s, a.y / s, a.z / s); }
CLR_FN vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
CLR_FN vec3& operator-=(vec3& a, vec3 b) { a = a - b; return a; }
CLR_FN vec3& operator*=(vec3& a, vec3 b) { a = a * b; return a; }
CLR_FN vec3& operator*=(vec3& a, float s) { a = a * s; return a; }

CLR_FN vec2 operator+(vec2 a, vec2 b) { return make_vec2(a.x + b.x, a.y + b.y); }
CLR_FN vec2 operator+(vec2 a, float s) { return make_vec2(a.x + s, a.y + s); }
CLR_FN vec2 operator-(vec2 a, float s) { return make_vec2(a.x - s, a.y - s); }
CLR_FN vec2 operator*(vec2 a, vec2 b) { return make_vec2(a.x * b.x, a.y * b.y); }
CLR_FN vec2 operator*(vec2 a, float s) { return make_vec2(a.x * s, a.y * s); }
CLR_FN vec2 operator*(float s, vec2 a) { return make_vec2(s * a.x, s * a.y); }

CLR_FN vec4 operator+(vec4 a, vec4 b) { return make_vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
CLR_FN vec4 operator*(float s, vec4 a) { return make_vec4(s * a.x, s * a.y, s * a.z, s * a.w); }

// dot / cross / length / normalize: sums are taken left to right
CLR_FN float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
CLR_FN float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
CLR_FN float dot(vec4 a, vec4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
CLR_FN vec3 cross(vec3 a, vec3 b) {
    return make_vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
CLR_FN float length(vec3 a) { return sqrtf(dot(a, a)); }
CLR_FN float length(vec4 a) { return sqrtf(dot(a, a)); }
CLR_FN vec3 normalize(vec3 a) { return a / length(a); }
#define fast_normalize normalize

// ---- exact division / square root with operand-range speculation (see clr_exact.inl).
// Generated shape functions and per-class steps are templates over a guard type and receive the
// guard as their last argument `clr_g`; code outside such a function finds this global, stateless
// guard under the same name and gets the generic arithmetic.
{% include 'clr_exact.inl' %}
static __device__ const clr_guard_safe clr_g = {};
#define native_sqrt(x) clr_sqrt((x), clr_g)

#define CLR_MATH_FN CLR_FN
// the one division inside clr_logf: operands inside the range of the short sequence by construction
#define CLR_LOG_QUOTIENT(a, d) clr_log_quotient((a), (d))
{% include 'clr_math.inl' %}
{% include 'clr_rng.inl' %}

// ---- interval arithmetic (static.cl:4-61); an interval is (x = begin, y = end)
typedef vec2 ia_type;
#define ia_new(b, e) make_vec2(b, e)
#define ia_new_exact(ex) make_vec2(ex, ex)
#define ia_begin(a) ((a).x)
#define ia_end(a) ((a).y)
#define ia_len(a) (ia_end(a) - ia_begin(a))
#define ia_contains_zero(a) (ia_begin(a) <= 0.0f && ia_end(a) >= 0.0f)
#define ia_center(a) ((ia_begin(a) + ia_end(a)) * 0.5f)
#define ia_sub(a, b) ia_new(ia_begin(a) - ia_end(b), ia_end(a) - ia_begin(b))
#define ia_neg(a) ia_new(-ia_end(a), -ia_begin(a))
#define ia_mul_pos_exact(a, ex) ((a) * (float)(ex))
#define ia_mul_neg_exact(a, ex) (ia_neg(a) * -(float)(ex))

// |a|: the three cases of the reference (static.cl:13-18) without branches, same numbers
CLR_FN ia_type ia_abs(ia_type a) {
    return ia_new(fmaxf(fmaxf(ia_begin(a), -ia_end(a)), 0.0f), fmaxf(-ia_begin(a), ia_end(a)));
}

// Interval product.  The reference (static.cl:20-47) picks the two bounding products by a case
// analysis on the signs of the end points; the lanes of a wave take different cases and the wave
// then runs all of them.  Rounding is monotone, so the products the case analysis picks are the
// smallest and the largest of the four end-point products: the branch-free form returns the same
// numbers (at most the sign of a zero bound can differ, which no comparison with zero and no
// non-zero sum or product downstream can see).  The oracle keeps the case analysis.
CLR_FN ia_type ia_mul(ia_type a, ia_type b) {
    const float p0 = ia_begin(a) * ia_begin(b), p1 = ia_begin(a) * ia_end(b),
                p2 = ia_end(a) * ia_begin(b), p3 = ia_end(a) * ia_end(b);
    return ia_new(fminf(fminf(p0, p1), fminf(p2, p3)), fmaxf(fmaxf(p0, p1), fmaxf(p2, p3)));
}

CLR_FN ia_type ia_pow2(ia_type a) { ia_type ab = ia_abs(a); return ab * ab; }
CLR_FN ia_type ia_pow3(ia_type a) { return a * a * a; }
CLR_FN ia_type ia_pow4(ia_type a) { ia_type a2 = ia_pow2(a); return a2 * a2; }

// ---- affine helpers (static.cl:66-82): rows of the matrix in m[0..2].xyz, translation m[3].xyz
CLR_FN vec3 apply_linear_transform(const float4* m, vec3 v) {
    return make_vec3(dot(make_vec3(m[0]), v), dot(make_vec3(m[1]), v), dot(make_vec3(m[2]), v));
}
CLR_FN vec3 apply_transposed_linear_transform(const float4* m, vec3 v) {
    return make_vec3(dot(make_vec3(m[0].x, m[1].x, m[2].x), v),
                     dot(make_vec3(m[0].y, m[1].y, m[2].y), v),
                     dot(make_vec3(m[0].z, m[1].z, m[2].z), v));
}
CLR_FN vec3 apply_affine_transform(const float4* m, vec3 v) {
    return apply_linear_transform(m, v) + make_vec3(m[3]);
}

// HalfSpace.can_hit_expression, second part: may the plane x = 0 be reached before `limit` (the
// closest hit so far / the light point)?  The reference's distance is -o.x / (l.x / len) / len with
// one and the same rounded len: -o.x / l.x times three roundings, within 3.1 * 2^-24 of it, and it
// counts only if it is below `limit`.  With p = limit |l.x| (one rounding), |o.x| >= p (1 + 2^-20)
// (one more) puts the exact quotient at least 14 * 2^-24 above `limit`: the reference's value is not
// below it.  Where p or l.x is so small that a rounding may be a denormal's, the full test decides.
CLR_FN bool clr_plane_before(const float o_x, const float l_x, const float limit) {
    const float p = limit * fabsf(l_x);
    // both p and |l.x| above 2^-100: one unsigned comparison of the smaller bit pattern (a NaN's
    // pattern is large and fails the comparison that follows)
    const uint smaller = min(__float_as_uint(p), __float_as_uint(fabsf(l_x)));
    const bool certainly_beyond = (smaller > 0x0D800000u) & (fabsf(o_x) >= p * 1.00000095f);
    return !certainly_beyond;
}

// Conservative miss test for the unit sphere from the un-normalised ray in tracer coordinates
// (Sphere.can_hit_expression): false only where the reference's test - normalise the ray, then
// half_b = -u.o <= 0 or disc = half_b^2 - |o|^2 + 1 < 0 - ends without a hit whatever its
// roundings (eps = 2^-24; the reference's u.o is within 7 eps |o| of d.o / |d| and its disc within
// 20 eps (|o|^2 + 1) of the exact one; the quantities below carry errors of 3.1 eps |d||o| and
// 16 eps |d|^2 (|o|^2 + 1); the thresholds leave a factor of two: DESIGN section 3.1).
// Origins on or inside the sphere, NaNs and infinities take the full test.
CLR_FN bool clr_unit_sphere_may_be_hit(const vec3 o, const vec3 d) {
    const float od = dot(o, d), oo = dot(o, o), dd = dot(d, d);
    if (!(oo > 1.0f)) return true;
    const bool heading_away = od > 0.0f && od * od > 0x1p-38f * (dd * oo);
    const bool passing_by = od * od - dd * (oo - 1.0f) < -0x1p-18f * (dd * (oo + 1.0f));
    return !(heading_away || passing_by);
}

// Bodies bounded by planes (objects/convex.py: plane_body_cull_ball).  `o` = the local origin
// relative to the centre of the ball, d = the un-normalised local ray, rho2 = the squared radius of
// the culling ball (1.25 x the radius of a ball that holds the body, and 5 % for the roundings
// here), limit = the distance up to which a hit counts.  False only where the reference's component
// tests are known to end without a hit whatever their roundings: the origin is within 64 radii,
// the stretch looked at is shorter than 256 radii, and the ray stays outside the ball on it -
// then every point of the stretch violates one of the body's planes by at least a quarter of the
// centre's distance from it (>= 1/80 of the body's radius), which is 80 x what the roundings of
// slope, offset and quotient can move a computed crossing.  NaNs and
// infinities take the full test.
CLR_FN bool clr_ball_may_be_hit(const vec3 o, const vec3 d, const float rho2, const float limit) {
    const float od = dot(o, d), oo = dot(o, o), dd = dot(d, d);
    if (!(oo > rho2 && oo < 0x1p+12f * rho2 && dd > 0x1p-60f && dd < 0x1p+60f)) return true;
    if (!(limit * limit * dd < 0x1p+16f * rho2)) return true;
    const bool heading_away = od > 0.0f && od * od > 0x1p-36f * (dd * oo);
    const bool passing_by = oo * dd - od * od > rho2 * dd;
    return !(heading_away || passing_by);
}

// ---- quaternions (static.cl:87-101): x = real part
CLR_FN vec4 quaternion_mult(vec4 q1, vec4 q2) {
    vec3 v1 = make_vec3(q1.y, q1.z, q1.w), v2 = make_vec3(q2.y, q2.z, q2.w);
    vec3 im = q1.x * v2 + q2.x * v1 + cross(v1, v2);
    return make_vec4(q1.x * q2.x - dot(v1, v2), im.x, im.y, im.z);
}
CLR_FN vec4 quaternion_square(vec4 q) {
    vec3 v = make_vec3(q.y, q.z, q.w);
    float two_r = 2 * q.x;
    return make_vec4(q.x * q.x - dot(v, v), two_r * q.y, two_r * q.z, two_r * q.w);
}
