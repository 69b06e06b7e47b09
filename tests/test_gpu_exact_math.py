"""hip_templates/clr_exact.inl against the compiler's IEEE division and square root on the GPU:
the shortened sequences must give the same bits for every operand, inside and outside their
fast range.  2^30 operand pairs per mode, drawn on the device."""
import os

import numpy as np
import pytest

from raytracer_amd.accelerator import Accelerator

pytestmark = pytest.mark.gpu

TEMPLATES = os.path.join(os.path.dirname(__file__), '..', 'raytracer_amd', 'hip_templates')

KERNEL = r'''
typedef unsigned int uint;
struct vec3 { float x, y, z; };
#define CLR_FN static __device__ __forceinline__
CLR_FN vec3 make_vec3(float x, float y, float z) { vec3 v; v.x = x; v.y = y; v.z = z; return v; }
%(exact)s

CLR_FN uint next_bits(uint &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s * 0x9E3779B1u; }

// mode 0: raw bit patterns (signaling NaNs made quiet: arithmetic never produces them);
// mode 1: exponents within +-span of 2^0, random mantissa and sign
CLR_FN float draw(uint &s, int mode, int span) {
    uint bits = next_bits(s);
    if (mode == 0) {
        if ((bits & 0x7f800000u) == 0x7f800000u && (bits & 0x007fffffu)) bits |= 0x00400000u;
        return __uint_as_float(bits);
    }
    uint e = 127 - span + next_bits(s) %% (2*span + 1);
    return __uint_as_float((bits & 0x807fffffu) | (e << 23));
}

CLR_FN bool same(float a, float b) {
    return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
}

CLR_FN void report(uint *out, int slot, float a, float b, float got, float want) {
    if (atomicAdd(out + slot, 1u) == 0) {
        out[8 + 4*slot + 0] = __float_as_uint(a);   out[8 + 4*slot + 1] = __float_as_uint(b);
        out[8 + 4*slot + 2] = __float_as_uint(got); out[8 + 4*slot + 3] = __float_as_uint(want);
    }
}

// Every speculative result whose guard stayed inside the range must have the bits of the generic
// operation; out[3..6] count how many results of each kind the guard accepted.
extern "C" __global__ void check_exact(uint *out, int iterations, int mode, int span, uint seed) {
    uint s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed;
    if (s == 0) s = 1;
    uint accepted[3] = {0, 0, 0};
    for (int i = 0; i < iterations; i++) {
        const float a = draw(s, mode, span), b = draw(s, mode, span), c = draw(s, mode, span),
                    d = draw(s, mode, span);
        clr_guard_spec g;
        clr_guard_reset(g);
        const vec3 v = make_vec3(a, b, c);
        const clr_direction dir = clr_make_direction(v, g);
        clr_guard_upper(dir, 0.0f, g);
        if (clr_guard_ok(g)) accepted[0]++;          // direction alone
        clr_guard_upper(dir, d, g);
        float t = 0.0f;
        if (d > 0) t = clr_over_length(d, dir, g);
        if (clr_guard_ok(g)) {
            accepted[1]++;
            const float len = sqrtf(a*a + b*b + c*c);
            if (!same(dir.len, len)) report(out, 1, a, b, dir.len, len);
            if (!same(dir.unit.x, a / len)) report(out, 1, a, len, dir.unit.x, a / len);
            if (!same(dir.unit.y, b / len)) report(out, 1, b, len, dir.unit.y, b / len);
            if (!same(dir.unit.z, c / len)) report(out, 1, c, len, dir.unit.z, c / len);
            if (d > 0 && !same(t, d / len)) report(out, 0, d, len, t, d / len);
        }

        clr_guard_reset(g);
        const float r = clr_sqrt(a, g);
        if (clr_guard_ok(g)) { accepted[2]++; if (!same(r, sqrtf(a))) report(out, 2, a, 0.0f, r, sqrtf(a)); }
    }
    for (int k = 0; k < 3; k++) atomicAdd(out + 3 + k, accepted[k] >> 4);
}

// clr_sqrt_ranged against sqrtf on raw bit patterns and on exponents around its range limits;
// clr_log_quotient against the compiler's division on every operand pair clr_logf can hand it
// (clr_math.inl: m = bits + 0x3f3504f3 for the 2^23 mantissa patterns, f = m - 1, f / (2 + f))
extern "C" __global__ void check_ranged(uint *out, int iterations, uint seed) {
    const uint tid = blockIdx.x * blockDim.x + threadIdx.x, threads = gridDim.x * blockDim.x;
    uint s = tid * 2654435761u + seed;
    if (s == 0) s = 1;
    for (int i = 0; i < iterations; i++) {
        const float a = draw(s, i & 1, 126);
        const float r = clr_sqrt_ranged(a);
        if (!same(r, sqrtf(a))) report(out, 0, a, 0.0f, r, sqrtf(a));
    }
    for (uint k = tid; k < (1u << 23); k += threads) {
        const float f = __uint_as_float(k + 0x3f3504f3u) - 1.0f, d = 2.0f + f;
        const float q = clr_log_quotient(f, d);
        if (!same(q, f / d)) report(out, 1, f, d, q, f / d);
        if (k == 0 || k == (1u << 23) - 1) atomicAdd(out + 3, 1u);     // both ends of the enumeration were reached
    }
}

// generic arithmetic through the same interface must be the plain operators
extern "C" __global__ void eval_safe(const float *a, const float *b, float *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const clr_guard_safe g = {};
    clr_direction dir;
    dir.len = b[i];
    out[i] = clr_over_length(a[i], dir, g);
    out[n + i] = clr_sqrt(__builtin_fabsf(a[i]), g);
}
'''


@pytest.fixture(scope='module')
def acc():
    a = Accelerator()
    exact = open(os.path.join(TEMPLATES, 'clr_exact.inl')).read()
    a.build_program(KERNEL % {'exact': exact}, ['-O3', '-ffp-contract=off'])
    return a


@pytest.mark.parametrize('mode,span,min_accepted', [(0, 0, 0.0), (1, 8, 0.99), (1, 29, 0.99), (1, 45, 0.1), (1, 126, 0.0)])
def test_short_sequences_give_ieee_bits(acc, mode, span, min_accepted):
    out = acc.new_array((32,), np.uint32, zeros=True)
    threads, iterations = 1 << 20, 1 << 10
    acc.call('check_exact', threads, (out,), value_args=(np.int32(iterations), np.int32(mode),
                                                         np.int32(span), np.int32(12345 + span)))
    res = out.get()
    detail = {name: res[8 + 4*k: 12 + 4*k].view(np.float32).tolist()
              for k, name in enumerate(('t / length', 'direction', 'sqrt')) if res[k]}
    assert res[:3].sum() == 0, "mismatches %s, first (a, b, got, want): %s" % (res[:3].tolist(), detail)
    share = res[3:6].astype(np.float64) * 16 / (threads * iterations)
    # the test is vacuous if the guard rejects everything; sqrt sees negative operands half the time
    # (a distance is positive half of the time and so is a sqrt argument)
    assert share[0] >= min_accepted and share[1] >= min_accepted * 0.9 and share[2] >= min_accepted * 0.45, share


def test_safe_guard_is_the_plain_operator(acc):
    edge = np.array([0.0, -0.0, 1.0, -1.0, 3.0, 1e-45, 1.17549435e-38, 3.4028235e38, np.inf, -np.inf,
                     np.nan, 2.0**-47, 2.0**30, 2.0**-94, 2.0**120, 0.1, 1.0/3.0], dtype=np.float32)
    a, b = (m.ravel().copy() for m in np.meshgrid(edge, edge))
    n = a.size
    out = acc.new_array((2 * n,), np.float32, zeros=True)
    acc.call('eval_safe', n, (acc.to_device(a), acc.to_device(b), out), value_args=(np.int32(n),))
    got = out.get()
    with np.errstate(all='ignore'):
        want = np.concatenate([a / b, np.sqrt(np.abs(a))]).astype(np.float32)
    both_nan = np.isnan(got) & np.isnan(want)
    assert ((got.view(np.uint32) == want.view(np.uint32)) | both_nan).all()


def test_ranged_square_root_and_the_logarithms_quotient(acc):
    """clr_sqrt_ranged (the Julia-set marcher's lengths) equals sqrtf on 2^28 operands, half raw bit
    patterns, half spread over all exponents; clr_log_quotient equals the compiler's division on
    every one of the 2^23 operand pairs that clr_logf can produce."""
    out = acc.new_array((32,), np.uint32, zeros=True)
    threads, iterations = 1 << 18, 1 << 10
    acc.call('check_ranged', threads, (out,), value_args=(np.int32(iterations), np.uint32(777)))
    res = out.get()
    detail = {name: res[8 + 4*k: 12 + 4*k].view(np.float32).tolist()
              for k, name in enumerate(('sqrt', 'log quotient')) if res[k]}
    assert res[:2].sum() == 0, "mismatches %s, first (a, b, got, want): %s" % (res[:2].tolist(), detail)
    assert res[3] == 2
