/* clr_rng.inl - counter-based random numbers per (pixel, pass, segment) and the samplers the
 * path tracer needs, shared verbatim by the device code and the CPU oracle (same bits).
 *
 * The reference draws every random number on the host and hands the SAME numbers to all pixels
 * of a frame (renderer.py:357-417); that mode is the replayed tape.  This file is
 * the other mode ("throughput runs"): every pixel has its own stream,
 * Philox-4x32-10 (Salmon et al., SC'11) keyed by the 64-bit seed with the counter
 * (pixel, pass, segment, block).  No counterpart in the reference, hence no parity claim
 * against it: HIP is checked against the oracle bit for bit and against the tape mode
 * statistically (tests/test_gpu_parity.py).
 *
 * Common subset of C and HIP, plain binary32 arithmetic in a fixed order (compiled without
 * contraction), CLR_MATH_FN supplies the qualifiers; needs clr_math.inl (clr_logf).
 */
#ifndef CLR_PHILOX_ROUNDS
#define CLR_PHILOX_ROUNDS 10         /* Philox-4x32-10, the published default; other values: measurement only */
#endif
CLR_MATH_FN unsigned clr_mulhi32(unsigned a, unsigned b) {
    return (unsigned)(((unsigned long long)a * (unsigned long long)b) >> 32);
}

CLR_MATH_FN void clr_philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                unsigned k0, unsigned k1, unsigned out[4]) {
    for (int round = 0; round < CLR_PHILOX_ROUNDS; ++round) {
#ifdef CLR_PHILOX_SPLIT_MUL     /* measurement only: high and low half as two multiplications */
        const unsigned hi0 = clr_mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = clr_mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
#else
        /* both halves of a product from one 32 x 32 -> 64 bit multiplication (v_mad_u64_u32 on the
         * device: the integer multiplier is the slow unit, and this halves its work) */
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * (unsigned long long)c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * (unsigned long long)c2;
        const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0;
        const unsigned hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
#endif
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 24 random bits -> [0, 1), exact */
CLR_MATH_FN float clr_u01(unsigned x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }

/* sin and cos of 2 pi u for u in [0, 1): quadrant + Taylor polynomials on [-pi/4, pi/4)
 * (truncation error < 3e-9), absolute error about 1e-7 */
CLR_MATH_FN void clr_sincos_2pi(float u, float *sin_out, float *cos_out) {
    const float x = u * 4.0f;
    int q = (int)x;
    if (q > 3) q = 3;
    const float a = ((x - (float)q) - 0.5f) * 1.57079637f;
    const float a2 = a * a;
    float s = 2.75573192e-06f;                    /* 1/9! */
    s = s * a2 - 1.98412701e-04f;
    s = s * a2 + 8.33333377e-03f;
    s = s * a2 - 1.66666672e-01f;
    s = s * a2 + 1.0f;
    s = s * a;
    float c = -2.75573200e-07f;                   /* -1/10! */
    c = c * a2 + 2.48015876e-05f;
    c = c * a2 - 1.38888892e-03f;
    c = c * a2 + 4.16666679e-02f;
    c = c * a2 - 0.5f;
    c = c * a2 + 1.0f;
    /* angle inside the quadrant = pi/4 + a */
    const float sb = (s + c) * 0.707106769f, cb = (c - s) * 0.707106769f;
    if (q == 0) { *sin_out = sb; *cos_out = cb; }
    else if (q == 1) { *sin_out = cb; *cos_out = -sb; }
    else if (q == 2) { *sin_out = -sb; *cos_out = -cb; }
    else { *sin_out = -cb; *cos_out = sb; }
}

/* uniform direction on the unit sphere from two uniforms */
CLR_MATH_FN void clr_sphere_direction(float u1, float u2, float out[3]) {
    const float z = 1.0f - 2.0f * u1;
    float r2 = 1.0f - z * z;
    if (r2 < 0.0f) r2 = 0.0f;
    const float r = __builtin_sqrtf(r2);
    float s, c;
    clr_sincos_2pi(u2, &s, &c);
    out[0] = r * c; out[1] = r * s; out[2] = z;
}

/* two independent standard normal numbers (Box-Muller) */
CLR_MATH_FN void clr_normal_pair(float u1, float u2, float *g0, float *g1) {
    const float radius = __builtin_sqrtf(-2.0f * clr_logf(1.0f - u1));     /* 1 - u1 in (0, 1] */
    float s, c;
    clr_sincos_2pi(u2, &s, &c);
    *g0 = radius * c;
    *g1 = radius * s;
}

/* The random numbers of one path segment, per pixel:
 *   block 0: direction on the sphere (2), stage sample p, Russian-roulette sample
 *   block 1: gaussian vector (3 of 4 normals)
 *   block 2: light choice, point on the light (2)          - only with next-event lights
 * The blocks are independent functions of the counter, so a caller may evaluate only the ones
 * it consumes (the device code draws blocks 1 and 2 where they are used) and still sees the
 * numbers of clr_draw_segment, which evaluates all of them. */
typedef struct clr_segment_draw {
    float rvec[3], p, gauss[3], russian, light_u, light_dir[3];
    float light_uniforms[3];     /* the three uniforms behind light_dir and one more: box lights */
} clr_segment_draw;

CLR_MATH_FN void clr_draw_block0(unsigned seed_lo, unsigned seed_hi, unsigned pixel, unsigned pass,
                                 unsigned segment, float rvec[3], float *p, float *russian) {
    unsigned r[4];
    clr_philox4x32(pixel, pass, segment, 0u, seed_lo, seed_hi, r);
    clr_sphere_direction(clr_u01(r[0]), clr_u01(r[1]), rvec);
    *p = clr_u01(r[2]);
    *russian = clr_u01(r[3]);
}

CLR_MATH_FN void clr_draw_gauss(unsigned seed_lo, unsigned seed_hi, unsigned pixel, unsigned pass,
                                unsigned segment, float gauss[3]) {
    unsigned r[4];
    float unused;
    clr_philox4x32(pixel, pass, segment, 1u, seed_lo, seed_hi, r);
    clr_normal_pair(clr_u01(r[0]), clr_u01(r[1]), &gauss[0], &gauss[1]);
    clr_normal_pair(clr_u01(r[2]), clr_u01(r[3]), &gauss[2], &unused);
}

CLR_MATH_FN void clr_draw_light(unsigned seed_lo, unsigned seed_hi, unsigned pixel, unsigned pass,
                                unsigned segment, float *light_u, float light_dir[3], float uniforms[3]) {
    unsigned r[4];
    clr_philox4x32(pixel, pass, segment, 2u, seed_lo, seed_hi, r);
    *light_u = clr_u01(r[0]);
    uniforms[0] = clr_u01(r[1]); uniforms[1] = clr_u01(r[2]); uniforms[2] = clr_u01(r[3]);
    clr_sphere_direction(uniforms[0], uniforms[1], light_dir);
}

/* Further points of the same light (light_samples > 1): point `sample` >= 1 of a segment takes block
 * 2 + sample of the pixel's stream; its first word (the choice of the light) is not used. */
CLR_MATH_FN void clr_draw_light_extra(unsigned seed_lo, unsigned seed_hi, unsigned pixel, unsigned pass,
                                      unsigned segment, unsigned sample, float light_dir[3], float uniforms[3]) {
    unsigned r[4];
    clr_philox4x32(pixel, pass, segment, 2u + sample, seed_lo, seed_hi, r);
    uniforms[0] = clr_u01(r[1]); uniforms[1] = clr_u01(r[2]); uniforms[2] = clr_u01(r[3]);
    clr_sphere_direction(uniforms[0], uniforms[1], light_dir);
}

/* Point and outward normal on a box light (Parallelepiped.random_surface_point_and_normal).
 * faces: 6 records of 16 floats = (start corner, share of the total area up to and including this
 * face), (edge u, 0), (edge v, 0), (outward unit normal, 0); the face is picked by area. */
CLR_MATH_FN void clr_box_light_point(const float *faces, float pick, float a, float b,
                                     float point[3], float normal[3]) {
    int f = 0;
    while (f < 5 && pick >= faces[16 * f + 3]) f++;
    const float *F = faces + 16 * f;
    for (int c = 0; c < 3; ++c) {
        point[c] = (F[c] + a * F[4 + c]) + b * F[8 + c];
        normal[c] = F[12 + c];
    }
}

CLR_MATH_FN void clr_draw_segment(unsigned seed_lo, unsigned seed_hi, unsigned pixel, unsigned pass,
                                  unsigned segment, int with_light, clr_segment_draw *d) {
    clr_draw_block0(seed_lo, seed_hi, pixel, pass, segment, d->rvec, &d->p, &d->russian);
    clr_draw_gauss(seed_lo, seed_hi, pixel, pass, segment, d->gauss);
    d->light_u = 0.0f;
    d->light_dir[0] = d->light_dir[1] = 0.0f; d->light_dir[2] = 1.0f;
    d->light_uniforms[0] = d->light_uniforms[1] = d->light_uniforms[2] = 0.0f;
    if (with_light) clr_draw_light(seed_lo, seed_hi, pixel, pass, segment, &d->light_u, d->light_dir, d->light_uniforms);
}
