/* clr_math.inl - platform-independent single-precision exp, log and tan.
 *
 * The shading pipeline needs three transcendental functions (shader.cl:130-167 uses exp,
 * log and tan; QuaternionJuliaSet2.cl:61 uses log).  OpenCL leaves their last bits to the
 * implementation (<= 3-5 ulp); vendor libraries differ from each other and from glibc.  So
 * that the device and the CPU oracle produce identical bits, both use the functions below:
 * plain binary32 +, -, *, / and rint in a fixed order (no FMA: the file is compiled with
 * -ffp-contract=off), accurate to about 2 ulp (tests/test_clr_math.py).
 *
 * Written in the common subset of C and HIP; CLR_MATH_FN supplies the qualifiers.
 */
#ifndef CLR_LOG_QUOTIENT
#define CLR_LOG_QUOTIENT(a, d) ((a) / (d))
#endif
CLR_MATH_FN float clr_bits_to_float(int i) { float f; __builtin_memcpy(&f, &i, 4); return f; }
CLR_MATH_FN int clr_float_to_bits(float f) { int i; __builtin_memcpy(&i, &f, 4); return i; }

CLR_MATH_FN float clr_expf(float x) {
    if (!(x >= -103.972084f && x <= 88.7228394f)) {        /* one test on the usual way */
        if (!(x == x)) return x;                           /* NaN */
        if (x > 88.7228394f) return clr_bits_to_float(0x7f800000);
        return 0.0f;                                       /* x < -103.972084 */
    }
    const float k = __builtin_rintf(x * 1.44269502f);
    /* r = x - k ln2, ln2 split so that k * hi is exact */
    const float r = (x - k * 0.693145752f) - k * 1.42860677e-06f;
    float p = 1.98412701e-04f;                             /* 1/5040 */
    p = p * r + 1.38888892e-03f;
    p = p * r + 8.33333377e-03f;
    p = p * r + 4.16666679e-02f;
    p = p * r + 1.66666672e-01f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    int ki = (int)k;
    if (ki < -126) {                                       /* subnormal result: scale twice */
        p = p * clr_bits_to_float((ki + 64 + 127) << 23);
        return p * 5.42101086e-20f;                        /* 2^-64 */
    }
    if (ki > 127) { p = p * 2.0f; ki -= 1; }
    return p * clr_bits_to_float((ki + 127) << 23);
}

CLR_MATH_FN float clr_logf(float x) {
    int e = 0;
    if (!(x >= 1.17549435e-38f && x < clr_bits_to_float(0x7f800000))) {   /* one test on the usual way */
        if (!(x == x)) return x;
        if (x < 0.0f) return clr_bits_to_float(0x7fc00000);
        if (x == 0.0f) return clr_bits_to_float((int)0xff800000);
        if (x == clr_bits_to_float(0x7f800000)) return x;
        x = x * 8388608.0f;                                    /* subnormal input */
        e = -23;
    }
    int bits = clr_float_to_bits(x);
    /* mantissa m in [sqrt(1/2), sqrt(2)) */
    bits += 0x3f800000 - 0x3f3504f3;
    e += (bits >> 23) - 127;
    bits = (bits & 0x007fffff) + 0x3f3504f3;
    const float f = clr_bits_to_float(bits) - 1.0f;
    /* On the device the quotient goes through the short exact sequence (clr_exact.inl): f is 0 or at
     * least 2^-24 in magnitude (a difference of binary32 numbers next to 1), the divisor lies in
     * [1.7, 2.42], so the operands are inside its range by construction and +0 / d comes out +0. */
    const float s = CLR_LOG_QUOTIENT(f, 2.0f + f);
    const float z = s * s;
    float t = 0.153846154f;                                /* 2/13 */
    t = t * z + 0.181818187f;                              /* 2/11 */
    t = t * z + 0.222222224f;                              /* 2/9 */
    t = t * z + 0.285714298f;                              /* 2/7 */
    t = t * z + 0.400000006f;                              /* 2/5 */
    t = t * z + 0.666666687f;                              /* 2/3 */
    /* log(1+f) = f - s*(f - z*t)  (since 2s = f - s*f) */
    const float log_m = f - s * (f - z * t);
    const float ef = (float)e;
    return ef * 0.693145752f + (ef * 1.42860677e-06f + log_m);
}

CLR_MATH_FN float clr_tanf(float x) {
    if (!(x == x) || x - x != 0.0f) return clr_bits_to_float(0x7fc00000);
    /* reduce to r in [-pi/4, pi/4]: x = k pi/2 + r, pi/2 split in three parts */
    const float k = __builtin_rintf(x * 0.636619747f);
    float r = x - k * 1.5703125f;
    r = r - k * 4.83751297e-04f;
    r = r - k * 7.54978942e-08f;
    const float z = r * r;
    /* sin r / r and cos r */
    float sp = -2.50521079e-08f;                           /* -1/11! */
    sp = sp * z + 2.75573188e-06f;
    sp = sp * z - 1.98412701e-04f;
    sp = sp * z + 8.33333377e-03f;
    sp = sp * z - 1.66666672e-01f;
    const float sn = r + r * (z * sp);
    float cp = 2.08767559e-09f;                            /* 1/12! */
    cp = cp * z - 2.75573188e-07f;
    cp = cp * z + 2.48015876e-05f;
    cp = cp * z - 1.38888892e-03f;
    cp = cp * z + 4.16666679e-02f;
    const float cs = (1.0f - 0.5f * z) + z * (z * cp);
    if (((int)k) & 1) return -cs / sn;
    return sn / cs;
}
