{#- interval-arithmetic evaluation of f = |z_n|^2 - 4 for the quaternion iteration
    z -> z^2 + c started at (x, y, z, 0) (reference objects/QuaternionJuliaSet.cl:2-33);
    plugged into the ImplicitSurface marching loop -#}

### macro f_code(obj)
    ia_type qr = x;
    ia_type qi = y;
    ia_type qj = z;
    ia_type qk = ia_new(0.0f, 0.0f);
    const float cr = {{ obj.c_code[0] }}, ci = {{ obj.c_code[1] }},
                cj = {{ obj.c_code[2] }}, ck = {{ obj.c_code[3] }};
    for (int k = 0; k < {{ obj.julia_itr }}; ++k) {
        const ia_type qr1 = ia_sub(ia_pow2(qr), ia_pow2(qi) + ia_pow2(qj) + ia_pow2(qk)) + cr;
        qi = 2 * ia_mul(qr, qi) + ci;
        qj = 2 * ia_mul(qr, qj) + cj;
        qk = 2 * ia_mul(qr, qk) + ck;
        qr = qr1;
    }
    f = ia_pow2(qr) + ia_pow2(qi) + ia_pow2(qj) + ia_pow2(qk) - 4.0f;
### endmacro

### macro df_code(obj)
    df = ia_new_exact(1.0f);   // never consulted: the set is built with self_intersection=False
### endmacro
