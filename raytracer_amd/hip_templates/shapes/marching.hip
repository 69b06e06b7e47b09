{#- Ray-marched shapes: interval-arithmetic root isolation (ImplicitSurface, also used by the
    interval Julia set), distance-estimator marching (QuaternionJuliaSet2) and sphere tracing of
    user-supplied distance code (DistanceField).  Macro pairs <Class>__tracer / <Class>__normal.
    Reference counterparts: cl_templates/objects/{bounding_volumes, ImplicitSurface,
    QuaternionJuliaSet, QuaternionJuliaSet2, DistanceField}.cl -#}
### from 'tracer.hip' import tracer_function_base, normal_function_base
### from 'helpers.hip' import vec3_literal

{#- Part of the ray inside a sphere of radius R about the local origin, cut to [0, t_limit]:
    assigns it to lo_var / hi_var, or `return`s from the enclosing tracer function when there is
    none (reference objects/bounding_volumes.cl:3-27).  `ray` is a unit vector, so the quadratic
    is t^2 - 2 m t + (|o|^2 - R^2) with m = -ray.o. -#}
### macro sphere_bounding_volume(R, lo_var, hi_var)
    {
        const float r2 = {{ (R*R)|float }}f;
        const float m = -dot(ray, origin), oo = dot(origin, origin);
        if (m <= 0 && !(oo < r2)) return;          // outside and heading away
        const float disc = m*m - oo + r2;
        if (disc < 0) return;
        const float h = native_sqrt(disc);
        {{ lo_var }} = fmaxf(m - h, 0.0f);
        {{ hi_var }} = fminf(m + h, t_limit);
        if ({{ hi_var }} <= {{ lo_var }}) return;
    }
### endmacro

{#- interval-arithmetic evaluation of f = |z_n|^2 - 4 for the quaternion iteration
    z -> z^2 + c started at (x, y, z, 0) (reference objects/QuaternionJuliaSet.cl:2-33);
    plugged into the ImplicitSurface marching loop -#}

### macro julia_interval_f(obj)
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

### macro julia_interval_df(obj)
    df = ia_new_exact(1.0f);   // never consulted: the set is built with self_intersection=False
### endmacro

{#- implicit surface f = 0 located by interval arithmetic along the ray
    (reference objects/ImplicitSurface.cl:6-120).  The ray is examined piece by piece: f over a
    piece is bounded with interval arithmetic; a piece that may hold a sign change of the wanted
    direction is halved until it is shorter than the precision (its centre is the hit), a piece that
    cannot is passed, and after two passed pieces in a row the piece length doubles.
    The generated expressions (objects/marching.py) read x, y, z, t, ray_*, origin_* and assign f
    and df: sympy orders sums and products by symbol name, so these names fix the order of the
    floating-point operations and stay as the reference has them. -#}

### macro ImplicitSurface__tracer(obj)
    ### call tracer_function_base(obj)
        ### if obj.no_self_intersection
        if (on_surface) return;
        ### endif

        ia_type t;                               // the piece of the ray under examination
        ### if obj.bndR
            {{ sphere_bounding_volume(obj.bndR, 'ia_begin(t)', 'ia_end(t)') }}
        ### else
            t = ia_new(0.0f, t_limit);
        ### endif
        const float t_stop = ia_end(t);
        const int piece_budget = {{ obj.max_itr }};
        const float resolution = {{ obj.precision|float }}f;

        const ia_type ray_x = ia_new_exact(ray.x), ray_y = ia_new_exact(ray.y), ray_z = ia_new_exact(ray.z);
        const ia_type origin_x = ia_new_exact(origin.x), origin_y = ia_new_exact(origin.y),
                      origin_z = ia_new_exact(origin.z);

        // leaving the own surface: a short first piece, so that the derivative test below can tell
        // the start point from the next root
        if (on_surface) ia_end(t) = fminf(ia_end(t), ia_begin(t) + 0.1f);

        int passed = 0;                          // pieces passed since the last halving
        for (int piece = 0; piece < piece_budget; ++piece) {
            if (ia_begin(t) >= t_stop) return;
            ia_end(t) = ia_end(t) > t_stop ? t_stop : ia_end(t);

            const ia_type x = ia_mul(ray_x, t) + origin_x,
                          y = ia_mul(ray_y, t) + origin_y,
                          z = ia_mul(ray_z, t) + origin_z;
            ia_type f;
            ### if obj.f_code_template() == 'QuaternionJuliaSet'
                {{ julia_interval_f(obj) }}
            ### else
                {{ obj.f_code }}
            ### endif

            // f > 0 outside the body: a ray inside looks for f reaching above zero, one outside below
            bool may_cross = inside ? ia_end(f) > 0 : ia_begin(f) < 0;
            if (may_cross && on_surface) {
                ia_type df;
                ### if obj.f_code_template() == 'QuaternionJuliaSet'
                    {{ julia_interval_df(obj) }}
                ### else
                    {{ obj.df_code }}
                ### endif
                // f moves away from zero over the whole piece: it is the start point's own root
                if (!ia_contains_zero(df) && (ia_end(df) < 0) == inside) may_cross = false;
            }

            float width = ia_len(t);
            if (may_cross) {
                if (width < resolution || piece == piece_budget - 1) {
                    *t_hit = ia_center(t);
                    return;
                }
                ia_end(t) = ia_begin(t) + width * 0.5f;
                passed = 0;
            } else {
                if (++passed > 1) width *= 2.0f;
                t = ia_new(ia_end(t), ia_end(t) + width);
            }
        }
    ### endcall
### endmacro

### macro ImplicitSurface__normal(obj)
    ### call normal_function_base(obj)
        *p_normal = fast_normalize({{ vec3_literal(obj.gradient_code) }});
    ### endcall
### endmacro

{#- quaternion Julia set z -> z^2 + c marched with the distance estimate
    0.5 |z| ln|z| / |z'| of Hart, Sandin & Kauffman 1989
    (reference objects/QuaternionJuliaSet2.cl:6-123) -#}

### macro QuaternionJuliaSet2__tracer(obj)
    ### call tracer_function_base(obj)
        if (on_surface) return;
        float travelled, t_stop;
        {{ sphere_bounding_volume(obj.bndR, 'travelled', 't_stop') }}

        const float resolution = {{ obj.precision|float }}f, longest_step = {{ obj.max_step|float }}f;
        const vec4 c = {{ obj.c_as_float4 }};
        vec3 at = travelled * ray + origin;

        for (int n = 0; n < {{ obj.max_itr }}; ++n) {
            // (the reference's directional-derivative chain only runs when the ray starts on
            //  this surface, which returned above)
            // The derivative starts at (1, 0, 0, 0), so its first update 2 q (1, 0, 0, 0) is 2 q: for
            // finite q the general product returns exactly that, up to the sign of components that
            // are zero, which no later sum with a non-zero term and no length can see.  The compiler
            // may not drop the 28 operations with the zeros (x * 0 is not 0 for every x); here they are.
            ### if obj.julia_itr >= 1
            vec4 q = make_vec4(at, 0.0f), qd = 2*q;
            q = quaternion_square(q) + c;
            ### else
            vec4 q = make_vec4(at, 0.0f), qd = make_vec4(1.0f, 0.0f, 0.0f, 0.0f);
            ### endif
            for (int k = 1; k < {{ obj.julia_itr }}; ++k) {
                qd = 2*quaternion_mult(q, qd);
                q = quaternion_square(q) + c;
            }
            // (the two lengths: correctly rounded roots, by the short sequence where the sums of
            //  squares are inside its range - the iteration can leave it)
            const float l = clr_sqrt_ranged(dot(q, q));
            // `0.5 * l * log(l) / length(qd)` with a double literal: one rounding at the end
            const float estimate = (float)(0.5 * (double)l * (double)clr_logf(l) / (double)clr_sqrt_ranged(dot(qd, qd)));
            const float step = fminf(estimate, longest_step);
            if (step < resolution) {
                *t_hit = travelled + step;
                return;
            }
            travelled += step;
            if (travelled > t_stop) return;
            at += step * ray;
        }
    ### endcall
### endmacro

### macro QuaternionJuliaSet2__normal(obj)
    ### call normal_function_base(obj)
        vec4 q = make_vec4(pos, 0.0f);
        const vec4 c = {{ obj.c_as_float4 }};
        // gradient of each quaternion component with respect to the start point
        vec3 gr = make_vec3(1.0f, 0.0f, 0.0f), gi = make_vec3(0.0f, 1.0f, 0.0f),
             gj = make_vec3(0.0f, 0.0f, 1.0f), gk = make_vec3(0.0f, 0.0f, 0.0f);
        for (int k = 0; k < {{ obj.julia_itr }}; ++k) {
            const vec3 gr1 = 2*(q.x*gr - q.y*gi - q.z*gj - q.w*gk);
            const vec3 gi1 = 2*(gr*q.y + q.x*gi);
            const vec3 gj1 = 2*(gr*q.z + q.x*gj);
            const vec3 gk1 = 2*(gr*q.w + q.x*gk);
            gr = gr1; gi = gi1; gj = gj1; gk = gk1;
            q = quaternion_square(q) + c;
        }
        const vec3 grad = fast_normalize(2*(q.x*gr + q.y*gi + q.z*gj + q.w*gk));
        if (isfinite(grad.x) && isfinite(grad.y) && isfinite(grad.z)) *p_normal = grad;
        else *p_normal = make_vec3(1.0f, 0.0f, 0.0f);
    ### endcall
### endmacro

{#- sphere tracing of a signed distance field given as a code string that assigns `dist`
    from x, y, z (reference objects/DistanceField.cl:6-113) -#}

### macro DistanceField__tracer(obj)
    ### call tracer_function_base(obj)
        ### if obj.no_self_intersection
        if (on_surface) return;
        ### endif
        float t, t_stop;
        ### if obj.bndR
            {{ sphere_bounding_volume(obj.bndR, 't', 't_stop') }}
        ### else
            t = 0.0f;
            t_stop = t_limit;
        ### endif

        const float resolution = {{ obj.precision|float }}f;
        const float side = inside ? -1.0f : 1.0f;   // the field is positive outside the body
        float previous = 0.0f;                      // distance at the step before
        float x, y, z;                              // read by the user's code

        for (int n = 0; n < {{ obj.max_itr }}; ++n) {
            if (t > t_stop) return;
            const vec3 at = origin + t * ray;
            x = at.x; y = at.y; z = at.z;
            float dist = 0.0f;
            {
            {{ obj.tracer_code }};
            }
            dist *= side;

            if (dist < resolution) {
                // A ray that leaves this surface starts within the resolution of it: it may stop
                // only where the distance has been falling (never at its first sample).
                const bool falling = on_surface ? (n > 0 && dist - previous < 0.0f) : true;
                if (falling) {
                    if (t < t_stop) *t_hit = t;
                    return;
                }
            }
            t += fmaxf(dist, resolution);
            previous = dist;
        }
    ### endcall
### endmacro

### macro DistanceField__normal(obj)
    ### call normal_function_base(obj)
        vec3 normal = make_vec3(0.0f, 0.0f, 0.0f);
        float x, y, z;
        ### if obj.normal_code
            x = pos.x; y = pos.y; z = pos.z;
            {
            {{ obj.normal_code }}
            }
        ### else
            // central differences of the distance function
            const float EPS = {{ obj.precision|float }}f;
            float delta, dist;
            ### for var in ['x', 'y', 'z']
                delta = 0.0f;
                ### for sign in [1, -1]
                    x = pos.x; y = pos.y; z = pos.z;
                    {{ var }} += 0.5f * EPS * {{ sign }};
                    {
                    {{ obj.tracer_code }};
                    }
                    delta += dist * {{ sign }};
                ### endfor
                normal.{{ var }} = delta;
            ### endfor
        ### endif
        *p_normal = normalize(normal);
    ### endcall
### endmacro
