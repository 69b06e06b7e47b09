{#- Ray-marched shapes: interval-arithmetic root isolation (ImplicitSurface, also used by the
    interval Julia set), distance-estimator marching (QuaternionJuliaSet2) and sphere tracing of
    user-supplied distance code (DistanceField).  Macro pairs <Class>__tracer / <Class>__normal.
    Reference counterparts: cl_templates/objects/{bounding_volumes, ImplicitSurface,
    QuaternionJuliaSet, QuaternionJuliaSet2, DistanceField}.cl -#}
### from 'tracer.hip' import tracer_function_base, normal_function_base
### from 'helpers.hip' import vec3_literal

{#- clip [lo_var, hi_var] to a bounding sphere of radius R about the origin; `return`s from
    the enclosing tracer function on a miss (reference objects/bounding_volumes.cl:3-27) -#}
### macro sphere_bounding_volume(R, lo_var, hi_var)
    {
        const float bound_r2 = {{ (R*R)|float }}f;
        float half_b = -dot(ray, origin);
        float o_len2 = dot(origin, origin);
        bool inside_bound = o_len2 < bound_r2;
        if (half_b <= 0 && !inside_bound) return;
        const float disc = half_b*half_b - o_len2 + bound_r2;
        if (disc < 0) return;
        const float root = native_sqrt(disc);
        {{ lo_var }} = fmaxf(half_b - root, 0.0f);
        {{ hi_var }} = fminf(half_b + root, t_limit);
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
    (reference objects/ImplicitSurface.cl:6-120).  The parameter interval t is advanced in
    steps; when f(t) may contain the surface the interval is halved until it is shorter than
    the precision, otherwise the step is doubled again after two clean steps. -#}

### macro ImplicitSurface__tracer(obj)
    ### call tracer_function_base(obj)
        ### if obj.no_self_intersection
        if (on_surface) return;
        ### endif

        ia_type t;
        ### if obj.bndR
            {{ sphere_bounding_volume(obj.bndR, 'ia_begin(t)', 'ia_end(t)') }}
        ### else
            ia_begin(t) = 0.0f;
            ia_end(t) = t_limit;
        ### endif
        const float t_max = ia_end(t);

        const int MAX_ITER = {{ obj.max_itr }};
        const float TARGET_EPS = {{ obj.precision|float }}f;
        const float FRACTION = 0.5f;
        const float SELF_MAX_BEGIN_STEP = 0.1f;

        ia_type f, df;
        ia_type x, y, z;
        float step;
        int steps_since_subdiv = 0;

        const ia_type ray_x = ia_new_exact(ray.x);
        const ia_type ray_y = ia_new_exact(ray.y);
        const ia_type ray_z = ia_new_exact(ray.z);
        const ia_type origin_x = ia_new_exact(origin.x);
        const ia_type origin_y = ia_new_exact(origin.y);
        const ia_type origin_z = ia_new_exact(origin.z);
        int need_subdiv;

        if (on_surface) ia_end(t) = fminf(ia_end(t), ia_begin(t) + SELF_MAX_BEGIN_STEP);

        for (int i = 0; i < MAX_ITER; i++) {
            if (ia_begin(t) >= t_max) return;
            if (ia_end(t) > t_max) ia_end(t) = t_max;

            x = ia_mul(ray_x, t) + origin_x;
            y = ia_mul(ray_y, t) + origin_y;
            z = ia_mul(ray_z, t) + origin_z;

            ### if obj.f_code_template() == 'QuaternionJuliaSet'
                {{ julia_interval_f(obj) }}
            ### else
                {{ obj.f_code }}
            ### endif

            step = ia_len(t);
            need_subdiv = 0;

            if ((inside && ia_end(f) > 0) || (!inside && ia_begin(f) < 0)) {
                need_subdiv = 1;
                if (on_surface) {
                    ### if obj.f_code_template() == 'QuaternionJuliaSet'
                        {{ julia_interval_df(obj) }}
                    ### else
                        {{ obj.df_code }}
                    ### endif
                    if (!ia_contains_zero(df)) {
                        if ((ia_end(df) < 0) == inside) need_subdiv = 0;
                    }
                }
                if (need_subdiv) {
                    if (step < TARGET_EPS || i == MAX_ITER-1) {
                        *t_hit = ia_center(t);
                        return;
                    }
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
        float march_begin, march_end;
        {{ sphere_bounding_volume(obj.bndR, 'march_begin', 'march_end') }}

        const int MAX_ITER = {{ obj.max_itr }};
        const float TARGET_EPS = {{ obj.precision|float }}f;
        const float MAX_STEP = {{ obj.max_step|float }}f;
        const vec4 c = {{ obj.c_as_float4 }};
        float step, dist = march_begin;
        vec3 pos = march_begin * ray + origin;

        for (int i = 0; i < MAX_ITER; i++) {
            // (the reference's directional-derivative chain only runs when the ray starts on
            //  this surface, which returned above)
            // The derivative starts at (1, 0, 0, 0), so its first update 2 q (1, 0, 0, 0) is 2 q: for
            // finite q the general product returns exactly that, up to the sign of components that
            // are zero, which no later sum with a non-zero term and no length can see.  The compiler
            // may not drop the 28 operations with the zeros (x * 0 is not 0 for every x); here they are.
            ### if obj.julia_itr >= 1
            vec4 q = make_vec4(pos, 0.0f), qd = 2*q;
            q = quaternion_square(q) + c;
            ### else
            vec4 q = make_vec4(pos, 0.0f), qd = make_vec4(1.0f, 0.0f, 0.0f, 0.0f);
            ### endif
            for (int k = 1; k < {{ obj.julia_itr }}; ++k) {
                qd = 2*quaternion_mult(q, qd);
                q = quaternion_square(q) + c;
            }
            float l = length(q);
            // `0.5 * l * log(l) / length(qd)` with a double literal: one rounding at the end
            step = (float)(0.5 * (double)l * (double)clr_logf(l) / (double)length(qd));
            step = fminf(step, MAX_STEP);
            if (step < TARGET_EPS) {
                *t_hit = dist + step;
                return;
            }
            dist += step;
            if (dist > march_end) return;
            pos += step * ray;
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
        float t, t_end;
        float sign = 1.0f;
        if (inside) sign = -1.0f;
        ### if obj.bndR
            {{ sphere_bounding_volume(obj.bndR, 't', 't_end') }}
        ### else
            t = 0.0f;
            t_end = t_limit;
        ### endif

        const int MAX_ITER = {{ obj.max_itr }};
        const float EPS = {{ obj.precision|float }}f;
        float x, y, z;
        float last_dist = 0.0f;
        float derivative = -1.0f;

        for (int i = 0; i < MAX_ITER; i++) {
            if (t > t_end) return;
            const vec3 pos = origin + t * ray;
            x = pos.x; y = pos.y; z = pos.z;
            float dist = 0.0f;
            {
            {{ obj.tracer_code }};
            }
            dist *= sign;

            if (dist < EPS) {
                if (on_surface) {
                    // leaving the own surface: accept only once the distance decreases
                    if (i > 0) derivative = dist - last_dist;
                    else derivative = 0.0f;
                }
                if (derivative < 0.0f) {
                    if (t < t_end) *t_hit = t;
                    return;
                }
            }
            t += fmaxf(dist, EPS);
            last_dist = dist;
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
