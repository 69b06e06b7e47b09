### extends 'tracer.hip'
### from 'helpers.hip' import vec3_literal
### from 'objects/bounding_volumes.hip' import sphere_bounding_volume
{#- implicit surface f = 0 located by interval arithmetic along the ray
    (reference objects/ImplicitSurface.cl:6-120).  The parameter interval t is advanced in
    steps; when f(t) may contain the surface the interval is halved until it is shorter than
    the precision, otherwise the step is doubled again after two clean steps. -#}

### macro tracer_function(obj)
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

            ### if obj.f_code_template()
                ### import obj.f_code_template() as custom
                {{ custom.f_code(obj) }}
            ### else
                {{ obj.f_code }}
            ### endif

            step = ia_len(t);
            need_subdiv = 0;

            if ((inside && ia_end(f) > 0) || (!inside && ia_begin(f) < 0)) {
                need_subdiv = 1;
                if (on_surface) {
                    ### if obj.f_code_template()
                        {{ custom.df_code(obj) }}
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

### macro normal_function(obj)
    ### call normal_function_base(obj)
        *p_normal = fast_normalize({{ vec3_literal(obj.gradient_code) }});
    ### endcall
### endmacro
