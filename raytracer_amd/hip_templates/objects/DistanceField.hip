### extends 'tracer.hip'
### from 'objects/bounding_volumes.hip' import sphere_bounding_volume
{#- sphere tracing of a signed distance field given as a code string that assigns `dist`
    from x, y, z (reference objects/DistanceField.cl:6-113) -#}

### macro tracer_function(obj)
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

### macro normal_function(obj)
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
