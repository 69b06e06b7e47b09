### extends 'tracer.hip'
### from 'objects/bounding_volumes.hip' import sphere_bounding_volume
{#- quaternion Julia set z -> z^2 + c marched with the distance estimate
    0.5 |z| ln|z| / |z'| of Hart, Sandin & Kauffman 1989
    (reference objects/QuaternionJuliaSet2.cl:6-123) -#}

### macro tracer_function(obj)
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
            vec4 q = make_vec4(pos, 0.0f), qd = make_vec4(1.0f, 0.0f, 0.0f, 0.0f);
            // (the reference's directional-derivative chain only runs when the ray starts on
            //  this surface, which returned above)
            for (int k = 0; k < {{ obj.julia_itr }}; ++k) {
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

### macro normal_function(obj)
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
