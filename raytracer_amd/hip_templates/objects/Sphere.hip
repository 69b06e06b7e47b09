### extends 'tracer.hip'
{#- unit sphere at the origin of tracer coordinates (reference objects/Sphere.cl:3-38) -#}

### macro tracer_function(obj)
    ### call tracer_function_base(obj)
        float half_b = -dot(ray, origin);
        float o_len2 = dot(origin, origin);

        // heading away from the centre and not inside: no intersection
        if (half_b <= 0 && !inside) return;

        float disc = half_b*half_b - o_len2 + 1.0f;
        if (disc < 0) return;
        float root = native_sqrt(disc);

        float dist;
        if (inside) dist = half_b + root;
        else dist = half_b - root;

        if (dist <= 0) return;
        *t_hit = dist;
    ### endcall
### endmacro

### macro normal_function(obj)
    ### call normal_function_base(obj)
        *p_normal = pos;
    ### endcall
### endmacro
