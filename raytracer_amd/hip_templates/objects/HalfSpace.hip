### extends 'tracer.hip'
{#- half-space x <= 0 in tracer coordinates (reference objects/HalfSpace.cl:3-18) -#}

### macro tracer_function(obj)
    ### call tracer_function_base(obj)
        const float dist = -origin.x/ray.x;
        if (dist > 0) *t_hit = dist;
    ### endcall
### endmacro

### macro normal_function(obj)
    ### call normal_function_base(obj)
        *p_normal = make_vec3(1.0f, 0.0f, 0.0f);
    ### endcall
### endmacro
