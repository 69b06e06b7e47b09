### extends 'tracer.hip'
{#- half-space normal.x <= h (reference objects/HalfSpaceComponent.cl:3-23) -#}

### macro tracer_function(obj)
    ### call tracer_component_function_base(obj)
        float slope = dot(ray, normal);
        float dist = dot(-origin, normal) + h;
        dist = dist/slope;
        if (slope < 0) *t_enter = dist;
        else *t_exit = dist;
        *p_subobject = 0;
    ### endcall
### endmacro

### macro normal_function(obj)
    ### call normal_function_base(obj)
        *p_normal = normal;
    ### endcall
### endmacro
