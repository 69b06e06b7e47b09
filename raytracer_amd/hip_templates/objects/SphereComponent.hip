### extends 'tracer.hip'
{#- sphere of radius R (reference objects/SphereComponent.cl:3-34) -#}

### macro tracer_function(obj)
    ### call tracer_component_function_base(obj)
        float half_b = -dot(ray, origin);
        float o_len2 = dot(origin, origin);
        float disc = half_b*half_b - o_len2 + R*R;
        if (disc < 0) {
            // the ray misses the sphere: empty interval
            *t_enter = 1;
            *t_exit = 0;
            return;
        }
        float root = native_sqrt(disc);
        *t_enter = half_b - root;
        *t_exit = half_b + root;
    ### endcall
### endmacro

### macro normal_function(obj)
    ### call normal_function_base(obj)
        *p_normal = pos / R;
    ### endcall
### endmacro
