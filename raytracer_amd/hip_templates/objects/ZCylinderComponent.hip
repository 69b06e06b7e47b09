### extends 'tracer.hip'
{#- infinite unit cylinder about z (reference objects/ZCylinderComponent.cl:3-43) -#}

### macro tracer_function(obj)
    ### call tracer_component_function_base(obj)
        vec2 o_perp = make_vec2(origin.x, origin.y);
        vec2 d_perp = make_vec2(ray.x, ray.y);
        float half_b = dot(d_perp, o_perp);
        float o_perp2 = dot(o_perp, o_perp);
        float d_perp2 = dot(d_perp, d_perp);
        float disc = half_b*half_b - d_perp2*(o_perp2 - 1.0f);
        if (disc < 0) {
            *t_enter = 1;
            *t_exit = 0;
            return;
        }
        float root = native_sqrt(disc);
        float d1 = -half_b - root;
        *t_enter = d1 / d_perp2;
        *t_exit = (d1 + 2*root) / d_perp2;
    ### endcall
### endmacro

### macro normal_function(obj)
    ### call normal_function_base(obj)
        vec3 o_perp = pos;
        o_perp.z = 0;
        *p_normal = o_perp;
    ### endcall
### endmacro
