### extends 'tracer.hip'
{#- infinite cylinder of radius R about `axis` (reference objects/CylinderComponent.cl:3-44) -#}

### macro tracer_function(obj)
    ### call tracer_component_function_base(obj)
        float o_ax = dot(origin, axis), d_ax = dot(ray, axis);
        vec3 o_perp = origin - o_ax*axis;
        vec3 d_perp = ray - d_ax*axis;
        float half_b = dot(d_perp, o_perp);
        float o_perp2 = dot(o_perp, o_perp);
        float d_perp2 = dot(d_perp, d_perp);
        float disc = half_b*half_b - d_perp2*(o_perp2 - R*R);
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
        vec3 o_perp = pos - dot(pos, axis)*axis;
        *p_normal = o_perp / R;
    ### endcall
### endmacro
