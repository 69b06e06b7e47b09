### extends 'tracer.hip'
{#- infinite unit cylinder about z (reference objects/ZCylinderComponent.cl:3-43) -#}

### macro tracer_function(obj)
    ### call tracer_component_function_base(obj)
        vec2 o_perp = make_vec2(origin.x, origin.y);
        vec2 d_perp = make_vec2(ray.x, ray.y);
        float half_b = dot(d_perp, o_perp);
        float o_perp2 = dot(o_perp, o_perp);
        float d_perp2 = dot(d_perp, d_perp);
        // the reference writes `perp2 - 1.0` with a double literal, so on a double-capable OpenCL
        // device the product and the difference are formed in binary64 and rounded once; the
        // discriminant is ill-conditioned for distant rays, so this is kept
        float disc = (float)((double)(half_b*half_b) - (double)d_perp2*((double)o_perp2 - 1.0));
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
