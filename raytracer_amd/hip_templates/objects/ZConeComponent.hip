### extends 'tracer.hip'
{#- unit-slope cone about +z with its tip at the origin (reference objects/ZConeComponent.cl:3-70) -#}

### macro tracer_function(obj)
    ### call tracer_component_function_base(obj)
        const float o_ax = origin.z;
        const float d_ax = ray.z;
        const vec2 o_perp = make_vec2(origin.x, origin.y);
        const vec2 d_perp = make_vec2(ray.x, ray.y);
        const float d_perp2 = dot(d_perp, d_perp);
        const float o_perp2 = dot(o_perp, o_perp);

        const float a = d_perp2 - d_ax*d_ax;
        const float hb = dot(o_perp, d_perp) - d_ax*o_ax;
        const float c = o_perp2 - o_ax*o_ax;

        const float disc = hb*hb - a*c;
        if (disc < 0) {
            *t_enter = 1;
            *t_exit = 0;
            return;
        }

        const float root = native_sqrt(disc);
        float t_near, t_far, dist = (-hb - root)/a;
        if (a >= 0) {
            t_near = dist;
            t_far = dist + 2*root/a;
        }
        else {
            t_far = dist;
            t_near = dist + 2*root/a;
        }

        const float ax_near = o_ax + d_ax * t_near;
        const float ax_far = o_ax + d_ax * t_far;

        if (ax_near < 0 && ax_far < 0) {
            *t_enter = 1;
            *t_exit = 0;
        }
        else {
            if (ax_near < 0) *t_enter = t_far;
            else if (ax_far < 0) *t_exit = t_near;
            else {
                *t_enter = t_near;
                *t_exit = t_far;
            }
        }
    ### endcall
### endmacro

### macro normal_function(obj)
    ### call normal_function_base(obj)
        vec3 nn = pos / pos.z;
        nn.z = -1.0f;
        *p_normal = nn;
    ### endcall
### endmacro
