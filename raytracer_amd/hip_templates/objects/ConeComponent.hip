### extends 'tracer.hip'
{#- semi-infinite cone with its tip at the origin, opening along +axis
    (reference objects/ConeComponent.cl:3-73) -#}

### macro tracer_function(obj)
    ### call tracer_component_function_base(obj)
        const float s2 = slope*slope;
        float o_ax = dot(origin, axis);
        float d_ax = dot(ray, axis);

        vec3 o_par = o_ax*axis;
        vec3 o_perp = origin - o_par;
        vec3 d_par = d_ax*axis;
        vec3 d_perp = ray - d_par;
        float d_perp2 = dot(d_perp, d_perp);
        float o_perp2 = dot(o_perp, o_perp);

        float a = d_perp2 - s2*d_ax*d_ax;
        float hb = dot(o_perp, d_perp) - s2 * d_ax*o_ax;
        float c = o_perp2 - s2 * o_ax*o_ax;

        float disc = hb*hb - a*c;
        if (disc < 0) {
            *t_enter = 1;
            *t_exit = 0;
            return;
        }

        float root = native_sqrt(disc);
        float t_near, t_far, dist = (-hb - root)/a;
        if (a >= 0) {
            t_near = dist;
            t_far = dist + 2*root/a;
        }
        else {
            t_far = dist;
            t_near = dist + 2*root/a;
        }

        float ax_near = o_ax + d_ax * t_near;
        float ax_far = o_ax + d_ax * t_far;

        if (ax_near < 0 && ax_far < 0) {
            // both roots on the mirror cone
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
        float o_ax = dot(pos, axis);
        vec3 o_perp = pos - o_ax*axis;
        *p_normal = fast_normalize(o_perp / (slope * o_ax) - axis * slope);
    ### endcall
### endmacro
