{#- Convex bodies: the intersection of convex components, unrolled over the component list at
    code-generation time, and the components themselves (sphere, half space, slabs, cylinders,
    cones).  Role of reference cl_templates/objects/ConvexIntersection.cl and the
    *Component.cl files next to it.

    Every component clips the parametric ray interval [lo, hi] and reports which of its faces
    did the clipping; face ids of the whole body are handed out host-side by
    ConvexIntersection.face_ranges() (first, end) so that the generated code only contains
    literals.  The hit is lo when the ray starts outside the body and hi when it starts inside. -#}
### from 'tracer.hip' import tracer_function_base, normal_function_base, tracer_component_function_base

### macro ConvexIntersection__tracer(obj)
    ### call tracer_function_base(obj)
        vec3 rel = origin;
        float lo = 0.0f, hi = t_limit;
        float c_enter, c_exit;
        uint face = 0, c_face, prev_face = *p_subobject;

        ### for idx, component, first, end in obj.face_ranges()
        {   // component {{ idx }}: {{ component.__class__.__name__ }}, faces {{ first + 1 }}..{{ end }}
            c_face = 0;
            c_enter = lo;
            c_exit = hi;
            if (on_surface && prev_face > {{ first }})
                c_face = prev_face - {{ first }};

            {{ component.tracer_function_name }}(
                {{ obj.component_tracer_call_params(idx) }}, clr_g
            );

            if (c_enter > lo) {
                lo = c_enter;
                if (!inside) face = c_face + {{ first }};
            }
            if (c_exit < hi) {
                hi = c_exit;
                if (inside) face = c_face + {{ first }};
            }
            if (lo > hi || lo > t_limit) return;
        }
        ### endfor

        *t_hit = inside ? hi : lo;
        *p_subobject = face;
    ### endcall
### endmacro

### macro ConvexIntersection__normal(obj)
    ### call normal_function_base(obj)
        const vec3 p = pos;
        ### for idx, component, first, end in obj.face_ranges()
            {% if not loop.first %}else {% endif %}if ({% if first > 0 %}subobject >= {{ first }} && {% endif %}subobject < {{ end }})
                {{ component.normal_function_name }}(
                    {{ obj.component_normal_call_params(idx, first) }}
                );
        ### endfor
    ### endcall
### endmacro

{#- ===== SphereComponent: |x| <= R ====================================================== -#}
### macro SphereComponent__tracer(obj)
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

### macro SphereComponent__normal(obj)
    ### call normal_function_base(obj)
        *p_normal = pos / R;
    ### endcall
### endmacro

{#- ===== HalfSpaceComponent: normal . x <= h ============================================ -#}
### macro HalfSpaceComponent__tracer(obj)
    ### call tracer_component_function_base(obj)
        const int zero_mask = clr_zero_mask(normal);
        float slope = clr_dot_lit(ray, normal, zero_mask, clr_g);
        float dist = clr_dot_lit(-origin, normal, zero_mask, clr_g) + h;
        dist = dist/slope;
        if (slope < 0) *t_enter = dist;
        else *t_exit = dist;
        *p_subobject = 0;
    ### endcall
### endmacro

### macro HalfSpaceComponent__normal(obj)
    ### call normal_function_base(obj)
        *p_normal = normal;
    ### endcall
### endmacro

{#- ===== slabs: sub-object 0 / 1 = which face ============================================
    slope_expr: ray component along the slab normal, base_expr: -origin along it -#}
### macro slab_body(slope_expr, base_expr, thickness)
        float slope = {{ slope_expr }}, d1, d2;
        if (slope > 0) {
            d1 = {{ base_expr }};
            d2 = d1 + {{ thickness }};
        }
        else {
            d2 = {{ base_expr }};
            d1 = d2 + {{ thickness }};
        }
        if ((slope > 0) == inside) *p_subobject = 1;
        else *p_subobject = 0;
        *t_enter = d1 / slope;
        *t_exit = d2 / slope;
### endmacro

### macro LayerComponent__tracer(obj)
    ### call tracer_component_function_base(obj)
        const int zero_mask = clr_zero_mask(normal);
        {{ slab_body('clr_dot_lit(ray, normal, zero_mask, clr_g)', 'clr_dot_lit(-origin, normal, zero_mask, clr_g)', 'h') }}
    ### endcall
### endmacro

### macro LayerComponent__normal(obj)
    ### call normal_function_base(obj)
        if (subobject == 1) *p_normal = normal;
        else *p_normal = -normal;
    ### endcall
### endmacro

### macro ZLayerComponent__tracer(obj)
    ### call tracer_component_function_base(obj)
        {{ slab_body('ray.z', '-origin.z', '1.0f') }}
    ### endcall
### endmacro

### macro ZLayerComponent__normal(obj)
    ### call normal_function_base(obj)
        if (subobject == 1) *p_normal = make_vec3(0.0f, 0.0f, 1.0f);
        else *p_normal = make_vec3(0.0f, 0.0f, -1.0f);
    ### endcall
### endmacro

{#- ===== infinite cylinders ==============================================================
    given half_b = d_perp . o_perp, o_perp2, d_perp2 and the discriminant expression -#}
### macro cylinder_roots(disc_expr)
        float disc = {{ disc_expr }};
        if (disc < 0) {
            *t_enter = 1;
            *t_exit = 0;
            return;
        }
        float root = native_sqrt(disc);
        float d1 = -half_b - root;
        *t_enter = d1 / d_perp2;
        *t_exit = (d1 + 2*root) / d_perp2;
### endmacro

### macro CylinderComponent__tracer(obj)
    ### call tracer_component_function_base(obj)
        const int zero_mask = clr_zero_mask(axis);
        float o_ax = clr_dot_lit(origin, axis, zero_mask, clr_g), d_ax = clr_dot_lit(ray, axis, zero_mask, clr_g);
        vec3 o_perp = clr_reject_lit(origin, o_ax, axis, zero_mask, clr_g);
        vec3 d_perp = clr_reject_lit(ray, d_ax, axis, zero_mask, clr_g);
        float half_b = dot(d_perp, o_perp);
        float o_perp2 = dot(o_perp, o_perp);
        float d_perp2 = dot(d_perp, d_perp);
        {{ cylinder_roots('half_b*half_b - d_perp2*(o_perp2 - R*R)') }}
    ### endcall
### endmacro

### macro CylinderComponent__normal(obj)
    ### call normal_function_base(obj)
        vec3 o_perp = pos - dot(pos, axis)*axis;
        *p_normal = o_perp / R;
    ### endcall
### endmacro

### macro ZCylinderComponent__tracer(obj)
    ### call tracer_component_function_base(obj)
        vec2 o_perp = make_vec2(origin.x, origin.y);
        vec2 d_perp = make_vec2(ray.x, ray.y);
        float half_b = dot(d_perp, o_perp);
        float o_perp2 = dot(o_perp, o_perp);
        float d_perp2 = dot(d_perp, d_perp);
        // the reference writes `perp2 - 1.0` with a double literal, so on a double-capable OpenCL
        // device the product and the difference are formed in binary64 and rounded once; the
        // discriminant is ill-conditioned for distant rays, so this is kept
        {{ cylinder_roots('(float)((double)(half_b*half_b) - (double)d_perp2*((double)o_perp2 - 1.0))') }}
    ### endcall
### endmacro

### macro ZCylinderComponent__normal(obj)
    ### call normal_function_base(obj)
        vec3 o_perp = pos;
        o_perp.z = 0;
        *p_normal = o_perp;
    ### endcall
### endmacro

{#- ===== semi-infinite cones, tip at the origin, opening along the positive axis =========
    given o_ax, d_ax (origin / ray along the axis) and the quadratic a, hb, c -#}
### macro cone_interval()
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
            // both roots lie on the mirror cone
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
### endmacro

### macro ConeComponent__tracer(obj)
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
        {{ cone_interval() }}
    ### endcall
### endmacro

### macro ConeComponent__normal(obj)
    ### call normal_function_base(obj)
        float o_ax = dot(pos, axis);
        vec3 o_perp = pos - o_ax*axis;
        *p_normal = fast_normalize(o_perp / (slope * o_ax) - axis * slope);
    ### endcall
### endmacro

### macro ZConeComponent__tracer(obj)
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
        {{ cone_interval() }}
    ### endcall
### endmacro

### macro ZConeComponent__normal(obj)
    ### call normal_function_base(obj)
        vec3 nn = pos / pos.z;
        nn.z = -1.0f;
        *p_normal = nn;
    ### endcall
### endmacro
