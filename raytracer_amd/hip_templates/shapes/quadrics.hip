{#- Quadric and planar shapes in tracer coordinates: the two stand-alone primitives (unit sphere,
    half-space x <= 0) and the components with which ConvexIntersection clips a ray interval
    [t_enter, t_exit].  One macro pair per class: <Class>__tracer(obj) / <Class>__normal(obj);
    the axis-aligned "Z" classes share their bodies with the general ones through the helper
    macros below.  Arithmetic follows the reference's cl_templates/objects/{Sphere, HalfSpace,
    SphereComponent, HalfSpaceComponent, LayerComponent, ZLayerComponent, CylinderComponent,
    ZCylinderComponent, ConeComponent, ZConeComponent}.cl operation by operation. -#}
### from 'tracer.hip' import tracer_function_base, normal_function_base, tracer_component_function_base

{#- ===== Sphere: |x| = 1 ================================================================ -#}
### macro Sphere__tracer(obj)
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

### macro Sphere__normal(obj)
    ### call normal_function_base(obj)
        *p_normal = pos;
    ### endcall
### endmacro

{#- ===== HalfSpace: x <= 0 ============================================================== -#}
### macro HalfSpace__tracer(obj)
    ### call tracer_function_base(obj)
        const float dist = -origin.x/ray.x;
        if (dist > 0) *t_hit = dist;
    ### endcall
### endmacro

### macro HalfSpace__normal(obj)
    ### call normal_function_base(obj)
        *p_normal = make_vec3(1.0f, 0.0f, 0.0f);
    ### endcall
### endmacro
