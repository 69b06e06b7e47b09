{#- Scaffolding shared by all shape templates: function signatures, parameter fetch and the
    three per-class step functions (closest hit, shadow test, normal).  Counterpart of the
    reference's tracer.cl (/root/reference/cl_templates/tracer.cl:2-284); there the three
    steps are separate kernels launched per object / per class, here they are __device__
    functions inlined into the fused bounce kernel (main.hip). -#}

### macro normal_function_base(obj)
{#- exterior normal at a surface point, in tracer coordinates #}
CLR_FN void {{ obj.normal_function_name }}(
        const vec3 pos,              // a point on the surface
        const uint subobject,        // sub-object number computed by the tracer
        vec3 *p_normal               // [out]
        {{ obj.parameter_declaration_string() }}) {
{{ caller() }}
}
### endmacro

### macro tracer_function_base(obj)
{#- distance to the nearest valid intersection of a ray with the shape, tracer coordinates #}
template<class CLR_G> {{ 'CLR_OUTLINED' if obj.outlined else 'CLR_FN' }} void {{ obj.tracer_function_name }}(
        const vec3 origin,
        const vec3 ray,
        const float t_limit,   // upper bound for the distance
        float *t_hit,      // [out] stays 0 when there is no hit
        uint *p_subobject,           // [in/out] e.g. which face
        bool inside,                 // ray travels inside this object
        bool on_surface             // ray starts on this object's surface
        {{ obj.parameter_declaration_string() }},
        CLR_G &clr_g) {              // operand-range guard of native_sqrt (clr_exact.inl)
{{ caller() }}
}
### endmacro

### macro tracer_component_function_base(obj)
template<class CLR_G> CLR_FN void {{ obj.tracer_function_name }}(
        const vec3 origin,
        const vec3 ray,
        float *t_enter,         // [in/out] interval of the ray inside the component
        float *t_exit,
        uint *p_subobject,
        bool inside
        {{ obj.parameter_declaration_string() }},
        CLR_G &clr_g) {
{{ caller() }}
}
### endmacro

{#- actual arguments for the declared parameters of obj, read from the parameter table P #}
### macro tracer_params(obj)
    ### if obj.has_data()
        , vector_data + clr_data_offsets[2*(object_id-1)]
        , integer_data + clr_data_offsets[2*(object_id-1)+1]
    ### endif
    ### for cl_type in obj.parameter_types()
        ### if cl_type == 'float3'
        , make_vec3(P[{{ loop.index0 }}])
        ### elif cl_type == 'int'
        , __float_as_int(P[{{ loop.index0 }}].x)
        ### else
        , P[{{ loop.index0 }}].x
        ### endif
    ### endfor
### endmacro

### macro tracer_kernel(obj)
// closest-hit step for one object of this class (tracer.cl:72-141)
template<class CLR_G> CLR_FN void {{ obj.tracer_kernel_name }}(
    const vec3 pos, const vec3 ray, float &isec_dist, uint &whichobject, uint &which_subobject,
    const uint last_whichobject, const uint last_which_subobject, const uint inside,
    const float4 * __restrict__ vector_data, const int * __restrict__ integer_data,
    const int object_id, const float4 *P, const int affine_skip, const clr_direction &world_dir, CLR_G &clr_g)
{
    const bool within = inside == (uint)object_id,
               on_surface = last_whichobject == (uint)object_id;
    ### if obj.convex
    if (!on_surface || within) {
    ### endif
        float new_isec_dist = 0;
        uint cur_subobject = last_which_subobject;
        const float4 *p_inverse_affine = P + {{ obj.parameter_types()|length }};

        const vec3 local_pos = clr_to_local_position(p_inverse_affine, pos, affine_skip, clr_g);
        const vec3 local_ray = clr_to_local_direction(p_inverse_affine, ray, affine_skip, clr_g);
    ### if obj.can_hit_expression
        // sign test of the class: where it fails the ray test below is known to report no hit
        if ({{ obj.can_hit_expression }}) {
    ### endif
        // unit direction, stretch = |local_ray|; the four quotients by stretch share one reciprocal
        const clr_direction dir = clr_direction_of(local_ray, world_dir, affine_skip, clr_g);

        {{ obj.tracer_function_name }}(
            local_pos, dir.unit, isec_dist*dir.len, &new_isec_dist, &cur_subobject,
            within, on_surface
            {{ tracer_params(obj) }}
            , clr_g
        );

        clr_guard_upper(dir, new_isec_dist, clr_g);
        if (new_isec_dist > 0) {
            new_isec_dist = clr_over_length(new_isec_dist, dir, clr_g);
            if (new_isec_dist < isec_dist) {
                isec_dist = new_isec_dist;
                which_subobject = cur_subobject;
                whichobject = object_id;
            }
        }
    ### if obj.can_hit_expression
        }
    ### endif
    ### if obj.convex
    }
    ### endif
}
### endmacro

### macro shadow_kernel(obj)
// visibility step towards the light point for one object of this class (tracer.cl:143-221);
// `ray`/`isec_dist` = unit direction and distance to the light point, computed by the caller
template<class CLR_G> CLR_FN void {{ obj.shadow_kernel_name }}(
    const vec3 pos, const vec3 ray, const float isec_dist,
    const uint whichobject, const uint which_subobject, const uint inside, float &shadow_mask,
    const float4 * __restrict__ vector_data, const int * __restrict__ integer_data,
    const int object_id, const float4 *P, const int affine_skip, const clr_direction &world_dir, CLR_G &clr_g)
{
    uint subobject = which_subobject;
    const bool within = inside == (uint)object_id,
               on_surface = whichobject == (uint)object_id;
    float new_isec_dist = 0;
    ### if obj.convex
    if (!on_surface || within) {
    ### endif
        const float4 *p_inverse_affine = P + {{ obj.parameter_types()|length }};
        const vec3 local_pos = clr_to_local_position(p_inverse_affine, pos, affine_skip, clr_g);
        const vec3 local_ray = clr_to_local_direction(p_inverse_affine, ray, affine_skip, clr_g);
    ### if obj.can_hit_expression
        // sign test of the class: where it fails the ray test below is known to report no hit
        if ({{ obj.can_hit_expression }}) {
    ### endif
        // unit direction, stretch = |local_ray|; the four quotients by stretch share one reciprocal
        const clr_direction dir = clr_direction_of(local_ray, world_dir, affine_skip, clr_g);

        {{ obj.tracer_function_name }}(
            local_pos, dir.unit, isec_dist*dir.len, &new_isec_dist, &subobject,
            within, on_surface
            {{ tracer_params(obj) }}
            , clr_g
        );

        clr_guard_upper(dir, new_isec_dist, clr_g);
        if (new_isec_dist > 0) {
            new_isec_dist = clr_over_length(new_isec_dist, dir, clr_g);
            if (new_isec_dist < isec_dist) shadow_mask = 0.0f;
        }
    ### if obj.can_hit_expression
        }
    ### endif
    ### if obj.convex
    }
    ### endif
}
### endmacro

### macro normal_kernel(obj)
// world-space normal at `pos` on object `object_id` of this class (tracer.cl:223-284)
template<class CLR_G> CLR_FN vec3 {{ obj.normal_kernel_name }}(
    const vec3 pos, const vec3 ray, const uint subobject, const uint inside,
    const float4 * __restrict__ vector_data, const int * __restrict__ integer_data,
    const int object_id, const float4 *P, CLR_G &clr_g)
{
    const float4 *p_inverse_affine = P + {{ obj.parameter_types()|length }};
    const vec3 local_pos = apply_affine_transform(p_inverse_affine, pos);
    vec3 normal = make_vec3(0.0f, 0.0f, 0.0f);
    {{ obj.normal_function_name }}(
        local_pos, subobject, &normal
        {{ tracer_params(obj) }}
    );
    ### if obj.speculative_normal
    {   // v / |v| through the short sequences where the guard allows (clr_exact.inl): the same bits
        const clr_direction unit_normal = clr_make_direction(apply_transposed_linear_transform(p_inverse_affine, normal), clr_g);
        clr_guard_upper(unit_normal, 0.0f, clr_g);
        normal = unit_normal.unit;
    }
    ### else
    normal = normalize(apply_transposed_linear_transform(p_inverse_affine, normal));
    ### endif
    ### if obj.auto_flip_normal
        if (dot(normal, ray) > 0) normal = -normal;
    ### else
        if (inside == (uint)object_id) normal = -normal;
    ### endif
    return normal;
}
### endmacro
