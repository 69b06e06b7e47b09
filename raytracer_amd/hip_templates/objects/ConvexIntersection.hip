### extends 'tracer.hip'
{#- intersection of convex components, unrolled over the component list at code-generation
    time (reference objects/ConvexIntersection.cl:3-70).  Each component clips the ray
    interval [lo, hi]; the hit is lo from outside, hi from inside. -#}

### macro tracer_function(obj)
    ### call tracer_function_base(obj)
        vec3 rel = origin;
        float lo = 0.0f, hi = t_limit;
        float c_enter, c_exit;
        uint face = 0, c_face, prev_face = *p_subobject;

        ### set ns = namespace(offset=0)
        ### for component in obj.components
            c_face = 0;
            c_enter = lo;
            c_exit = hi;

            if (on_surface && prev_face > {{ ns.offset }})
                c_face = prev_face - {{ ns.offset }};

            {{ component.tracer_function_name }}(
                {{ obj.component_tracer_call_params(loop.index0) }}
            );

            if (c_enter > lo) {
                lo = c_enter;
                if (!inside) face = c_face + {{ ns.offset }};
            }
            if (c_exit < hi) {
                hi = c_exit;
                if (inside) face = c_face + {{ ns.offset }};
            }
            if (lo > hi || lo > t_limit) return;

            ### set ns.offset = ns.offset + component.n_subobjects
        ### endfor

        if (inside) *t_hit = hi;
        else *t_hit = lo;
        *p_subobject = face;
    ### endcall
### endmacro

### macro normal_function(obj)
    ### call normal_function_base(obj)
        const vec3 p = pos;
        ### set ns = namespace(offset=0)
        ### for component in obj.components
            {% if ns.offset > 0 %}else {% endif %}if ({% if ns.offset > 0 %}subobject >= {{ ns.offset }} && {% endif %}subobject < {{ ns.offset + component.n_subobjects }})
                {{ component.normal_function_name }}(
                    {{ obj.component_normal_call_params(loop.index0, ns.offset) }}
                );
            ### set ns.offset = ns.offset + component.n_subobjects
        ### endfor
    ### endcall
### endmacro
