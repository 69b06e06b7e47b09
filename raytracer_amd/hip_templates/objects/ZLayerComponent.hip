### extends 'tracer.hip'
{#- slab 0 <= z <= 1 (reference objects/ZLayerComponent.cl:3-33) -#}

### macro tracer_function(obj)
    ### call tracer_component_function_base(obj)
        float slope = ray.z, d1, d2;
        if (slope > 0) {
            d1 = -origin.z;
            d2 = d1 + 1.0f;
        }
        else {
            d2 = -origin.z;
            d1 = d2 + 1.0f;
        }
        if ((slope > 0) == inside) *p_subobject = 1;
        else *p_subobject = 0;
        *t_enter = d1 / slope;
        *t_exit = d2 / slope;
    ### endcall
### endmacro

### macro normal_function(obj)
    ### call normal_function_base(obj)
        if (subobject == 1) *p_normal = make_vec3(0.0f, 0.0f, 1.0f);
        else *p_normal = make_vec3(0.0f, 0.0f, -1.0f);
    ### endcall
### endmacro
