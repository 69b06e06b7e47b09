{#- clip [lo_var, hi_var] to a bounding sphere of radius R about the origin; `return`s from
    the enclosing tracer function on a miss (reference objects/bounding_volumes.cl:3-27) -#}
### macro sphere_bounding_volume(R, lo_var, hi_var)
    {
        const float bound_r2 = {{ (R*R)|float }}f;
        float half_b = -dot(ray, origin);
        float o_len2 = dot(origin, origin);
        bool inside_bound = o_len2 < bound_r2;
        if (half_b <= 0 && !inside_bound) return;
        const float disc = half_b*half_b - o_len2 + bound_r2;
        if (disc < 0) return;
        const float root = native_sqrt(disc);
        {{ lo_var }} = fmaxf(half_b - root, 0.0f);
        {{ hi_var }} = fminf(half_b + root, t_limit);
        if ({{ hi_var }} <= {{ lo_var }}) return;
    }
### endmacro
