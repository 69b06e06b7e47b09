{%- macro vec3_literal(p) -%}
make_vec3({{p[0]}}, {{p[1]}}, {{p[2]}})
{%- endmacro %}
