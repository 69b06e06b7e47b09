### extends 'tracer.hip'
{#- triangle mesh, every face tested (reference objects/TriangleMesh.cl:7-115).
    `triangle_loop` is shared with the octree: the caller supplies the expression that maps
    the loop counter j to a face index.  Moeller-Trumbore without epsilon, two-sided. -#}

### macro triangle_loop()
    for (uint j = 0; j < n_triangles; ++j) {
        const int face_index = {{ caller('j') }};
        if (prev_face == (uint)face_index && on_surface) continue;

        const int ia = integer_data[3*face_index];
        const int ib = integer_data[3*face_index+1];
        const int ic = integer_data[3*face_index+2];
        const vec3 va = make_vec3(vector_data[ia]);
        const vec3 vb = make_vec3(vector_data[ib]);
        const vec3 vc = make_vec3(vector_data[ic]);
        const vec3 e1 = vb - va;
        const vec3 e2 = vc - va;

        const vec3 pvec = cross(ray, e2);
        const float det = dot(e1, pvec);
        const float inv_det = 1.f / det;
        const vec3 tvec = origin - va;
        const float u = dot(tvec, pvec) * inv_det;
        if (u < 0.f || u > 1.f) continue;
        const vec3 qvec = cross(tvec, e1);
        const float v = dot(ray, qvec) * inv_det;
        if (v < 0.f || u + v > 1.f) continue;
        const float t = dot(e2, qvec) * inv_det;
        if (t > 0.0f && t < best_t) {
            best_t = t;
            face = face_index;
        }
    }
### endmacro

### macro tracer_function(obj)
    ### call tracer_function_base(obj)
        uint prev_face = *p_subobject, face = 0;
        float best_t = t_limit;
        ### call(j) triangle_loop()
            {{ j }}
        ### endcall
        if (best_t < t_limit) {
            *t_hit = best_t;
            *p_subobject = face;
        }
    ### endcall
### endmacro

### macro normal_function(obj)
    ### call normal_function_base(obj)
        const int ia = integer_data[3*subobject],
                  ib = integer_data[3*subobject+1],
                  ic = integer_data[3*subobject+2];
        const vec3 va = make_vec3(vector_data[ia]);
        const vec3 vb = make_vec3(vector_data[ib]);
        const vec3 vc = make_vec3(vector_data[ic]);
        const vec3 e1 = vb - va;
        const vec3 e2 = vc - va;
        vec3 normal = cross(e1, e2);
        ### if obj.shading != 'flat'
            // barycentric blend of the vertex normals stored after the vertices
            const float4 * __restrict__ normal_data = vector_data + n_vertices;
            const vec3 f1 = va - pos;
            const vec3 f2 = vb - pos;
            const float normal_length = length(normal);
            const vec3 tri_normal = normal / length(normal);
            const float wc = dot(tri_normal, cross(e1, f1)) / normal_length;
            const float wb = dot(tri_normal, cross(f1, e2)) / normal_length;
            const float wa = dot(tri_normal, cross(vc - vb, f2)) / normal_length;
            normal = make_vec3(normal_data[ia])*wa + make_vec3(normal_data[ib])*wb + make_vec3(normal_data[ic])*wc;
        ### endif
        *p_normal = fast_normalize(normal);
    ### endcall
### endmacro
