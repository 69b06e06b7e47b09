### extends 'tracer.hip'
### import 'objects/TriangleMesh.hip' as tri_mesh
{#- octree traversal in the reference's serialised integer stream
    (reference objects/Octree.cl:32-168): clip the ray to the root cube, locate the leaf under
    the entry point, test its triangles up to the leaf's exit distance, hop to the neighbour
    cell through integer cell coordinates, climbing while the hop leaves the parent.
    NOTE: this is the parity layout; the flattened HBM layout for large meshes is separate. -#}

### macro tracer_function(obj)
    ### call tracer_function_base(obj)
        const vec3 slopes = make_vec3(1.0f / ray.x, 1.0f / ray.y, 1.0f / ray.z);
        const int nx = slopes.x < 0.0f, ny = slopes.y < 0.0f, nz = slopes.z < 0.0f;   // entry walls
        const vec3 far_walls = make_vec3((float)(1 - nx), (float)(1 - ny), (float)(1 - nz));
        const int sx = 1 - 2*nx, sy = 1 - 2*ny, sz = 1 - 2*nz;

        const vec3 rel = root_origin - origin;
        vec3 t_in = slopes * (rel + make_vec3((float)nx, (float)ny, (float)nz)*root_size),
             t_out = slopes * (rel + far_walls*root_size);
        float seg_begin = fmaxf(fmaxf(t_in.x, t_in.y), t_in.z);
        float seg_end = fminf(fminf(t_out.x, t_out.y), t_out.z);
        if (seg_begin > seg_end || seg_end < 0.0f || seg_begin > t_limit) return;

        const int * __restrict__ octree_data = integer_data + face_data_length;
        #define CLR_OCTREE_LEVELS {{ obj.__class__.MAX_DEPTH + 1 }}
        #define CLR_OCTREE_MAX_STEPS 300

        int cx = 0, cy = 0, cz = 0;       // integer cell coordinates at the current depth
        int bx = 0, by = 0, bz = 0;       // their lowest bits (position inside the parent)
        int path[CLR_OCTREE_LEVELS];
        path[0] = root_data_offset;
        uint depth = 0;
        uint prev_face = *p_subobject, face = 0;
        float node_size = root_size;

        for (int step = 0; step < CLR_OCTREE_MAX_STEPS; step++) {
            vec3 ray_pos = origin + seg_begin * ray, node_origin = make_vec3(0.0f, 0.0f, 0.0f);
            int child_mask = 0, data_ptr = 0, child_idx;

            while (depth < CLR_OCTREE_LEVELS-1) {
                child_mask = octree_data[path[depth]];
                data_ptr = octree_data[path[depth]+1];
                node_origin = root_origin + make_vec3((float)cx, (float)cy, (float)cz)*node_size;
                if (child_mask == 0 || child_mask == 0x100) break;     // leaf / empty leaf
                const vec3 rc = ray_pos - node_origin;
                const float half = 0.5f*node_size;
                bx = 1 - (rc.x < half); by = 1 - (rc.y < half); bz = 1 - (rc.z < half);
                child_idx = (bx > 0 ? 4 : 0) + (by > 0 ? 2 : 0) + (bz > 0 ? 1 : 0);
                path[depth+1] = data_ptr + child_idx*2;
                cx = (cx << 1) | bx; cy = (cy << 1) | by; cz = (cz << 1) | bz;
                node_size *= 0.5f;
                depth++;
            }

            t_out = slopes * (node_origin - origin + far_walls*node_size);
            seg_end = fminf(fminf(t_out.x, t_out.y), t_out.z);

            if (child_mask != 0x100) {
                const uint n_triangles = (uint)max(octree_data[data_ptr], 0);
                if (n_triangles > 0) {
                    float best_t = seg_end;
                    ### call(j) tri_mesh.triangle_loop()
                        octree_data[data_ptr + 1 + {{ j }}]
                    ### endcall
                    if (best_t < seg_end) {
                        *t_hit = best_t;
                        *p_subobject = face;
                        return;
                    }
                }
            }

            int hx = 0, hy = 0, hz = 0;
            if (seg_end == t_out.x) hx = sx;
            else if (seg_end == t_out.y) hy = sy;
            else if (seg_end == t_out.z) hz = sz;

            while (depth > 0) {
                bx += hx; by += hy; bz += hz;
                if (bx >= 0 && bx <= 1 && by >= 0 && by <= 1 && bz >= 0 && bz <= 1) break;
                depth--;
                cx >>= 1; cy >>= 1; cz >>= 1;
                bx = cx & 1; by = cy & 1; bz = cz & 1;
                node_size *= 2.0f;
            }
            if (depth == 0) return;        // left the root cube

            cx = (cx & ~1) | bx; cy = (cy & ~1) | by; cz = (cz & ~1) | bz;
            child_idx = (bx > 0 ? 4 : 0) + (by > 0 ? 2 : 0) + (bz > 0 ? 1 : 0);
            data_ptr = octree_data[path[depth-1]+1];
            path[depth] = data_ptr + child_idx*2;

            seg_begin = seg_end;
            if (seg_begin > t_limit) return;
        }
        #undef CLR_OCTREE_LEVELS
        #undef CLR_OCTREE_MAX_STEPS
    ### endcall
### endmacro

### macro normal_function(obj)
    {{ tri_mesh.normal_function(obj) }}
### endmacro
