{#- Triangle meshes: brute force over all faces (TriangleMesh) and octree traversal (Octree) in
    the reference's stream layout or the flattened layout.  Macro pairs <Class>__tracer /
    <Class>__normal.  Reference counterparts: cl_templates/objects/{TriangleMesh, Octree}.cl -#}
### from 'tracer.hip' import tracer_function_base, normal_function_base

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

### macro TriangleMesh__tracer(obj)
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

### macro TriangleMesh__normal(obj)
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

{#- octree traversal (reference objects/Octree.cl:32-168): clip the ray to the root cube,
    locate the leaf under the entry point, test its triangles up to the leaf's exit distance,
    hop to the neighbour cell through integer cell coordinates, climbing while the hop leaves
    the parent.  Two data layouts, same visiting order and same arithmetic (bit-identical hits):
      stream: the reference's integer stream (header = [child mask, payload offset]);
      flat  : int2 node records with the 8 children of a node contiguous, and per-leaf
              pre-gathered triangle records (v0|face id, e1, e2 = 48 B, one coalesced run per
              leaf) instead of face index -> 3 vertex indices -> 3 gathered vertices (60 B).
    The per-lane stack of node positions lives in LDS (one column per lane: conflict-free),
    not in scratch memory as a dynamically indexed private array would. -#}

{#- flat layout: 36-byte records (v0, e1, e2) back to back, 4-byte aligned; face ids apart -#}
### macro flat_record_types()
        typedef float clr_f4u __attribute__((ext_vector_type(4), aligned(4)));
        typedef int clr_i2u __attribute__((ext_vector_type(2), aligned(4)));
### endmacro

### macro flat_triangle_loop()
    // the record of triangle j+1 is requested before triangle j is tested: the traversal is
    // bound by memory latency, not by arithmetic
    const float * __restrict__ rec = tris + 9*first_triangle;
    clr_f4u n0 = *(const clr_f4u *)rec, n1 = *(const clr_f4u *)(rec + 4);
    float n2 = rec[8];
    for (int j = 0; j < n_leaf_triangles; ++j) {
        const clr_f4u r0 = n0, r1 = n1;
        const float r2 = n2;
        if (j + 1 < n_leaf_triangles) {
            n0 = *(const clr_f4u *)(rec + 9*j + 9); n1 = *(const clr_f4u *)(rec + 9*j + 13); n2 = rec[9*j + 17];
        }
        if (on_surface && prev_face == (uint)tri_ids[first_triangle + j]) continue;
        const vec3 va = make_vec3(r0.x, r0.y, r0.z);
        const vec3 e1 = make_vec3(r0.w, r1.x, r1.y);
        const vec3 e2 = make_vec3(r1.z, r1.w, r2);
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
            face = tri_ids[first_triangle + j];
        }
    }
### endmacro

{#- Flat layout with pooled triangle tests.  Measured on the 1 M-triangle scene, 93 % of the
    traversal time is triangle tests and only a quarter of the lanes of a wave are busy in them
    (rays that missed the root cube or already found their hit idle, leaves hold 1..50 faces).
    Here every step has three phases:
      A  each lane still traversing locates its leaf: (first record, count), exit distance;
      B  the (lane, triangle) pairs of the whole wave form one list (prefix sum of the counts
         over the lanes executing this call, taken from ballots of their bit planes and kept in
         LDS by rank) that is dealt out evenly:
         every lane - also one that has finished its own traversal - takes list entry
         round*A + rank, finds the owner by binary search in the prefix sums, reads the owner's
         ray from LDS and tests the triangle with the arithmetic of flat_triangle_loop; a hit
         goes to the owner with a 64-bit LDS minimum on (t bits, record number), which is
         the sequential rule "smallest t, earlier triangle on ties";
      C  each lane reads its own minimum and either returns the hit or hops to the next cell.
    A lane never leaves the loop before all lanes of the call are done. -#}
### macro Octree_pooled__tracer(obj)
    ### call tracer_function_base(obj)
        const int LEVELS = {{ obj.levels }};
        const int MAX_STEPS = {{ obj.max_steps }};
        #define POOL_WORDS 65                          // 64 words of item bits + one, so that a window may reach past the last item
### if not obj.ropes
        __shared__ int stack_lds[{{ obj.levels }} * 256];
### endif
        __shared__ float4 pool_origin[256];            // origin.xyz, exit distance of the current leaf
        __shared__ float4 pool_ray[256];               // ray.xyz, face to skip (int bits; -1 = none)
        __shared__ int2 pool_owner[256];               // q-th lane of the wave that stands in a leaf with triangles:
                                                       // (thread index, first record of the leaf - first item of the lane)
        __shared__ unsigned long long pool_ends[4 * POOL_WORDS];   // per wave: bit w set = item w is the last one of its owner
        __shared__ unsigned long long pool_best[256];  // per lane: min of (t bits << 32 | index in leaf)
### if not obj.ropes
        int *stack = stack_lds + threadIdx.x;
        #define STACK(level) stack[(level) * 256]
### endif
        #define POOL_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier()
        ### if obj.profile_phases
        // development build (CLR_MESH_PROFILE=1, tools/mesh_phases.py): cycles per phase, kept per
        // lane during the call and added by one lane per wave to one of 64 copies of the counters
        #define PHASE_CLOCK() ((long long)__builtin_readcyclecounter())
        long long phase_sum[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        #define PHASE_ADD(slot, value) phase_sum[slot] += (long long)(value)
        long long phase_mark = PHASE_CLOCK();
        #define PHASE_END(slot) { const long long now = PHASE_CLOCK(); phase_sum[slot] += now - phase_mark; phase_mark = now; }
        // parts of a phase (slots 10-13), also inside code that only some lanes run: every lane that
        // runs it keeps the sum, the lane that walks longest ran all of it (largest sum of the wave)
        long long split_mark = 0;
        __shared__ unsigned long long phase_split[4 * 4];
        #define SPLIT_BEGIN() split_mark = PHASE_CLOCK()
        #define SPLIT_END(slot) { const long long now = PHASE_CLOCK(); phase_sum[slot] += now - split_mark; split_mark = now; }
        ### else
        #define PHASE_ADD(slot, value)
        #define PHASE_END(slot)
        #define SPLIT_BEGIN()
        #define SPLIT_END(slot)
        ### endif

        const int tid = threadIdx.x, wave_base = tid & ~63;
        const unsigned long long exec_mask = __ballot(1);
        const unsigned long long lanes_before = (1ull << (tid & 63)) - 1ull;
        const int rank = __popcll(exec_mask & lanes_before);
        const int n_lanes = __popcll(exec_mask);

        const vec3 slopes = make_vec3(1.0f / ray.x, 1.0f / ray.y, 1.0f / ray.z);
        const int nx = slopes.x < 0.0f, ny = slopes.y < 0.0f, nz = slopes.z < 0.0f;   // entry walls
        const vec3 far_walls = make_vec3((float)(1 - nx), (float)(1 - ny), (float)(1 - nz));
        const int sx = 1 - 2*nx, sy = 1 - 2*ny, sz = 1 - 2*nz;

        const vec3 rel = root_origin - origin;
        vec3 t_in = slopes * (rel + make_vec3((float)nx, (float)ny, (float)nz)*root_size),
             t_out = slopes * (rel + far_walls*root_size);
        float seg_begin = fmaxf(fmaxf(t_in.x, t_in.y), t_in.z);
        float seg_end = fminf(fminf(t_out.x, t_out.y), t_out.z);
        bool done = seg_begin > seg_end || seg_end < 0.0f || seg_begin > t_limit;

        const int * __restrict__ octree_data = integer_data + face_data_length;
        {{ flat_record_types() }}
        const float * __restrict__ tris = (const float *)(vector_data + flat_tri_offset);
        const int * __restrict__ tri_ids = octree_data + flat_ids_offset;

        const uint prev_face = *p_subobject;
        pool_ray[tid] = make_float4(ray.x, ray.y, ray.z, __int_as_float(on_surface ? (int)prev_face : -1));
        unsigned long long *ends = pool_ends + (tid >> 6) * POOL_WORDS;

        int cx = 0, cy = 0, cz = 0;       // integer cell coordinates at the current depth
        int bx = 0, by = 0, bz = 0;       // their lowest bits (position inside the parent)
### if obj.ropes
        // Ropes (no counterpart in the reference, which finds the next leaf by a descent from the
        // common ancestor, Octree.cl:120-160): per (node, exit face) one record (R | depth << 27, R's
        // node record), R = the deepest node beyond that face that is a leaf or as deep as the node
        // (objects/meshes.py, octree_ropes); per node its parent, for the rare hop that the
        // reference's arithmetic takes another way.
        const int * __restrict__ node_ropes = octree_data + flat_rope_offset;
        const int * __restrict__ node_parents = octree_data + flat_parent_offset;
        int node_at = root_data_offset;   // node whose record is in `ahead`
        int rope_target = -1;             // >= 0: the previous step hopped; R | depth of R << 27
        int hop_cell = 0;                 // depth << 27 | x << 18 | y << 9 | z of the cell beyond the exit face, at the depth of the leaf left
        int left_leaf = 0;                // the leaf left | (child number of S - child number of the ancestor + 8) << 27
        int hop_face = 0;                 // 2 axis + (towards larger coordinates) of that hop
        const float rope_margin = {{ obj.rope_margin_literal }};
### else
        STACK(0) = root_data_offset;
### endif
        int depth = 0;
        float node_size = root_size;

        // `ahead` = record of the node at STACK(depth), asked for one phase early: the root's here,
        // later the neighbour cell's right after the hop, which is taken before the triangle tests
        // so that the record travels while they run
        clr_i2u ahead = *(const clr_i2u *)(octree_data + 2*root_data_offset);

        PHASE_END(1)      // entry: root cube, LDS set-up
        for (int step = 0; step < MAX_STEPS && __any(!done); step++) {
            // ---- A: locate the leaf under the current point
            int count = 0, first = 0;
            bool last_cell = false;       // the ray leaves the tree (or its range) after this cell
            PHASE_ADD(5, 1); PHASE_ADD(6, __popcll(__ballot(!done)));
### if obj.ropes
            if (!done) {
                vec3 ray_pos = origin + seg_begin * ray, node_origin = make_vec3(0.0f, 0.0f, 0.0f);
                int head = 0, payload = 0, child_idx;
                clr_i2u record = ahead;
                if (rope_target >= 0) {
                    // The hop of the previous step left (depth, cx, cy, cz, node_size) at S, the
                    // sibling of an ancestor of the leaf just left, from where the reference
                    // descends again, one dependent record per level, choosing the child under
                    // ray_pos.  The leaf's rope names R, the deepest node on the way from S to the
                    // cell beyond the exit face (hop_cell, at the leaf's depth) that is a leaf or
                    // as deep as the leaf left.  The choices of that descent are arithmetic on
                    // registers: they are made here without the records, and if every one of them
                    // leads towards R (all but rays within a rounding error of a cell edge) the
                    // traversal continues at R with the record that came with the rope.
                    const int leaf_depth = hop_cell >> 27, hop_x = (hop_cell >> 18) & 511, hop_y = (hop_cell >> 9) & 511,
                              hop_z = hop_cell & 511, rope_depth = rope_target >> 27;
                    // Sure without looking at the levels one by one: the cells whose centres the
                    // descent compares ray_pos with all contain R, so a ray_pos inside R's cell by
                    // more than the rounding errors of those comparisons (rope_margin: 16 units in
                    // the last place of the tree's coordinates, the comparisons err by < 3) goes R's
                    // way at every level.  On the hop axis ray_pos lies on the wall it came through,
                    // which is an outer wall of S and at least R's size away from all those centres.
                    const float size_r = node_size * __int_as_float((127 - (rope_depth - depth)) << 23);
                    bool same_way;
                    {
                        const int up = leaf_depth - rope_depth;
                        const vec3 rc = ray_pos - (root_origin + make_vec3((float)(hop_x >> up), (float)(hop_y >> up), (float)(hop_z >> up))*size_r);
                        const int hop_axis = hop_face >> 1;
                        const float wall = (hop_face & 1) ? 0.0f : size_r, quarter = 0.25f*size_r;
                        const float lo_m = rope_margin, hi_m = size_r - rope_margin;
                        same_way = (hop_axis == 0 ? fabsf(rc.x - wall) < quarter : (rc.x > lo_m && rc.x < hi_m))
                                && (hop_axis == 1 ? fabsf(rc.y - wall) < quarter : (rc.y > lo_m && rc.y < hi_m))
                                && (hop_axis == 2 ? fabsf(rc.z - wall) < quarter : (rc.z > lo_m && rc.z < hi_m));
                    }
                    float size_j = same_way ? size_r : node_size;
                    // otherwise (a ray_pos within the margin of a wall of R) the comparisons themselves
                    for (int j = same_way ? rope_depth : depth; j < rope_depth; ++j) {
                        const int up = leaf_depth - j;
                        const vec3 cell_origin = root_origin + make_vec3((float)(hop_x >> up), (float)(hop_y >> up), (float)(hop_z >> up))*size_j;
                        const vec3 rc = ray_pos - cell_origin;
                        const float half = 0.5f*size_j;
                        same_way = same_way && (1 - (rc.x < half)) == ((hop_x >> (up - 1)) & 1)
                                            && (1 - (rc.y < half)) == ((hop_y >> (up - 1)) & 1)
                                            && (1 - (rc.z < half)) == ((hop_z >> (up - 1)) & 1);
                        size_j *= 0.5f;
                    }
                    if (same_way) {
                        const int up = leaf_depth - rope_depth;
                        depth = rope_depth;
                        cx = hop_x >> up; cy = hop_y >> up; cz = hop_z >> up;
                        bx = cx & 1; by = cy & 1; bz = cz & 1;
                        node_size = size_j;
                        node_at = rope_target & 0x7ffffff;
                    }
                    else {
                        // the reference's own way: S is a sibling of the leaf's ancestor at this depth
                        int at = left_leaf & 0x7ffffff;
                        for (int j = depth; j < leaf_depth; ++j) at = node_parents[at];
                        node_at = at + ((left_leaf >> 27) - 8);
                        record = *(const clr_i2u *)(octree_data + 2*node_at);
                    }
                }
                while (depth < LEVELS-1) {
                    head = record.x;                   // >= 0: leaf with that many triangles, -1: inner
                    payload = record.y;                // first triangle record / first child
                    node_origin = root_origin + make_vec3((float)cx, (float)cy, (float)cz)*node_size;
                    if (head >= 0) break;
                    const vec3 rc = ray_pos - node_origin;
                    const float half = 0.5f*node_size;
                    bx = 1 - (rc.x < half); by = 1 - (rc.y < half); bz = 1 - (rc.z < half);
                    child_idx = (bx > 0 ? 4 : 0) + (by > 0 ? 2 : 0) + (bz > 0 ? 1 : 0);
                    node_at = payload + child_idx;
                    cx = (cx << 1) | bx; cy = (cy << 1) | by; cz = (cz << 1) | bz;
                    node_size *= 0.5f;
                    depth++;
                    if (depth < LEVELS-1) record = *(const clr_i2u *)(octree_data + 2*node_at);
                }
                t_out = slopes * (node_origin - origin + far_walls*node_size);
                seg_end = fminf(fminf(t_out.x, t_out.y), t_out.z);
                if (head > 0) { count = head; first = payload; }

                // the hop to the neighbour cell does not depend on the triangle tests of this one
                int hx = 0, hy = 0, hz = 0, face = 0;
                if (seg_end == t_out.x) { hx = sx; face = 1 - nx; }          // face = 2 axis + (towards larger coordinates)
                else if (seg_end == t_out.y) { hy = sy; face = 3 - ny; }
                else if (seg_end == t_out.z) { hz = sz; face = 5 - nz; }
                const int leaf_at = node_at;
                hop_cell = (depth << 27) | ((cx + hx) << 18) | ((cy + hy) << 9) | (cz + hz);
                hop_face = face;
                rope_target = -1;
                while (depth > 0) {
                    bx += hx; by += hy; bz += hz;
                    if (bx >= 0 && bx <= 1 && by >= 0 && by <= 1 && bz >= 0 && bz <= 1) break;
                    depth--;
                    cx >>= 1; cy >>= 1; cz >>= 1;
                    bx = cx & 1; by = cy & 1; bz = cz & 1;
                    node_size *= 2.0f;
                }
                if (depth == 0) last_cell = true;        // leaves the root cube
                else {
                    const int old_idx = ((cx & 1) ? 4 : 0) + ((cy & 1) ? 2 : 0) + ((cz & 1) ? 1 : 0);
                    cx = (cx & ~1) | bx; cy = (cy & ~1) | by; cz = (cz & ~1) | bz;
                    child_idx = (bx > 0 ? 4 : 0) + (by > 0 ? 2 : 0) + (bz > 0 ? 1 : 0);
                    left_leaf = leaf_at | ((child_idx - old_idx + 8) << 27);
                    seg_begin = seg_end;
                    if (seg_begin > t_limit) last_cell = true;
                    else if ((hx | hy | hz) == 0) {
                        // no exit wall found (a NaN among the distances): like the reference, the
                        // same cell again
                        rope_target = leaf_at | (hop_cell & (15 << 27));
                        ahead.x = head; ahead.y = payload;
                    }
                    else {
                        // one 12-byte record per hop: (R | its depth << 27, R's node record)
                        const int * __restrict__ rope = node_ropes + 3*(6*leaf_at + face);
                        rope_target = rope[0];
                        ahead.x = rope[1]; ahead.y = rope[2];
                    }
                }
            }
### else
            if (!done) {
                SPLIT_BEGIN();
                vec3 ray_pos = origin + seg_begin * ray, node_origin = make_vec3(0.0f, 0.0f, 0.0f);
                int head = 0, payload = 0, child_idx;
                clr_i2u record = ahead;
### if obj.two_level
                // Records of objects/meshes.py: octree_two_level.  An inner node's record says which of
                // its children are inner nodes and where their children lie, so where the child under
                // ray_pos is one of them the descent goes on to the grandchild (the choice is
                // arithmetic on registers) and only the grandchild's record is read: the reference's
                // node sequence, one dependent access per two levels.
                while (depth < LEVELS-1) {
                    head = record.x;                   // >= 0: leaf with that many triangles
                    payload = record.y;                // first triangle record / children of the first inner child
                    node_origin = root_origin + make_vec3((float)cx, (float)cy, (float)cz)*node_size;
                    if (head >= 0) break;
                    const int first_child = head & 0x7fffff, inner_children = (head >> 23) & 255;
                    vec3 rc = ray_pos - node_origin;
                    float half = 0.5f*node_size;
                    bx = 1 - (rc.x < half); by = 1 - (rc.y < half); bz = 1 - (rc.z < half);
                    child_idx = (bx > 0 ? 4 : 0) + (by > 0 ? 2 : 0) + (bz > 0 ? 1 : 0);
                    int at = first_child + child_idx;
                    STACK(depth+1) = at;
                    cx = (cx << 1) | bx; cy = (cy << 1) | by; cz = (cz << 1) | bz;
                    node_size *= 0.5f;
                    depth++;
                    if ((inner_children >> child_idx) & 1) {
                        const int block = payload + 8*__popc(inner_children & ((1 << child_idx) - 1));
                        rc = ray_pos - (root_origin + make_vec3((float)cx, (float)cy, (float)cz)*node_size);
                        half = 0.5f*node_size;
                        bx = 1 - (rc.x < half); by = 1 - (rc.y < half); bz = 1 - (rc.z < half);
                        child_idx = (bx > 0 ? 4 : 0) + (by > 0 ? 2 : 0) + (bz > 0 ? 1 : 0);
                        at = block + child_idx;
                        STACK(depth+1) = at;
                        cx = (cx << 1) | bx; cy = (cy << 1) | by; cz = (cz << 1) | bz;
                        node_size *= 0.5f;
                        depth++;
                    }
                    record = *(const clr_i2u *)((const char *)octree_data + 8u*(unsigned)at);
                }
### else
                while (depth < LEVELS-1) {
                    head = record.x;                   // >= 0: leaf with that many triangles, -1: inner
                    payload = record.y;                // first triangle record / first child
                    node_origin = root_origin + make_vec3((float)cx, (float)cy, (float)cz)*node_size;
                    if (head >= 0) break;
                    const vec3 rc = ray_pos - node_origin;
                    const float half = 0.5f*node_size;
                    bx = 1 - (rc.x < half); by = 1 - (rc.y < half); bz = 1 - (rc.z < half);
                    child_idx = (bx > 0 ? 4 : 0) + (by > 0 ? 2 : 0) + (bz > 0 ? 1 : 0);
                    STACK(depth+1) = payload + child_idx;
                    cx = (cx << 1) | bx; cy = (cy << 1) | by; cz = (cz << 1) | bz;
                    node_size *= 0.5f;
                    depth++;
                    if (depth < LEVELS-1) record = *(const clr_i2u *)(octree_data + 2*(payload + child_idx));
                }
### endif
                SPLIT_END(10)     // waiting for the record asked for in the last step + descent
                t_out = slopes * (node_origin - origin + far_walls*node_size);
                seg_end = fminf(fminf(t_out.x, t_out.y), t_out.z);
                if (head > 0) { count = head; first = payload; }

                // the hop to the neighbour cell does not depend on the triangle tests of this one
                int hx = 0, hy = 0, hz = 0;
                if (seg_end == t_out.x) hx = sx;
                else if (seg_end == t_out.y) hy = sy;
                else if (seg_end == t_out.z) hz = sz;
                // The reference walks up level by level until the step stays inside a parent
                // (Octree.cl:120-140): that many levels as the low bits of the cell coordinate along
                // the hop are all ones (a step up) or all zeros (a step down) - a carry chain.
                {
                    const int along = hx ? cx : (hy ? cy : cz), step_sign = hx + hy + hz;
                    int up = step_sign > 0 ? __builtin_ctz(~along) : (step_sign < 0 && along ? __builtin_ctz(along) : (step_sign ? 32 : 0));
                    up = min(up, depth);
                    depth -= up;
                    cx >>= up; cy >>= up; cz >>= up;
                    node_size = ldexpf(node_size, up);
                }
                if (depth == 0) last_cell = true;        // leaves the root cube
                else {
                    // the eight children of a node lie side by side: the neighbour's record is the
                    // old one moved by the difference of the child numbers (no look at the parent)
                    cx += hx; cy += hy; cz += hz;
                    const int at = STACK(depth) + (4*hx + 2*hy + hz);
                    STACK(depth) = at;
                    seg_begin = seg_end;
                    if (seg_begin > t_limit) last_cell = true;
                    else ahead = *(const clr_i2u *)((const char *)octree_data + 8u*(unsigned)at);
                }
                SPLIT_END(11)     // exit distances, hop, request of the next record
            }
### endif
            pool_origin[tid] = make_float4(origin.x, origin.y, origin.z, seg_end);
            pool_best[tid] = ~0ull;
            PHASE_END(2)

            // ---- B: inclusive prefix sums of the counts over the lanes of this call
            // (neighbouring rays cross empty space together: a step in which no lane of the wave
            // stands in a leaf with triangles goes straight on to the hop)
            if (__any(count > 0)) {
            // bit plane b of the counts is one ballot; the lanes before this one that have the bit
            // set add 2^b to its sum (no LDS traffic, any set of executing lanes)
            SPLIT_BEGIN();
            int before = 0, total = 0;
            for (int b = 0; __any((count >> b) != 0); ++b) {
                const unsigned long long plane = __ballot((count >> b) & 1);
                before += __popcll(plane & lanes_before) << b;
                total += __popcll(plane) << b;
            }
            if (total > 64 * (POOL_WORDS - 1)) {
                // more items than the bitmap of range ends holds (leaves of hundreds of faces at the
                // depth limit): every lane walks its own leaf; same rule - smallest t, the earlier
                // triangle on ties
                const float * __restrict__ rec = tris + 9*first;
                const int skip_face = on_surface ? (int)prev_face : -1;
                float best_t = seg_end;
                int best_j = -1;
                for (int j = 0; j < count; ++j) {
                    const clr_f4u r0 = *(const clr_f4u *)(rec + 9*j), r1 = *(const clr_f4u *)(rec + 9*j + 4);
                    const float r2 = rec[9*j + 8];
                    if (skip_face >= 0 && skip_face == tri_ids[first + j]) continue;
                    const vec3 va = make_vec3(r0.x, r0.y, r0.z);
                    const vec3 e1 = make_vec3(r0.w, r1.x, r1.y);
                    const vec3 e2 = make_vec3(r1.z, r1.w, r2);
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
                    if (t > 0.0f && t < best_t) { best_t = t; best_j = j; }
                }
                if (best_j >= 0) pool_best[tid] = ((unsigned long long)__float_as_uint(best_t) << 32) | (unsigned)(first + best_j);
            } else {
            // The owner of item w is the q-th lane with triangles, q = number of range ends below w:
            // a bitmap of the ends in LDS, of which a round looks at the window of its own items.
            for (int i = rank; i <= ((total - 1) >> 6) + 1; i += n_lanes) ends[i] = 0ull;
            POOL_SYNC();
            if (count > 0) {
                const int q = __popcll(__ballot(1) & lanes_before), last = before + count - 1;
                pool_owner[wave_base + q] = make_int2(tid, first - before);
                atomicOr(&ends[last >> 6], 1ull << (last & 63));
            }
            POOL_SYNC();
            // Work item w = (owner lane, triangle j of its leaf).  Two stages, software-pipelined:
            // while the record of item w is on its way from memory, the owner of item w + n_lanes
            // is looked up and its record requested.  Rounds are fetched in order: `ends_below`
            // counts the range ends before the round being fetched.
            struct PoolItem { int owner, record; clr_f4u r0, r1; float r2; };   // owner < 0: none
            int ends_below = 0;
            const unsigned long long round_lanes = n_lanes == 64 ? ~0ull : (1ull << n_lanes) - 1ull;
            const unsigned long long ranks_before = (1ull << rank) - 1ull;
            auto pool_fetch = [&](const int round_base) {
                PoolItem it;
                it.owner = -1;
                if (round_base >= total) return it;
                const int word = round_base >> 6, shift = round_base & 63;
                const unsigned long long lo = ends[word], hi = ends[word + 1];
                const unsigned long long window = (shift ? (lo >> shift) | (hi << (64 - shift)) : lo) & round_lanes;
                const int w = round_base + rank;
                if (w < total) {
                    const int q = ends_below + __popcll(window & ranks_before);
                    const int2 owner = pool_owner[wave_base + q];
                    it.owner = owner.x;
                    const int record = it.record = w + owner.y;
                    // 32-bit byte offsets from the uniform base (scalar base + vector offset addressing;
                    // 36 B records: up to 119 M of them, checked where the data is packed)
                    const char * __restrict__ rec = (const char *)tris + 36u*(unsigned)record;
                    it.r0 = *(const clr_f4u *)rec; it.r1 = *(const clr_f4u *)(rec + 16);
                    it.r2 = *(const float *)(rec + 32);
                }
                ends_below += __popcll(window);
                return it;
            };
            SPLIT_END(12)         // prefix sums, range ends, owners
            auto pool_test = [&](const PoolItem &cur) {
                float4 o4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), d4 = o4;
                if (cur.owner >= 0) { o4 = pool_origin[cur.owner]; d4 = pool_ray[cur.owner]; }
                if (cur.owner >= 0) {
                    const vec3 o_ray = make_vec3(d4), o_origin = make_vec3(o4);
                    const vec3 va = make_vec3(cur.r0.x, cur.r0.y, cur.r0.z);
                    const vec3 e1 = make_vec3(cur.r0.w, cur.r1.x, cur.r1.y);
                    const vec3 e2 = make_vec3(cur.r1.z, cur.r1.w, cur.r2);
                    const vec3 pvec = cross(o_ray, e2);
                    const float det = dot(e1, pvec);
                    const float inv_det = 1.f / det;
                    const vec3 tvec = o_origin - va;
                    const float u = dot(tvec, pvec) * inv_det;
                    if (!(u < 0.f || u > 1.f)) {
                        const vec3 qvec = cross(tvec, e1);
                        const float v = dot(o_ray, qvec) * inv_det;
                        if (!(v < 0.f || u + v > 1.f)) {
                            const float t = dot(e2, qvec) * inv_det;
                            // the face the ray starts on is left out (TriangleMesh.cl:20): looked at
                            // only here, for the few items that got this far - a skipped triangle
                            // and a tested one whose hit is dropped leave the same minimum
                            if (t > 0.0f && t < o4.w && (__float_as_int(d4.w) < 0 || tri_ids[cur.record] != __float_as_int(d4.w)))
                                atomicMin(&pool_best[cur.owner],
                                          ((unsigned long long)__float_as_uint(t) << 32) | (unsigned)cur.record);
                        }
                    }
                }
            };
            PoolItem cur = pool_fetch(0);
            for (int base = 0; base < total; base += n_lanes) {
                const PoolItem next = pool_fetch(base + n_lanes);
                pool_test(cur);
                cur = next;
            }
            SPLIT_END(13)         // the rounds of tests
            }
            PHASE_ADD(7, 1); PHASE_ADD(8, total);
            }
            POOL_SYNC();
            PHASE_END(3)

            // ---- C: own result, or on to the cell the hop went to
            if (!done) {
                const unsigned long long best = pool_best[tid];
                if (best != ~0ull) {
                    *t_hit = __uint_as_float((uint)(best >> 32));
                    *p_subobject = (uint)tri_ids[(int)(best & 0xffffffffull)];
                    done = true;
                }
                else if (last_cell) done = true;
            }
            POOL_SYNC();
            PHASE_END(4)
        }
        PHASE_ADD(0, 1); PHASE_ADD(9, n_lanes);
        ### if obj.profile_phases
        if (rank == 0)
            for (int slot = 0; slot < 4; ++slot) phase_split[(tid >> 6) * 4 + slot] = 0ull;
        POOL_SYNC();
        for (int slot = 0; slot < 4; ++slot) atomicMax(&phase_split[(tid >> 6) * 4 + slot], (unsigned long long)phase_sum[10 + slot]);
        POOL_SYNC();
        if (rank == 0) {
            for (int slot = 0; slot < 10; ++slot)
                atomicAdd(&clr_mesh_phases[(blockIdx.x & 63) * 16 + slot], (unsigned long long)phase_sum[slot]);
            for (int slot = 0; slot < 4; ++slot)
                atomicAdd(&clr_mesh_phases[(blockIdx.x & 63) * 16 + 10 + slot], phase_split[(tid >> 6) * 4 + slot]);
        }
        POOL_SYNC();
        ### endif
### if not obj.ropes
        #undef STACK
### endif
        #undef POOL_SYNC
        #undef POOL_WORDS
        #undef PHASE_ADD
        #undef PHASE_END
        #undef SPLIT_BEGIN
        #undef SPLIT_END
    ### endcall
### endmacro

### macro Octree__tracer(obj)
### if obj.pooled
{{ Octree_pooled__tracer(obj) }}
### else
    ### call tracer_function_base(obj)
        ### set flat = obj.layout == 'flat'
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
        ### if flat
        {{ flat_record_types() }}
        const float * __restrict__ tris = (const float *)(vector_data + flat_tri_offset);
        const int * __restrict__ tri_ids = octree_data + flat_ids_offset;
        ### endif
        const int LEVELS = {{ obj.levels }};
        const int MAX_STEPS = {{ obj.max_steps }};

        // stack[level] = position of the node on the current branch (stream: header offset,
        // flat: node index); one LDS column per lane of the 256-thread block
        __shared__ int stack_lds[{{ obj.levels }} * 256];
        int *stack = stack_lds + threadIdx.x;
        #define STACK(level) stack[(level) * 256]

        int cx = 0, cy = 0, cz = 0;       // integer cell coordinates at the current depth
        STACK(0) = root_data_offset;
        int depth = 0;
        uint prev_face = *p_subobject, face = 0;
        float node_size = root_size;

        for (int step = 0; step < MAX_STEPS; step++) {
            vec3 ray_pos = origin + seg_begin * ray, node_origin = make_vec3(0.0f, 0.0f, 0.0f);
            int head = 0, payload = 0, child_idx;

            while (depth < LEVELS-1) {
                const int at = STACK(depth);
                ### if flat
                head = octree_data[2*at];          // >= 0: leaf with that many triangles, -1: inner
                payload = octree_data[2*at+1];     // first triangle record / first child
                ### else
                head = octree_data[at];            // child mask; 0 = leaf, 0x100 = empty leaf
                payload = octree_data[at+1];
                ### endif
                node_origin = root_origin + make_vec3((float)cx, (float)cy, (float)cz)*node_size;
                ### if flat
                if (head >= 0) break;
                ### else
                if (head == 0 || head == 0x100) break;
                ### endif
                const vec3 rc = ray_pos - node_origin;
                const float half = 0.5f*node_size;
                const int bx = 1 - (rc.x < half), by = 1 - (rc.y < half), bz = 1 - (rc.z < half);
                child_idx = (bx > 0 ? 4 : 0) + (by > 0 ? 2 : 0) + (bz > 0 ? 1 : 0);
                STACK(depth+1) = payload + child_idx{{ '' if flat else '*2' }};
                cx = (cx << 1) | bx; cy = (cy << 1) | by; cz = (cz << 1) | bz;
                node_size *= 0.5f;
                depth++;
            }

            t_out = slopes * (node_origin - origin + far_walls*node_size);
            seg_end = fminf(fminf(t_out.x, t_out.y), t_out.z);

            ### if flat
            if (head > 0) {
                const int n_leaf_triangles = head, first_triangle = payload;
                float best_t = seg_end;
                {{ flat_triangle_loop() }}
                if (best_t < seg_end) {
                    *t_hit = best_t;
                    *p_subobject = face;
                    return;
                }
            }
            ### else
            if (head != 0x100) {
                const uint n_triangles = (uint)max(octree_data[payload], 0);
                if (n_triangles > 0) {
                    float best_t = seg_end;
                    ### call(j) triangle_loop()
                        octree_data[payload + 1 + {{ j }}]
                    ### endcall
                    if (best_t < seg_end) {
                        *t_hit = best_t;
                        *p_subobject = face;
                        return;
                    }
                }
            }
            ### endif

            int hx = 0, hy = 0, hz = 0;
            if (seg_end == t_out.x) hx = sx;
            else if (seg_end == t_out.y) hy = sy;
            else if (seg_end == t_out.z) hz = sz;

            // The reference walks up level by level until the step stays inside a parent
            // (Octree.cl:120-140): that many levels as the low bits of the cell coordinate along
            // the hop are all ones (a step up) or all zeros (a step down) - a carry chain.
            {
                const int along = hx ? cx : (hy ? cy : cz), step_sign = hx + hy + hz;
                int up = step_sign > 0 ? __builtin_ctz(~along) : (step_sign < 0 && along ? __builtin_ctz(along) : (step_sign ? 32 : 0));
                up = min(up, depth);
                depth -= up;
                cx >>= up; cy >>= up; cz >>= up;
                node_size = ldexpf(node_size, up);
            }
            if (depth == 0) return;        // left the root cube

            // the eight children of a node lie side by side: the neighbour's record is the old one
            // moved by the difference of the child numbers (no look at the parent)
            cx += hx; cy += hy; cz += hz;
            STACK(depth) = STACK(depth) + (4*hx + 2*hy + hz){{ '' if flat else '*2' }};

            seg_begin = seg_end;
            if (seg_begin > t_limit) return;
        }
        #undef STACK
    ### endcall
### endif
### endmacro

### macro Octree__normal(obj)
    {{ TriangleMesh__normal(obj) }}
### endmacro
