// octree_build.cpp - host-side octree builder for triangle meshes (part of libclray_hip.so).
//
// Same construction rule as the reference's objects/octree.py:69-117 / 201-233 (root = 1.01 x
// the bounding cube; a face belongs to a node when its bounding sphere overlaps the node's
// bounding sphere AND the node sphere reaches the face's plane; a node becomes a leaf when at
// most max_faces_per_leaf of its faces have a bounding radius below half the node size, or at
// max_depth), but built top-down per node in O(N log N) instead of the reference's dense
// (nodes x faces) numpy matrices, which cannot be materialised for 1 M faces.  Testing a child
// only against its parent's faces is equivalent: the child's bounding sphere lies inside the
// parent's, so the overlap predicate is monotone.
//
// Second overlap rule, for throughput (clr_octree_build_rule, rule 1): a face belongs to a node
// when the TRIANGLE intersects the node's box (separating-axis test of Akenine-Moeller 2001, box
// widened by a small margin against rounding).  The bounding-sphere rule above files a face of a
// fine mesh under a dozen leaves it never touches (12.7 leaf records per face on the 1 M-triangle
// knot); the exact rule changes which leaves list a face, not which face a ray hits: the
// traversal accepts a hit only up to the leaf's exit distance, such a hit point lies inside the
// leaf's box, so the face intersects the box and is listed there under either rule.
//
// Two serialisations: the reference's integer stream (objects/octree.py:119-156), used for
// parity with the reference layout, and a flattened layout for the HIP traversal kernel:
//   nodes: int2 per node, children of an inner node contiguous;
//          a >= 0: leaf with a triangles starting at record b;  a == -1: inner, b = first child
//   tris : 3 float4 per (leaf, face) reference: (v0, face id bits), (e1, 0), (e2, 0), leaves
//          contiguous, so a leaf is one coalesced 48*n-byte read instead of the reference's
//          index (12 B) + three gathered vertices (48 B) per test.
#include "clray_hip.h"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Node {
    double origin[3];
    double size;
    std::vector<int> faces;
    int child[8];
    bool leaf;
};

}  // namespace

struct clr_octree {
    std::vector<Node> nodes;          // nodes[0] = root
    std::vector<int> stream;          // reference serialisation (node part)
    int root_data_offset;
    std::vector<int> flat_nodes;      // 2 ints per node
    std::vector<float> flat_tris;     // 12 floats per triangle reference
    double root_origin[3];
    double root_size;
    int n_leaves, max_depth_reached;
    long long leaf_face_refs;
};

namespace {

struct Builder {
    const float *v;
    const int *f;
    int n_faces, max_depth, max_leaf;
    int rule = 0;                      // 0: the reference's bounding-sphere rule, 1: triangle-box
    std::vector<double> cx, cy, cz, rad, nx, ny, nz, h;
    clr_octree *out;

    void face_geometry() {
        cx.resize(n_faces); cy.resize(n_faces); cz.resize(n_faces); rad.resize(n_faces);
        nx.resize(n_faces); ny.resize(n_faces); nz.resize(n_faces); h.resize(n_faces);
        for (int i = 0; i < n_faces; ++i) {
            double p[3][3];
            for (int k = 0; k < 3; ++k)
                for (int c = 0; c < 3; ++c) p[k][c] = v[3 * f[3 * i + k] + c];
            double c[3];
            for (int d = 0; d < 3; ++d) c[d] = (p[0][d] + p[1][d] + p[2][d]) / 3.0;
            double r2 = 0;
            for (int k = 0; k < 3; ++k) {
                double d2 = 0;
                for (int d = 0; d < 3; ++d) d2 += (p[k][d] - c[d]) * (p[k][d] - c[d]);
                if (d2 > r2) r2 = d2;
            }
            cx[i] = c[0]; cy[i] = c[1]; cz[i] = c[2]; rad[i] = std::sqrt(r2);
            double e1[3], e2[3];
            for (int d = 0; d < 3; ++d) { e1[d] = p[1][d] - p[0][d]; e2[d] = p[2][d] - p[0][d]; }
            double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            nx[i] = n[0] / len; ny[i] = n[1] / len; nz[i] = n[2] / len;
            h[i] = p[0][0] * nx[i] + p[0][1] * ny[i] + p[0][2] * nz[i];
        }
    }

    // exact triangle / axis-aligned box overlap: 3 box axes, the face normal, 9 edge cross products
    bool triangle_touches_box(const Node &node, int face) const {
        const double margin = 1e-6;                        // relative widening of the box
        const double half = node.size * 0.5 * (1.0 + margin) + 1e-12;
        double p[3][3];
        for (int k = 0; k < 3; ++k)
            for (int d = 0; d < 3; ++d)
                p[k][d] = (double)v[3 * f[3 * face + k] + d] - (node.origin[d] + node.size * 0.5);
        for (int d = 0; d < 3; ++d) {
            const double lo = std::fmin(p[0][d], std::fmin(p[1][d], p[2][d]));
            const double hi = std::fmax(p[0][d], std::fmax(p[1][d], p[2][d]));
            if (lo > half || hi < -half) return false;
        }
        const double n[3] = {nx[face], ny[face], nz[face]};
        const double dist = n[0] * p[0][0] + n[1] * p[0][1] + n[2] * p[0][2];
        if (std::fabs(dist) > half * (std::fabs(n[0]) + std::fabs(n[1]) + std::fabs(n[2]))) return false;
        for (int e = 0; e < 3; ++e) {
            double edge[3];
            for (int d = 0; d < 3; ++d) edge[d] = p[(e + 1) % 3][d] - p[e][d];
            for (int a = 0; a < 3; ++a) {
                // axis = unit vector a x edge
                double axis[3] = {0, 0, 0};
                const int b = (a + 1) % 3, c = (a + 2) % 3;
                axis[b] = -edge[c];
                axis[c] = edge[b];
                double lo = 0, hi = 0;
                for (int k = 0; k < 3; ++k) {
                    const double s = axis[0] * p[k][0] + axis[1] * p[k][1] + axis[2] * p[k][2];
                    if (k == 0 || s < lo) lo = s;
                    if (k == 0 || s > hi) hi = s;
                }
                const double r = half * (std::fabs(axis[0]) + std::fabs(axis[1]) + std::fabs(axis[2]));
                const double slack = 1e-9 * (std::fabs(lo) + std::fabs(hi) + r);
                if (lo > r + slack || hi < -r - slack) return false;
            }
        }
        return true;
    }

    bool overlaps(const Node &node, int face) const {
        if (rule == 1) return triangle_touches_box(node, face);
        const double half = node.size * 0.5;
        const double c[3] = {node.origin[0] + half, node.origin[1] + half, node.origin[2] + half};
        const double r = std::sqrt(3.0) * node.size * 0.5;
        const double dx = c[0] - cx[face], dy = c[1] - cy[face], dz = c[2] - cz[face];
        const double gap = std::sqrt(dx * dx + dy * dy + dz * dz) - r - rad[face];
        const double plane_gap = std::fabs(nx[face] * c[0] + ny[face] * c[1] + nz[face] * c[2] - h[face]) - r;
        return (gap > plane_gap ? gap : plane_gap) < 0;
    }

    void split(int index, int depth) {
        if (depth > out->max_depth_reached) out->max_depth_reached = depth;
        int n_small = 0;
        {
            const Node &node = out->nodes[index];
            for (int face : node.faces)
                if (rad[face] < node.size * 0.5) ++n_small;
        }
        if (n_small <= max_leaf || depth == max_depth) {
            out->nodes[index].leaf = true;
            out->n_leaves++;
            out->leaf_face_refs += (long long)out->nodes[index].faces.size();
            return;
        }
        out->nodes[index].leaf = false;
        std::vector<int> parent_faces;
        parent_faces.swap(out->nodes[index].faces);
        for (int k = 0; k < 8; ++k) {
            Node child;
            const Node &node = out->nodes[index];
            const int xyz[3] = {(k >> 2) & 1, (k >> 1) & 1, k & 1};
            for (int d = 0; d < 3; ++d) child.origin[d] = node.origin[d] + xyz[d] * 0.5 * node.size;
            child.size = node.size * 0.5;
            child.leaf = true;
            for (int d = 0; d < 8; ++d) child.child[d] = -1;
            for (int face : parent_faces)
                if (overlaps(child, face)) child.faces.push_back(face);
            const int ci = (int)out->nodes.size();
            out->nodes.push_back(std::move(child));
            out->nodes[index].child[k] = ci;
        }
        for (int k = 0; k < 8; ++k) split(out->nodes[index].child[k], depth + 1);
    }

    // reference stream: returns the (mask, offset) header of the node (octree.py:122-143)
    void write_node(int index, int header[2]) {
        const Node &node = out->nodes[index];
        std::vector<int> payload;
        int mask = 0;
        if (node.leaf) {
            if (node.faces.empty()) { header[0] = 0x100; header[1] = -99999; return; }
            payload.push_back((int)node.faces.size());
            payload.insert(payload.end(), node.faces.begin(), node.faces.end());
        } else {
            for (int k = 0; k < 8; ++k) {
                const Node &c = out->nodes[node.child[k]];
                int hd[2] = {0x100, -99999};
                if (!(c.leaf && c.faces.empty())) {
                    mask |= 1 << k;
                    write_node(node.child[k], hd);
                }
                payload.push_back(hd[0]);
                payload.push_back(hd[1]);
            }
            if (mask == 0) { header[0] = 0x100; header[1] = -99999; return; }
        }
        header[0] = mask;
        header[1] = (int)out->stream.size();
        out->stream.insert(out->stream.end(), payload.begin(), payload.end());
    }

    void flatten() {
        // breadth-first numbering so that the 8 children of a node are consecutive records
        std::vector<int> order{0}, first_child(out->nodes.size(), -1), new_index(out->nodes.size(), -1);
        new_index[0] = 0;
        for (size_t q = 0; q < order.size(); ++q) {
            const Node &node = out->nodes[order[q]];
            if (node.leaf) continue;
            first_child[order[q]] = (int)order.size();
            for (int k = 0; k < 8; ++k) {
                new_index[node.child[k]] = (int)order.size();
                order.push_back(node.child[k]);
            }
        }
        out->flat_nodes.assign(2 * order.size(), 0);
        out->flat_tris.clear();
        out->flat_tris.reserve((size_t)out->leaf_face_refs * 12);
        for (size_t q = 0; q < order.size(); ++q) {
            const Node &node = out->nodes[order[q]];
            if (!node.leaf) {
                out->flat_nodes[2 * q] = -1;
                out->flat_nodes[2 * q + 1] = first_child[order[q]];
                continue;
            }
            out->flat_nodes[2 * q] = (int)node.faces.size();
            out->flat_nodes[2 * q + 1] = (int)(out->flat_tris.size() / 12);
            for (int face : node.faces) {
                const float *a = v + 3 * f[3 * face], *b = v + 3 * f[3 * face + 1], *c = v + 3 * f[3 * face + 2];
                float rec[12];
                for (int d = 0; d < 3; ++d) {
                    rec[d] = a[d];
                    rec[4 + d] = b[d] - a[d];          // binary32 subtraction, as the kernel would do
                    rec[8 + d] = c[d] - a[d];
                }
                std::memcpy(&rec[3], &face, 4);
                rec[7] = rec[11] = 0.0f;
                out->flat_tris.insert(out->flat_tris.end(), rec, rec + 12);
            }
        }
    }
};

thread_local std::string g_octree_error;

}  // namespace

extern "C" {

int clr_octree_build(const float *vertices, int n_vertices, const int *faces, int n_faces,
                     int max_depth, int max_faces_per_leaf, clr_octree **out) {
    return clr_octree_build_rule(vertices, n_vertices, faces, n_faces, max_depth, max_faces_per_leaf, 0, out);
}

int clr_octree_build_rule(const float *vertices, int n_vertices, const int *faces, int n_faces,
                          int max_depth, int max_faces_per_leaf, int overlap_rule, clr_octree **out) {
    if (!vertices || !faces || !out || n_vertices <= 0 || n_faces <= 0 || max_depth < 0 || max_depth > 20 ||
        overlap_rule < 0 || overlap_rule > 1)
        return 1;
    for (int i = 0; i < 3 * n_faces; ++i)
        if (faces[i] < 0 || faces[i] >= n_vertices) return 1;
    clr_octree *oct = nullptr;
    try {                                  // no C++ exception (allocation failure) may cross the C boundary
    oct = new clr_octree;
    Builder b;
    b.rule = overlap_rule;
    b.v = vertices; b.f = faces; b.n_faces = n_faces; b.max_depth = max_depth; b.max_leaf = max_faces_per_leaf;
    b.out = oct;
    // bounding cube of the binary32 vertices (triangle_mesh.py:54-69), widened by 1 % (octree.py:71-74)
    float lo[3], hi[3];
    for (int d = 0; d < 3; ++d) lo[d] = hi[d] = vertices[d];
    for (int i = 0; i < n_vertices; ++i)
        for (int d = 0; d < 3; ++d) {
            if (vertices[3 * i + d] < lo[d]) lo[d] = vertices[3 * i + d];
            if (vertices[3 * i + d] > hi[d]) hi[d] = vertices[3 * i + d];
        }
    double side = 0, center[3];
    for (int d = 0; d < 3; ++d) {
        center[d] = ((double)lo[d] + (double)hi[d]) * 0.5;
        const double extent = std::fabs((double)(float)(hi[d] - lo[d]));
        if (extent > side) side = extent;
    }
    oct->root_size = side * 1.01;
    for (int d = 0; d < 3; ++d) oct->root_origin[d] = center[d] - oct->root_size * 0.5;
    oct->n_leaves = 0; oct->max_depth_reached = 0; oct->leaf_face_refs = 0;

    b.face_geometry();
    Node root;
    for (int d = 0; d < 3; ++d) root.origin[d] = oct->root_origin[d];
    root.size = oct->root_size;
    root.leaf = true;
    for (int d = 0; d < 8; ++d) root.child[d] = -1;
    oct->nodes.push_back(root);
    for (int i = 0; i < n_faces; ++i)
        if (b.overlaps(oct->nodes[0], i)) oct->nodes[0].faces.push_back(i);
    b.split(0, 0);

    int header[2];
    b.write_node(0, header);
    oct->root_data_offset = (int)oct->stream.size();
    oct->stream.push_back(header[0]);
    oct->stream.push_back(header[1]);
    b.flatten();
    } catch (...) {
        delete oct;
        return 2;
    }
    *out = oct;
    return 0;
}

int clr_octree_info(const clr_octree *oct, clr_octree_info_t *info) {
    if (!oct || !info) return 1;
    info->n_nodes = (int)oct->nodes.size();
    info->n_leaves = oct->n_leaves;
    info->max_depth = oct->max_depth_reached;
    info->stream_length = (int)oct->stream.size();
    info->root_data_offset = oct->root_data_offset;
    info->flat_triangles = (int)(oct->flat_tris.size() / 12);
    for (int d = 0; d < 3; ++d) info->root_origin[d] = oct->root_origin[d];
    info->root_size = oct->root_size;
    return 0;
}

int clr_octree_copy(const clr_octree *oct, int *stream, int *flat_nodes, float *flat_tris) {
    if (!oct) return 1;
    if (stream) std::memcpy(stream, oct->stream.data(), oct->stream.size() * sizeof(int));
    if (flat_nodes) std::memcpy(flat_nodes, oct->flat_nodes.data(), oct->flat_nodes.size() * sizeof(int));
    if (flat_tris) std::memcpy(flat_tris, oct->flat_tris.data(), oct->flat_tris.size() * sizeof(float));
    return 0;
}

int clr_octree_destroy(clr_octree *oct) {
    delete oct;
    return 0;
}

}  // extern "C"
