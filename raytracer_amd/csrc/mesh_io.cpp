// Native mesh readers (host only): OFF and Stanford "zipper" (ASCII PLY) files and the
// duplicate-face filter, with the record semantics of the reference's Python readers
// (/root/reference/mesh_formats.py:14-82): one record per line, polygons fan-triangulated
// around their first vertex, duplicate = same SET of vertex indices, first occurrence kept.
// A million-face file parses in a fraction of a second instead of the minute the
// line-by-line Python loops take.
#include "clray_hip.h"

#include <algorithm>
#include <array>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <unordered_set>
#include <vector>

extern "C" __attribute__((visibility("hidden"))) void clr_set_error(const char *text);

struct clr_mesh {
    std::vector<double> vertices;   // n x 3
    std::vector<int> triangles;     // m x 3
};

namespace {

int fail(const std::string &what) {
    clr_set_error(what.c_str());
    return 1;
}

struct Lines {
    std::vector<char> text;
    size_t pos = 0;
    long line_no = 0;

    // next line as [begin, end), without the newline; false at end of file
    bool next(char *&begin, char *&end) {
        if (pos >= text.size()) return false;
        begin = text.data() + pos;
        char *nl = (char *)memchr(begin, '\n', text.size() - pos);
        end = nl ? nl : text.data() + text.size();
        pos = (nl ? (size_t)(nl - text.data()) + 1 : text.size());
        *end = '\0';                      // the buffer has one spare byte at the end
        ++line_no;
        return true;
    }
};

bool load(const char *path, Lines &in) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    // a regular, seekable file: a FIFO or a /proc entry reports no size (-1)
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return false; }
    long size = ftell(f);
    if (size < 0 || fseek(f, 0, SEEK_SET) != 0) { fclose(f); return false; }
    in.text.resize((size_t)size + 1);
    size_t got = size ? fread(in.text.data(), 1, (size_t)size, f) : 0;
    fclose(f);
    in.text.resize(got + 1);
    in.text[got] = '\0';
    in.text.resize(got);                  // capacity keeps the spare byte
    in.text.reserve(got + 1);
    return true;
}

const char *skip_space(const char *p) {
    while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\f' || *p == '\v') ++p;
    return p;
}

std::string where(const Lines &in) { return " (line " + std::to_string(in.line_no) + ")"; }

// all whitespace-separated numbers of a line, as Python's [float(x) for x in line.split()]
bool parse_doubles(const char *p, std::vector<double> &out) {
    out.clear();
    for (p = skip_space(p); *p; p = skip_space(p)) {
        char *stop;
        errno = 0;
        double v = strtod(p, &stop);
        if (stop == p || (*stop && !strchr(" \t\r\f\v", *stop))) return false;
        out.push_back(v);
        p = stop;
    }
    return true;
}

bool parse_ints(const char *p, std::vector<long> &out) {
    out.clear();
    for (p = skip_space(p); *p; p = skip_space(p)) {
        char *stop;
        errno = 0;
        long v = strtol(p, &stop, 10);
        if (stop == p || (*stop && !strchr(" \t\r\f\v", *stop)) || errno == ERANGE) return false;
        out.push_back(v);
        p = stop;
    }
    return true;
}

// face records: "k i0 i1 ... i(k-1)" -> k-2 triangles (mesh_formats.py:10-12, 33-37)
int read_polygons(Lines &in, long n_faces, long n_vertices, std::vector<int> &triangles) {
    std::vector<long> fields;
    char *b, *e;
    for (long f = 0; f < n_faces; ++f) {
        if (!in.next(b, e)) return fail("unexpected end of file in the face list" + where(in));
        if (!parse_ints(b, fields) || fields.empty()) return fail("bad face record" + where(in));
        if (fields[0] != (long)fields.size() - 1)
            return fail("face record: vertex count does not match the record" + where(in));
        for (size_t k = 1; k < fields.size(); ++k)
            if (fields[k] < 0 || fields[k] >= n_vertices)
                return fail("face record: vertex index out of range" + where(in));
        for (size_t k = 2; k + 1 < fields.size(); ++k) {
            triangles.push_back((int)fields[1]);
            triangles.push_back((int)fields[k]);
            triangles.push_back((int)fields[k + 1]);
        }
    }
    return 0;
}

int read_off(Lines &in, clr_mesh &mesh) {
    char *b, *e;
    if (!in.next(b, e)) return fail("not OFF format");
    const char *p = skip_space(b);
    if (strncmp(p, "OFF", 3) != 0) return fail("not OFF format");
    p = skip_space(p + 3);
    std::vector<long> counts;
    if (!*p) {
        if (!in.next(b, e)) return fail("OFF: missing counts");
        p = b;
    }
    if (!parse_ints(p, counts) || counts.size() != 3 || counts[0] < 0 || counts[1] < 0)
        return fail("OFF: expected `n_points n_faces n_edges`" + where(in));
    // every record takes a line of at least two bytes: counts beyond that are not this file's
    if ((size_t)counts[0] > in.text.size() / 2 || (size_t)counts[1] > in.text.size() / 2)
        return fail("OFF: the header announces more records than the file can hold" + where(in));
    std::vector<double> row;
    mesh.vertices.reserve((size_t)counts[0] * 3);
    for (long v = 0; v < counts[0]; ++v) {
        if (!in.next(b, e)) return fail("unexpected end of file in the vertex list" + where(in));
        if (!parse_doubles(b, row) || row.size() < 3) return fail("bad vertex record" + where(in));
        mesh.vertices.insert(mesh.vertices.end(), row.begin(), row.begin() + 3);
    }
    return read_polygons(in, counts[1], counts[0], mesh.triangles);
}

int read_zipper(Lines &in, clr_mesh &mesh) {
    long n_vertices = -1, n_faces = -1;
    char *b, *e;
    for (;;) {
        if (!in.next(b, e)) return fail("unexpected eof/empty line");
        char key[32] = "", what[32] = "";
        long count = 0;
        int n = sscanf(b, "%31s %31s %ld", key, what, &count);
        if (n < 1) return fail("unexpected eof/empty line");
        if (!strcmp(key, "element") && n == 3) {
            if (!strcmp(what, "vertex")) n_vertices = count;
            else if (!strcmp(what, "face")) n_faces = count;
        } else if (!strcmp(key, "end_header")) {
            break;
        }
    }
    if (n_vertices < 0 || n_faces < 0) return fail("zipper: header lacks element vertex / element face");
    if ((size_t)n_vertices > in.text.size() / 2 || (size_t)n_faces > in.text.size() / 2)
        return fail("zipper: the header announces more records than the file can hold");
    std::vector<double> row;
    mesh.vertices.reserve((size_t)n_vertices * 3);
    for (long v = 0; v < n_vertices; ++v) {
        if (!in.next(b, e)) return fail("unexpected end of file in the vertex list" + where(in));
        // the reference converts the first three fields only; the rest may be anything
        const char *p = b;
        for (int c = 0; c < 3; ++c) {
            p = skip_space(p);
            char *stop;
            double x = strtod(p, &stop);
            if (stop == p) return fail("bad vertex record" + where(in));
            mesh.vertices.push_back(x);
            p = stop;
        }
    }
    return read_polygons(in, n_faces, n_vertices, mesh.triangles);
}

struct KeyHash {
    size_t operator()(const std::array<int, 3> &k) const {
        unsigned long long h = 1469598103934665603ull;
        for (int v : k) { h ^= (unsigned)v; h *= 1099511628211ull; }
        return (size_t)h;
    }
};

}  // namespace

extern "C" {

int clr_mesh_read(const char *path, int format, clr_mesh **out) {
    if (!path || !out) return fail("null argument");
    clr_mesh *mesh = nullptr;
    try {                                  // no C++ exception may cross the C boundary
        Lines in;
        if (!load(path, in)) return fail(std::string("cannot read ") + path + " (missing, or not a regular file)");
        mesh = new clr_mesh;
        int rc = format == 0 ? read_off(in, *mesh) : format == 1 ? read_zipper(in, *mesh) : fail("unknown mesh format");
        if (rc) { delete mesh; return rc; }
    } catch (const std::exception &e) {
        delete mesh;
        return fail(std::string("clr_mesh_read: ") + e.what());
    } catch (...) {
        delete mesh;
        return fail("clr_mesh_read: unknown failure");
    }
    *out = mesh;
    return 0;
}

int clr_mesh_info(const clr_mesh *mesh, int *n_vertices, int *n_triangles) {
    if (!mesh) return fail("null mesh");
    if (n_vertices) *n_vertices = (int)(mesh->vertices.size() / 3);
    if (n_triangles) *n_triangles = (int)(mesh->triangles.size() / 3);
    return 0;
}

int clr_mesh_copy(const clr_mesh *mesh, double *vertices, int *triangles) {
    if (!mesh) return fail("null mesh");
    if (vertices) memcpy(vertices, mesh->vertices.data(), mesh->vertices.size() * sizeof(double));
    if (triangles) memcpy(triangles, mesh->triangles.data(), mesh->triangles.size() * sizeof(int));
    return 0;
}

int clr_mesh_destroy(clr_mesh *mesh) {
    delete mesh;
    return 0;
}

int clr_mesh_unique_faces(const int *faces, int n_faces, int *kept, int *n_kept) {
    if (!faces || !kept || !n_kept || n_faces < 0) return fail("bad arguments");
    try {
    std::unordered_set<std::array<int, 3>, KeyHash> seen;
    seen.reserve((size_t)n_faces * 2);
    int count = 0;
    for (int f = 0; f < n_faces; ++f) {
        // the key is the SET of indices: sorted, repeated indices collapsed (frozenset semantics)
        std::array<int, 3> k = {faces[3*f], faces[3*f+1], faces[3*f+2]};
        std::sort(k.begin(), k.end());
        if (k[1] == k[2]) k[2] = -1;
        if (k[0] == k[1]) { k[1] = k[2]; k[2] = -1; }
        if (seen.insert(k).second) kept[count++] = f;
    }
    *n_kept = count;
    } catch (const std::exception &e) {
        return fail(std::string("clr_mesh_unique_faces: ") + e.what());
    }
    return 0;
}

}  // extern "C"
