"""Readers for the OFF and Stanford "zipper" (ASCII PLY) mesh formats.

Same functions and return values (vertex list, triangle list) as
/root/reference/mesh_formats.py:3-82; polygons are fan-triangulated.
"""


def open_or_yield(filename_or_file):
    if isinstance(filename_or_file, str):
        with open(filename_or_file, 'r') as f:
            yield f
    else:
        yield filename_or_file


def poly_to_triangle_fan(vertices):
    for k in range(1, len(vertices) - 1):
        yield [vertices[0], vertices[k], vertices[k + 1]]


def _read_polygons(stream, n_faces):
    triangles = []
    for _ in range(n_faces):
        fields = [int(tok) for tok in stream.readline().split()]
        assert fields[0] == len(fields) - 1
        triangles.extend(poly_to_triangle_fan(fields[1:]))
    return triangles


def read_off(f):
    for off in open_or_yield(f):
        head, sep, rest = off.readline().strip().partition('OFF')
        if head or sep != 'OFF':
            raise RuntimeError('not OFF format')
        rest = rest.strip() or off.readline().strip()
        n_points, n_faces, _ = (int(tok) for tok in rest.split())
        vertices = [[float(tok) for tok in off.readline().split()] for _ in range(n_points)]
        return vertices, _read_polygons(off, n_faces)


def read_zipper(f):
    """The format the Stanford bunny is published in."""
    for zipper in open_or_yield(f):
        n_vertices = n_faces = None
        while True:
            fields = zipper.readline().strip().split()
            if not fields:
                raise RuntimeError("unexpected eof/empty line")
            if fields[0] == 'element' and fields[1] == 'vertex':
                n_vertices = int(fields[2])
            elif fields[0] == 'element' and fields[1] == 'face':
                n_faces = int(fields[2])
            elif fields[0] == 'end_header':
                break
        vertices = []
        for _ in range(n_vertices):
            fields = zipper.readline().strip().split()
            vertices.append([float(fields[c]) for c in range(3)])
        return vertices, _read_polygons(zipper, n_faces)


def remove_duplicate_faces(faces, verbose=False):
    seen = set()
    unique = []
    for face in faces:
        key = frozenset(face)
        if key in seen:
            if verbose:
                print("WARNING: removed duplicate face %s" % face)
            continue
        seen.add(key)
        unique.append(face)
    return unique
