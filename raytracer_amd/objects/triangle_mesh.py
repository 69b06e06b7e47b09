"""Brute-force triangle mesh (reference objects/triangle_mesh.py:5-117)."""
import numpy
from ..tracer import Tracer


def to_vector_array(array_of_arrays, dtype=numpy.float32):
    arr = numpy.array(array_of_arrays, dtype=dtype)
    if arr.ndim != 2 or arr.shape[-1] != 3:
        raise RuntimeError('invalid shape %s' % (arr.shape,))
    return arr


def generate_smooth_normals(vertices, faces):
    """Per-vertex mean of the unit normals of the incident faces (unreferenced vertices: (1,1,1))."""
    v = vertices.astype(numpy.float32)
    tri = v[faces]
    face_n = numpy.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    face_n = face_n / numpy.sqrt(numpy.sum(face_n ** 2, axis=1))[:, None]
    sums = numpy.zeros(v.shape, dtype=numpy.float64)
    counts = numpy.zeros(v.shape[0], dtype=numpy.int64)
    for corner in range(3):
        numpy.add.at(sums, faces[:, corner], face_n)
        numpy.add.at(counts, faces[:, corner], 1)
    normals = numpy.ones(v.shape)
    used = counts > 0
    mean = sums[used] / counts[used][:, None]
    normals[used] = mean / numpy.sqrt(numpy.sum(mean ** 2, axis=1))[:, None]
    return normals


class TriangleMesh(Tracer):

    def __init__(self, vertices, faces, normals=None, center=(0, 0, 0), scale=1.0,
                 auto_scale=False, auto_flip_normal=False, shading='flat',
                 auto_smooth_normals=False):
        center = numpy.reshape(numpy.array(center), (1, 3))
        self.vertices = to_vector_array(vertices)
        self.faces = to_vector_array(faces, dtype=numpy.int32)
        if auto_scale:
            current_center, current_size = self.get_bounding_cube()
            self.vertices = self.vertices - current_center[numpy.newaxis, :]
            if current_size > 0.0:
                scale = scale / float(current_size * 0.5)
        Tracer.__init__(self, position=center, scaling=scale)
        self._auto_flip_normal = auto_flip_normal
        self.shading = shading
        self.unique_tracer_id = '_%s_autoflip_%s' % (shading, auto_flip_normal)
        if shading != 'flat':
            if auto_smooth_normals:
                normals = generate_smooth_normals(self.vertices, self.faces)
            self.normals = to_vector_array(normals)
            assert self.normals.shape[0] == self.vertices.shape[0]
        else:
            assert normals is None
            self.normals = None

    @property
    def auto_flip_normal(self):
        return self._auto_flip_normal

    def get_data(self):
        vectors = [self.vertices] if self.normals is None else [self.vertices, self.normals]
        return {'vector': numpy.vstack(vectors), 'integer': numpy.ravel(self.faces)}

    def get_bounding_box(self):
        lo = self.vertices.min(axis=0)
        hi = self.vertices.max(axis=0)
        return [[lo[i], hi[i]] for i in range(3)]

    def get_bounding_cube(self):
        box = self.get_bounding_box()
        center = [sum(r) * 0.5 for r in box]
        side = max(abs(r[1] - r[0]) for r in box)
        return numpy.array(center), side

    @property
    def n_faces(self):
        return self.faces.shape[0]

    @property
    def n_vertices(self):
        return self.vertices.shape[0]

    def parameter_declarations(self):
        return ['int n_vertices', 'int n_triangles']

    def parameter_values(self):
        return [self.n_vertices, self.n_faces]
