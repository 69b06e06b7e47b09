"""Native mesh readers (csrc/mesh_io.cpp) against the Python readers that restate the
reference's mesh_formats.py, on generated files: bit-exact vertices, identical triangles,
same rejections."""
import io
import os
import time

import numpy as np
import pytest

from raytracer_amd.accelerator import read_mesh, unique_faces, ClrError
from raytracer_amd.objects import meshes


def random_polygon_file(rng, n_vertices, n_faces, header_inline):
    vertices = rng.normal(size=(n_vertices, 3)) * 10.0 ** rng.randint(-6, 6, size=(n_vertices, 1))
    lines = ['OFF %d %d 0' % (n_vertices, n_faces)] if header_inline else ['OFF', '%d %d 0' % (n_vertices, n_faces)]
    for v in vertices:
        lines.append(' '.join(repr(float(x)) for x in v) + ('  0.5 0.5 0.5' if rng.rand() < 0.1 else ''))
    for _ in range(n_faces):
        k = rng.randint(3, 7)
        lines.append('%d  %s' % (k, ' '.join(str(i) for i in rng.randint(0, n_vertices, size=k))))
    return '\n'.join(lines) + '\n'


@pytest.mark.parametrize('inline', [False, True])
def test_off_matches_python_reader(tmp_path, inline):
    text = random_polygon_file(np.random.RandomState(3), 200, 300, inline)
    path = tmp_path / 'soup.off'
    path.write_text(text)
    pv, pt = meshes.read_off(io.StringIO(text))
    nv, nt = read_mesh(str(path))
    assert np.array_equal(np.array([v[:3] for v in pv]), nv)           # same doubles, bit for bit
    assert np.array_equal(np.array(pt, dtype=np.int32), nt)


def test_zipper_matches_python_reader(tmp_path):
    rng = np.random.RandomState(4)
    v = rng.normal(size=(50, 3))
    faces = rng.randint(0, 50, size=(80, 3))
    text = ('ply\nformat ascii 1.0\ncomment zipper output\nelement vertex 50\nproperty float x\nproperty float y\n'
            'property float z\nproperty float confidence\nproperty float intensity\nelement face 80\n'
            'property list uchar int vertex_indices\nend_header\n')
    text += ''.join('%r %r %r 0.8 0.5\n' % tuple(float(c) for c in row) for row in v)
    text += ''.join('3 %d %d %d\n' % tuple(f) for f in faces)
    path = tmp_path / 'bun.ply'
    path.write_text(text)
    pv, pt = meshes.read_zipper(io.StringIO(text))
    nv, nt = read_mesh(str(path))
    assert np.array_equal(np.array(pv), nv) and np.array_equal(np.array(pt, dtype=np.int32), nt)
    assert np.array_equal(read_mesh(str(path), 'zipper')[1], nt)


def test_duplicate_faces_are_sets_of_indices():
    rng = np.random.RandomState(5)
    faces = rng.randint(0, 12, size=(400, 3))
    faces[10] = faces[3][::-1]                       # same set, other order
    faces[11] = [1, 1, 2]
    faces[12] = [2, 2, 1]                            # {1, 2} again
    want = meshes.remove_duplicate_faces([list(map(int, f)) for f in faces])
    kept = unique_faces(faces)
    assert np.array_equal(faces[kept], np.array(want))
    assert len(kept) < len(faces)


def test_bad_files_are_rejected(tmp_path):
    cases = {
        'nooff.off': 'COFF\n1 0 0\n0 0 0\n',
        'count.off': 'OFF\n3 1 0\n0 0 0\n1 0 0\n0 1 0\n4 0 1 2\n',       # says 4 indices, has 3
        'range.off': 'OFF\n3 1 0\n0 0 0\n1 0 0\n0 1 0\n3 0 1 7\n',
        'short.off': 'OFF\n3 1 0\n0 0 0\n1 0 0\n',
        'vertex.off': 'OFF\n1 0 0\n0 zero 0\n',
        'header.ply': 'ply\nelement vertex 1\n\nend_header\n0 0 0\n',     # empty line in the header
    }
    for name, text in cases.items():
        path = tmp_path / name
        path.write_text(text)
        with pytest.raises(ClrError):
            read_mesh(str(path))
    with pytest.raises(ClrError):
        read_mesh(str(tmp_path / 'missing.off'))
    with pytest.raises(RuntimeError):
        meshes.read_off(io.StringIO(cases['nooff.off']))


def test_round_trip_of_a_generated_mesh_and_speed(tmp_path):
    """The C4 input path: generated mesh -> OFF file -> native reader, exact after the round trip
    through text (9 significant digits restore a float32 exactly)."""
    from raytracer_amd.scenes.library import torus_knot_mesh
    vertices, faces = torus_knot_mesh(200, 100)
    vertices = np.asarray(vertices, dtype=np.float32)
    path = str(tmp_path / 'knot.off')
    meshes.write_off(path, vertices, faces)
    t0 = time.time()
    nv, nt = meshes.read_mesh_file(path)
    native = time.time() - t0
    assert np.array_equal(nv.astype(np.float32), vertices) and np.array_equal(nt, np.asarray(faces))
    t0 = time.time()
    pv, pt = meshes.read_off(path)
    python = time.time() - t0
    assert np.array_equal(np.array(pv), nv) and np.array_equal(np.array(pt, dtype=np.int32), nt)
    assert native < python


def test_hostile_headers_and_unseekable_files_fail_cleanly(tmp_path):
    """Counts a file cannot hold and files without a size are refused with an error, not with an
    allocation failure crossing the C boundary (which would end the Python process)."""
    big = tmp_path / 'big.off'
    big.write_text('OFF\n2000000000 5 0\n0 0 0\n')
    with pytest.raises(ClrError) as err:
        read_mesh(str(big))
    assert 'more records than the file can hold' in str(err.value)
    bigply = tmp_path / 'big.ply'
    bigply.write_text('ply\nelement vertex 3\nelement face 1999999999\nend_header\n0 0 0\n')
    with pytest.raises(ClrError):
        read_mesh(str(bigply))
    fifo = str(tmp_path / 'pipe.off')
    os.mkfifo(fifo)
    import threading

    def writer():
        with open(fifo, 'w') as f:
            f.write('OFF\n1 0 0\n0 0 0\n')
    t = threading.Thread(target=writer)
    t.start()
    with pytest.raises(ClrError) as err:
        read_mesh(fifo)
    t.join()
    assert 'not a regular file' in str(err.value)


@pytest.mark.parametrize('name', ['a.off', 'b.off', 'c.ply'])
def test_readers_match_what_the_reference_readers_returned(tmp_path, name):
    """tests/golden/host-io.npz (tools/make_golden.py host-io): generated OFF / zipper files and
    what the reference's own read_off / read_zipper / remove_duplicate_faces
    (/root/reference/mesh_formats.py:14-82) returned for them.  The native readers of
    csrc/mesh_io.cpp and this package's Python readers must return the same arrays."""
    from conftest import GOLDEN_DIR
    fx = np.load(os.path.join(GOLDEN_DIR, 'host-io.npz'))
    key = 'mesh_' + name.replace('.', '_')
    text = fx[key + '_text'].tobytes().decode('ascii')
    path = tmp_path / name
    path.write_text(text)
    want_vertices, want_faces, want_unique = fx[key + '_vertices'], fx[key + '_faces'], fx[key + '_unique_faces']
    assert len(want_unique) < len(want_faces)                    # the files do hold repeated faces
    vertices, triangles = read_mesh(str(path))
    assert vertices.dtype == np.float64 and np.array_equal(vertices, want_vertices)
    assert np.array_equal(triangles, want_faces)
    assert np.array_equal(triangles[unique_faces(triangles)], want_unique)
    reader = meshes.read_off if name.endswith('.off') else meshes.read_zipper
    py_vertices, py_faces = reader(io.StringIO(text))
    assert np.array_equal(np.array([v[:3] for v in py_vertices]), want_vertices)
    assert [len(v) for v in py_vertices] == list(fx[key + '_vertex_columns'])
    assert np.array_equal(np.array(py_faces), want_faces)
    assert np.array_equal(np.array(meshes.remove_duplicate_faces(py_faces)), want_unique)
