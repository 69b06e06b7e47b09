"""TriangleMesh (every face tested, reference objects/triangle_mesh.py:5-117) and Octree over a
TriangleMesh (reference objects/octree.py:4-233).

``build()`` restates the reference's breadth-first construction (bounding-sphere and
face-plane overlap test per node, leaf when few "small" faces remain or at max depth) with a
per-node vectorised test instead of the reference's dense (nodes x faces) matrices, and
``get_data()`` emits the reference's integer stream (faces, then post-order node payloads,
root header last).  Face indices are tracked explicitly, so the reference's latent
``active_faces`` indexing hazard (octree.py:215-231) is not reproduced; results coincide
whenever no face drops out of the active set, e.g. the scene-test cube.
"""
import os
import numpy
from .base import Tracer


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
    outline_candidate = True

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


EMPTY_HEADER = [0x100, -99999]


POOLED_FROM_FACES = 30000     # meshes from this size on get the pooled triangle tests by default


class Octree(Tracer):
    outline_candidate = True

    MAX_DEPTH = 7

    def __init__(self, triangle_mesh, max_depth=3, max_faces_per_leaf=5, builder='python',
                 layout='stream', max_steps=300, pooled=None, overlap='reference', ropes=None,
                 two_level=None):
        """builder: 'python' (restatement of the reference's construction) or 'native' (C++
        builder of libclray_hip.so, same tree, any size, max_depth up to 20).
        layout: 'stream' = the reference's integer stream; 'flat' = flattened nodes + per-leaf
        pre-gathered triangle records (needs the native builder).
        max_steps: leaf-step budget of the traversal (300 in the reference).
        pooled (flat layout only; default: from POOLED_FROM_FACES faces on): the triangle tests of all lanes of a wave are
        pooled and dealt out evenly, so lanes whose ray left the mesh or sits in a small leaf
        work on the leaves of the others (same tests, same results).
        overlap (native builder): 'reference' = the reference's bounding-sphere rule for filing a
        face under a node (objects/octree.py:201-233), 'sat' = the triangle against the node's box:
        same hits, a fraction of the leaf records (for large meshes).
        ropes / two_level (pooled traversal only): how the next leaf is found - through a table of
        neighbour links (measured slower, off), or with node records that take the descent down two
        levels per memory access (default with pooled tests); same leaves, same hits either way."""
        if overlap not in ('reference', 'sat'):
            raise ValueError("overlap must be 'reference' or 'sat'")
        if overlap != 'reference' and builder == 'python' and layout != 'flat':
            raise RuntimeError("overlap='sat' needs the native builder")
        self.overlap = overlap
        if layout == 'flat':
            builder = 'native'
        # measured on the torus knot (tools/mesh_sweep.sh, Msamples/s pooled / per lane): 1.4 k faces
        # 4442 / 5863, 10 k 3065 / 3403, 90 k 2510 / 1994, 1 M 1661 / 1085
        if pooled is None:
            pooled = layout == 'flat' and len(triangle_mesh.faces) >= POOLED_FROM_FACES
        self.pooled = bool(pooled)
        # pooled traversal only: the next leaf through a table of neighbour links instead of a
        # descent from the common ancestor (octree_ropes below; same leaves, same hits).  The packed
        # cell coordinates of the kernel hold 9 bits per axis.
        # Measured on the 1 M-triangle knot (profiles/r03_mesh_ropes.log): bit-identical and 3 % SLOWER
        # (6 % without the shortcut that skips the level-by-level check) - the node records of the
        # descent (3.6 MB) live in L2, the rope table (32 MB) does not, so one fetch from it costs
        # about what the three to five L2 hits of the descent cost.  Off unless asked for.
        if ropes is None:
            ropes = False
        if ropes and not (self.pooled and max_depth <= ROPES_MAX_DEPTH):
            raise RuntimeError("ropes need the pooled traversal and max_depth <= %d" % ROPES_MAX_DEPTH)
        self.ropes = bool(ropes)
        # pooled traversal only: node records in breadth-first order that also name the children
        # blocks of a node's inner children (octree_two_level below), so that the descent goes down
        # two levels per dependent memory access.  Same nodes, same leaves, same table size.
        # Measured on the 1 M-triangle knot: bit-identical, 1834 -> 1896 Msamples/s
        # (profiles/r03_mesh_two_level.log).  Default of the pooled traversal; a tree with more
        # nodes than the record's 23-bit field holds keeps the plain records (_native_data).
        self._two_level_by_default = two_level is None
        if two_level is None:
            two_level = self.pooled and not self.ropes
        if two_level and not (self.pooled and not self.ropes):
            raise RuntimeError("two-level records need the pooled traversal without ropes")
        self.two_level = bool(two_level)
        # development: code-generation variants under measurement (CLR_VARIANTS, as Renderer.variants)
        self.variants = frozenset(filter(None, os.environ.get('CLR_VARIANTS', '').split(',')))
        # development: cycle counters per phase of the pooled traversal (tools/mesh_phases.py)
        self.profile_phases = os.environ.get('CLR_MESH_PROFILE') == '1'
        if self.pooled and layout != 'flat':
            raise RuntimeError("pooled triangle tests need the flat layout")
        self.builder, self.layout, self.max_steps = builder, layout, max_steps
        mesh_pose = triangle_mesh.coordinates
        self.triangle_mesh = triangle_mesh
        Tracer.__init__(self)          # assigns an identity pose through the setter below
        self.coordinates = mesh_pose   # ... so put the mesh's own pose back
        self.max_depth = max_depth
        self.max_faces_per_leaf = max_faces_per_leaf
        if self.max_depth >= self.__class__.MAX_DEPTH and builder == 'python':
            raise RuntimeError("max_depth >= MAX_DEPTH")
        self.build()
        if layout != 'stream' or max_steps != 300 or self.max_depth >= self.__class__.MAX_DEPTH:
            self.unique_tracer_id = '_%s%s%s%s_d%d_s%d' % (layout, '_pooled' if self.pooled else '',
                                                           '_ropes' if self.ropes else '',
                                                           '_two' if self.two_level else '', self.levels, max_steps)

    @property
    def rope_margin_literal(self):
        """16 units in the last place of the largest coordinate of the root cube, as a HIP literal
        (the pooled traversal's shortcut for hops through ropes, meshes.hip)."""
        largest = float(numpy.max(numpy.abs(numpy.asarray(self.root.origin, dtype=numpy.float64))) + self.root.size)
        return float(numpy.float32(16.0 * numpy.spacing(numpy.float32(largest)))).hex() + 'f'

    @property
    def levels(self):
        """Entries of the traversal stack (reference: MAX_DEPTH + 1)."""
        return max(self.__class__.MAX_DEPTH + 1, self.max_depth + 2)

    @property
    def auto_flip_normal(self):
        return self.triangle_mesh.auto_flip_normal

    @property
    def shading(self):
        # The reference's shared normal template tests `obj.shading` on the Octree itself,
        # where the attribute does not exist; Jinja's Undefined then selects the smooth
        # branch, which reads vertex normals past the mesh's own data.  Here the octree
        # shades like the mesh it wraps.
        return self.triangle_mesh.shading

    @property
    def coordinates(self):
        return self.triangle_mesh.coordinates

    @coordinates.setter
    def coordinates(self, value):
        # Tracer.__init__ assigns a fresh identity before the mesh pose is restored
        self.triangle_mesh.coordinates = value

    class Node:
        child_order = [[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)]

        def __init__(self, origin, size, parent=None):
            self.children = []
            self.faces = []
            self.origin = origin
            self.size = size
            self.parent = parent

        def is_empty(self):
            return len(self.faces) == 0

        def is_leaf(self):
            return len(self.children) == 0

        @property
        def center(self):
            return numpy.array([self.size * 0.5] * 3) + self.origin

        def get_bounding_sphere(self):
            return self.center, numpy.sqrt(3) * self.size * 0.5

        def get_children(self):
            if not self.children:
                for xyz in Octree.Node.child_order:
                    self.children.append(Octree.Node(
                        self.origin + numpy.array(xyz) * 0.5 * self.size, self.size * 0.5, self))
            return self.children

    def build(self):
        if self.builder == 'native':
            return self._build_native()
        mesh = self.triangle_mesh
        self.center, self.size = mesh.get_bounding_cube()
        self.center = numpy.asarray(self.center, dtype=numpy.float64)
        self.size = float(self.size) * 1.01
        self.origin = numpy.array(self.center) - numpy.array([self.size * 0.5] * 3)
        self.root = Octree.Node(self.origin, self.size)

        face_centers, face_radii = face_bounding_spheres(mesh.vertices, mesh.faces)
        face_normals, face_h = face_planes(mesh.vertices, mesh.faces)

        self.leaves = []
        depth = 0
        active_nodes = [self.root]
        active_faces = numpy.arange(mesh.n_faces)
        while active_nodes:
            touched = numpy.zeros(len(active_faces), dtype=bool)
            for node in active_nodes:
                c, r = node.get_bounding_sphere()
                gap = numpy.sqrt(numpy.sum((c[None, :] - face_centers[active_faces]) ** 2, 1))
                gap = gap - r - face_radii[active_faces]
                plane_gap = numpy.abs(face_normals[active_faces] @ c - face_h[active_faces]) - r
                hit = numpy.maximum(gap, plane_gap) < 0
                node.faces.extend(int(f) for f in active_faces[hit])
                touched |= hit
            active_faces = active_faces[touched]

            next_nodes = []
            for node in active_nodes:
                n_small = sum(1 for f in node.faces if face_radii[f] < node.size * 0.5)
                if n_small <= self.max_faces_per_leaf or depth == self.max_depth:
                    self.leaves.append(node)
                else:
                    node.faces = []
                    next_nodes += node.get_children()
            active_nodes = next_nodes
            depth += 1

    def _build_native(self):
        import types
        from ..accelerator import build_octree
        mesh = self.triangle_mesh
        self._native = build_octree(mesh.vertices, mesh.faces, self.max_depth, self.max_faces_per_leaf,
                                    overlap=self.overlap)
        self.root = types.SimpleNamespace(origin=self._native['root_origin'], size=self._native['root_size'])
        self.origin, self.size = self.root.origin, self.root.size
        self.n_leaves = self._native['n_leaves']
        if self.two_level and numpy.asarray(self._native['flat_nodes']).size // 2 >= TWO_LEVEL_MAX_NODES:
            if not self._two_level_by_default:
                raise RuntimeError("two-level records hold up to 2^23 nodes")
            self.two_level = False

    def _native_data(self):
        native = self._native
        data = self.triangle_mesh.get_data()
        faces = numpy.ravel(data['integer']).astype(numpy.int32)
        if self.layout == 'flat':
            vectors = numpy.zeros((numpy.asarray(data['vector']).shape[0], 4), dtype=numpy.float32)
            vectors[:, :3] = data['vector']
            self.flat_tri_offset = vectors.shape[0]
            # per (leaf, face) record: v0, e1, e2 = 9 floats = 36 B, packed back to back; the
            # face ids (needed only for the hit and for the self-intersection test) go to the
            # integer data, so a triangle test reads 36 B instead of 48
            recs = numpy.asarray(native['flat_tris'], dtype=numpy.float32).reshape(-1, 3, 4)
            face_ids = numpy.ascontiguousarray(recs[:, 0, 3]).view(numpy.int32)
            packed = numpy.ascontiguousarray(recs[:, :, :3]).reshape(-1)
            if self.pooled and (packed.size * 4 >= 1 << 32 or numpy.asarray(native['flat_nodes']).size * 4 >= 1 << 32):
                raise RuntimeError("the pooled traversal addresses its tables with 32-bit byte offsets: under 4 GB each")
            packed = numpy.concatenate([packed, numpy.zeros((-packed.size) % 4, dtype=numpy.float32)])
            data['vector'] = numpy.vstack([vectors, packed.reshape(-1, 4)])
            nodes = (octree_two_level(native['flat_nodes']) if self.two_level else native['flat_nodes']).ravel()
            self.flat_ids_offset = nodes.size
            tables = [faces, nodes, face_ids]
            self.flat_rope_offset = self.flat_parent_offset = 0
            if self.ropes:
                ropes, parents = octree_ropes(native['flat_nodes'])
                self.flat_rope_offset = nodes.size + face_ids.size
                self.flat_parent_offset = self.flat_rope_offset + ropes.size
                tables += [ropes.ravel(), parents]
            data['integer'] = numpy.concatenate(tables)
            self.root_data_offset = 0
        else:
            self.flat_tri_offset = self.flat_ids_offset = 0
            self.root_data_offset = int(native['root_data_offset'])
            data['integer'] = numpy.concatenate([faces, native['stream']])
        return data

    def get_data(self):
        """Serialise: mesh faces (3N ints) followed by the node stream."""
        if self.builder == 'native':
            return self._native_data()
        def write_node(node, stream):
            child_mask = 0
            if node.is_leaf():
                if node.is_empty():
                    return list(EMPTY_HEADER)
                payload = [len(node.faces)] + list(node.faces)
            else:
                payload = []
                for i, child in enumerate(node.children):
                    if child.is_empty() and child.is_leaf():
                        header = list(EMPTY_HEADER)
                    else:
                        child_mask |= 1 << i
                        header = write_node(child, stream)
                    payload += header
                if child_mask == 0:
                    return list(EMPTY_HEADER)
            offset = len(stream)
            stream.extend(payload)
            return [child_mask, offset]

        stream = []
        root_header = write_node(self.root, stream)
        self.root_data_offset = len(stream)
        stream.extend(root_header)

        data = self.triangle_mesh.get_data()
        data['integer'] = numpy.concatenate(
            [numpy.ravel(data['integer']), numpy.array(stream, dtype=numpy.int64)]).astype(numpy.int32)
        return data

    @property
    def total_faces(self):
        return len(self.triangle_mesh.faces)

    def parameter_declarations(self):
        decls = ['int n_vertices', 'int n_triangles', 'int root_data_offset',
                 'int face_data_length', 'float3 root_origin', 'float root_size']
        if self.layout == 'flat':
            decls += ['int flat_tri_offset', 'int flat_ids_offset']
        if self.ropes:
            decls += ['int flat_rope_offset', 'int flat_parent_offset']
        return decls

    def parameter_values(self):
        if not hasattr(self, 'root_data_offset'):
            self.get_data()
        values = [self.triangle_mesh.n_vertices, self.triangle_mesh.n_faces,
                  self.root_data_offset, self.total_faces * 3, self.root.origin, self.root.size]
        if self.layout == 'flat':
            values += [self.flat_tri_offset, self.flat_ids_offset]
        if self.ropes:
            values += [self.flat_rope_offset, self.flat_parent_offset]
        return values


ROPES_MAX_DEPTH = 9
TWO_LEVEL_MAX_NODES = 1 << 23


def octree_two_level(flat_nodes):
    """The node records of a flattened octree (rows (head, payload), octree_cells above) in
    breadth-first order and with inner nodes written as
        head    = 1 << 31 | (bit c set: child c is an inner node) << 23 | first child record
        payload = first record of the children of the first inner child,
    the children blocks of a node's inner children lying back to back in child order (breadth-first
    order does that).  From one such record the traversal knows where the grandchild under a point
    is without reading the child's record (hip_templates/shapes/meshes.hip, two_level): one
    dependent memory access per two levels of the reference's descent
    (cl_templates/objects/Octree.cl:120-160), the same nodes visited.  Leaves keep (number of
    triangles, first triangle record)."""
    nodes = numpy.asarray(flat_nodes, dtype=numpy.int64).reshape(-1, 2)
    n = nodes.shape[0]
    if n >= TWO_LEVEL_MAX_NODES:
        raise RuntimeError("octree_two_level: more than 2^23 nodes")
    is_inner = nodes[:, 0] == -1
    order = [numpy.array([0], dtype=numpy.int64)]
    while order[-1].size:
        level = order[-1]
        inner = level[is_inner[level]]
        order.append((nodes[inner, 1][:, None] + numpy.arange(8)[None, :]).ravel())
    order = numpy.concatenate(order)
    if order.size != n or numpy.unique(order).size != n:
        raise RuntimeError("octree_two_level: the records do not form one tree")
    new_index = numpy.empty(n, dtype=numpy.int64)
    new_index[order] = numpy.arange(n)

    out = nodes[order].copy()
    inner = numpy.nonzero(is_inner[order])[0]                 # new indices of the inner nodes
    old_children = nodes[order[inner], 1][:, None] + numpy.arange(8)[None, :]
    first_child = new_index[old_children[:, 0]]
    assert numpy.array_equal(new_index[old_children], first_child[:, None] + numpy.arange(8)[None, :])
    child_inner = is_inner[old_children]
    mask = (child_inner << numpy.arange(8)[None, :]).sum(axis=1)
    # children blocks of the inner children: back to back, in child order
    blocks = numpy.where(child_inner, new_index[numpy.where(child_inner, nodes[old_children, 1], 0)], -1)
    rank = numpy.cumsum(child_inner, axis=1) - child_inner
    first_block = numpy.where(child_inner.any(axis=1), blocks.max(axis=1, initial=-1, where=child_inner & (rank == 0)), 0)
    assert numpy.array_equal(blocks[child_inner], (first_block[:, None] + 8 * rank)[child_inner])
    out[inner, 0] = -(1 << 31) + (mask << 23) + first_child
    out[inner, 1] = first_block
    return out.astype(numpy.int32)


def octree_cells(flat_nodes):
    """(depth, x, y, z, parent) of every node of a flattened octree (rows (head, payload): head -1 =
    inner node with its 8 children from record `payload` on, in the order x*4 + y*2 + z; node 0 =
    root)."""
    nodes = numpy.asarray(flat_nodes, dtype=numpy.int64).reshape(-1, 2)
    n = nodes.shape[0]
    depth = numpy.zeros(n, numpy.int64)
    cell = numpy.zeros((n, 3), numpy.int64)
    parent = numpy.full(n, -1, numpy.int64)
    frontier = numpy.array([0])
    while frontier.size:
        inner = frontier[nodes[frontier, 0] == -1]
        children = (nodes[inner, 1][:, None] + numpy.arange(8)[None, :])
        bits = numpy.array([[(k >> 2) & 1, (k >> 1) & 1, k & 1] for k in range(8)])
        depth[children] = depth[inner][:, None] + 1
        cell[children] = cell[inner][:, None, :] * 2 + bits[None, :, :]
        parent[children] = inner[:, None]
        frontier = children.ravel()
    return depth, cell, parent


def octree_ropes(flat_nodes):
    """Neighbour links of a flattened octree, for the pooled traversal (hip_templates/shapes/
    meshes.hip).  ropes[node, face] = (R | depth of R << 27, head of R, payload of R), face =
    2 * axis + (1 if towards larger coordinates), where R is the deepest node that covers the cell
    beyond that face of `node` (same depth as `node`) and is not deeper than `node`: the neighbour
    of equal size if the tree has it, else the larger leaf that contains it; -1 where the face lies
    on the root cube.  Also returns the parent of every node (-1 for the root).  No counterpart in
    the reference, whose traversal finds the neighbour by a descent from the common ancestor
    (cl_templates/objects/Octree.cl:120-160); the kernel checks that this descent would arrive at R."""
    nodes = numpy.asarray(flat_nodes, dtype=numpy.int64).reshape(-1, 2)
    n = nodes.shape[0]
    if n >= 1 << 27:
        raise RuntimeError("octree_ropes: more than 2^27 nodes")
    depth, cell, parent = octree_cells(nodes)
    max_depth = int(depth.max())
    if max_depth > ROPES_MAX_DEPTH:
        raise RuntimeError("octree_ropes: tree deeper than %d" % ROPES_MAX_DEPTH)
    key = (cell[:, 0] << 40) | (cell[:, 1] << 20) | cell[:, 2]
    levels = []
    for d in range(max_depth + 1):
        at = numpy.nonzero(depth == d)[0]
        order = numpy.argsort(key[at], kind='stable')
        levels.append((key[at][order], at[order]))
    ropes = numpy.full((n, 6, 3), -1, numpy.int64)
    for face in range(6):
        axis, step = face // 2, 2 * (face % 2) - 1
        beyond = cell.copy()
        beyond[:, axis] += step
        inside = (beyond[:, axis] >= 0) & (beyond[:, axis] < (1 << depth))
        target = numpy.full(n, -1, numpy.int64)
        open_ = inside.copy()
        for d in range(max_depth, -1, -1):          # deepest first
            sel = numpy.nonzero(open_ & (depth >= d))[0]
            if not sel.size:
                continue
            up = (depth[sel] - d)[:, None]
            c = beyond[sel] >> up
            k = (c[:, 0] << 40) | (c[:, 1] << 20) | c[:, 2]
            keys, ids = levels[d]
            pos = numpy.minimum(numpy.searchsorted(keys, k), keys.size - 1)
            found = keys[pos] == k
            target[sel[found]] = ids[pos[found]]
            open_[sel[found]] = False
        assert not open_.any(), "a cell inside the root cube is covered by some node"
        have = target >= 0
        ropes[have, face, 0] = target[have] | (depth[target[have]] << 27)
        ropes[have, face, 1:] = nodes[target[have]]
    return ropes.astype(numpy.int32), parent.astype(numpy.int32)


def face_bounding_spheres(vertices, faces):
    tri = vertices[faces].astype(numpy.float64)   # (F, 3 corners, 3)
    centers = tri.mean(axis=1)
    radii = numpy.sqrt(numpy.max(numpy.sum((tri - centers[:, None, :]) ** 2, axis=2), axis=1))
    return centers, radii


def face_planes(vertices, faces):
    v = vertices.astype(numpy.float64)
    p0 = v[faces[:, 0]]
    normals = numpy.cross(v[faces[:, 1]] - p0, v[faces[:, 2]] - p0)
    normals = normals / numpy.sqrt(numpy.sum(normals ** 2, 1))[:, None]
    return normals, numpy.sum(p0 * normals, 1)


# ---------------------------------------------------------------- mesh file readers
# OFF and Stanford "zipper" (ASCII PLY) readers, polygons fan-triangulated; same functions and
# return values (vertex list, triangle list) as the reference's mesh_formats.py:3-82.
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


# ---- native counterparts (csrc/mesh_io.cpp) for large files: same records, numpy arrays out
def read_mesh_file(path, fmt=None, dedupe=False):
    """(vertices (n, 3) float64, triangles (m, 3) int32) of an OFF or zipper file, parsed by the
    native reader; dedupe=True also drops duplicate faces as remove_duplicate_faces does."""
    from ..accelerator import read_mesh, unique_faces
    vertices, triangles = read_mesh(path, fmt)
    if dedupe:
        triangles = triangles[unique_faces(triangles)]
    return vertices, triangles


def write_off(path, vertices, faces):
    """Plain OFF writer (the reference only reads), used to build large synthetic inputs."""
    vertices = numpy.asarray(vertices, dtype=numpy.float64)
    faces = numpy.asarray(faces, dtype=numpy.int64)
    with open(path, 'w') as f:
        f.write('OFF\n%d %d 0\n' % (vertices.shape[0], faces.shape[0]))
        numpy.savetxt(f, vertices, fmt='%.9g')
        numpy.savetxt(f, numpy.hstack([numpy.full((faces.shape[0], 1), faces.shape[1]), faces]), fmt='%d')
