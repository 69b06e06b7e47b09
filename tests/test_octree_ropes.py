"""Neighbour links ("ropes") of the flattened octree (objects/meshes.py: octree_ropes; traversal in
hip_templates/shapes/meshes.hip).  CPU tests: the table itself, and a binary32 emulation of the
two ways to find the next leaf - the reference's descent from the common ancestor
(/root/reference/cl_templates/objects/Octree.cl:120-160) and the rope with its check - which must
visit the same leaves with the same exit distances, ray by ray."""
import numpy as np
import pytest

from raytracer_amd.objects.meshes import octree_cells, octree_ropes
from raytracer_amd.scenes.library import scene_mesh

f32 = np.float32


def knot_tree(n_u, n_v, **kw):
    scene = scene_mesh(n_u, n_v, layout='flat', overlap='sat', pooled=True, **dict({'ropes': True}, **kw))
    return [o.tracer for o in scene.objects if o.name == 'knot'][0]


def test_rope_table_is_consistent():
    knot = knot_tree(120, 60)
    assert knot.ropes and knot.pooled
    nodes = np.asarray(knot._native['flat_nodes']).reshape(-1, 2)
    ropes, parents = octree_ropes(nodes)
    depth, cell, parent = octree_cells(nodes)
    assert np.array_equal(parents, parent) and parents[0] == -1 and (parents[1:] >= 0).all()
    assert (nodes[parents[1:], 0] == -1).all()
    for face in range(6):
        axis, step = face // 2, 2 * (face % 2) - 1
        target = ropes[:, face, 0]
        have = target >= 0
        idx, d_r = target[have] & 0x7ffffff, target[have] >> 27
        assert (d_r == depth[idx]).all() and (d_r <= depth[have]).all()
        beyond = cell[have].copy()
        beyond[:, axis] += step
        assert ((beyond >> (depth[have] - d_r)[:, None]) == cell[idx]).all()      # R covers the cell beyond the face
        assert (nodes[idx[d_r < depth[have]], 0] >= 0).all()                       # a larger neighbour is a leaf
        assert (ropes[have, face, 1:] == nodes[idx]).all()                         # R's record rides along
        edge = np.where(step < 0, 0, (1 << depth[~have]) - 1)
        assert (cell[~have][:, axis] == edge).all()                                # no rope only on the root cube
    # the data the kernel reads: tables behind the face ids, offsets among the declared parameters
    data = knot.get_data()
    names = [d.split()[1] for d in knot.parameter_declarations()]
    values = dict(zip(names, knot.parameter_values()))
    octree_data = data['integer'][values['face_data_length']:]
    assert np.array_equal(octree_data[values['flat_rope_offset']:values['flat_rope_offset'] + ropes.size], ropes.ravel())
    assert np.array_equal(octree_data[values['flat_parent_offset']:values['flat_parent_offset'] + parents.size], parents)
    with pytest.raises(RuntimeError):
        knot_tree(60, 30, max_depth=10, ropes=True)


SHORTCUT = {'sure': 0, 'checked': 0}


def leaf_sequence(nodes, ropes, parents, root_origin, root_size, origin, ray, t_limit, use_ropes, max_steps=400,
                  margin=None):
    """The pooled kernel's phase A in binary32, one ray: [(leaf node, exit distance)] and the
    number of hops that fell back to the reference's way."""
    origin, ray, root_origin = (np.asarray(v, f32) for v in (origin, ray, root_origin))
    root_size = f32(root_size)
    with np.errstate(divide='ignore', invalid='ignore'):
        slopes = f32(1.0) / ray
    neg = (slopes < 0).astype(int)
    far_walls = (1 - neg).astype(f32)
    sign = 1 - 2 * neg
    rel = root_origin - origin
    t_in = slopes * (rel + neg.astype(f32) * root_size)
    t_out = slopes * (rel + far_walls * root_size)
    seg_begin, seg_end = f32(max(t_in)), f32(min(t_out))
    if seg_begin > seg_end or seg_end < 0 or seg_begin > t_limit:
        return [], 0
    c, b, depth, node_size = np.zeros(3, int), np.zeros(3, int), 0, root_size
    stack = {0: 0}
    node_at, record = 0, nodes[0]
    rope_target, hop_cell, leaf_depth, left_leaf, delta = -1, None, 0, 0, 0
    visited, fallbacks = [], 0
    for _ in range(max_steps):
        ray_pos = origin + seg_begin * ray
        if use_ropes and rope_target >= 0:
            rope_depth, same_way, size_j = rope_target >> 27, True, node_size
            for j in range(depth, rope_depth):
                up = leaf_depth - j
                cell_origin = root_origin + (hop_cell >> up).astype(f32) * size_j
                rc = ray_pos - cell_origin
                chosen = 1 - (rc < f32(0.5) * size_j).astype(int)
                same_way = same_way and np.array_equal(chosen, (hop_cell >> (up - 1)) & 1)
                size_j = size_j * f32(0.5)
            if margin is not None:
                # the kernel's shortcut: inside R's cell by more than the margin => no need to look
                up = leaf_depth - rope_depth
                size_r = node_size * f32(2.0) ** f32(depth - rope_depth)
                rc = ray_pos - (root_origin + (hop_cell >> up).astype(f32) * size_r)
                sure = True
                for axis in range(3):
                    if axis == hop_face >> 1:
                        wall = f32(0.0) if hop_face & 1 else size_r
                        sure = sure and abs(rc[axis] - wall) < f32(0.25) * size_r
                    else:
                        sure = sure and margin < rc[axis] < size_r - margin
                SHORTCUT['sure' if sure else 'checked'] += 1
                assert same_way or not sure, "the shortcut may only say yes where the comparisons agree"
            if same_way:
                depth = rope_depth
                c = hop_cell >> (leaf_depth - rope_depth)
                b = c & 1
                node_size, node_at = size_j, rope_target & 0x7ffffff
            else:
                fallbacks += 1
                at = left_leaf
                for _j in range(depth, leaf_depth):
                    at = parents[at]
                node_at = at + delta
                record = nodes[node_at]
        while True:
            head, payload = int(record[0]), int(record[1])
            node_origin = root_origin + c.astype(f32) * node_size
            if head >= 0:
                break
            rc = ray_pos - node_origin
            b = 1 - (rc < f32(0.5) * node_size).astype(int)
            child = b[0] * 4 + b[1] * 2 + b[2]
            node_at = payload + child
            stack[depth + 1] = node_at
            c = c * 2 + b
            node_size = node_size * f32(0.5)
            depth += 1
            record = nodes[node_at]
        t_out = slopes * (node_origin - origin + far_walls * node_size)
        seg_end = f32(min(t_out))
        visited.append((node_at, float(seg_end)))
        hop = np.zeros(3, int)
        for axis in range(3):
            if seg_end == t_out[axis]:
                hop[axis] = sign[axis]
                face = 2 * axis + 1 - neg[axis]
                break
        leaf_at, leaf_depth, hop_cell = node_at, depth, c + hop
        hop_face = face if hop.any() else 0
        while depth > 0:
            b = b + hop
            if ((b >= 0) & (b <= 1)).all():
                break
            depth -= 1
            c = c >> 1
            b = c & 1
            node_size = node_size * f32(2.0)
        if depth == 0:
            break
        old = (c[0] & 1) * 4 + (c[1] & 1) * 2 + (c[2] & 1)
        c = (c & ~1) | b
        delta = b[0] * 4 + b[1] * 2 + b[2] - old
        left_leaf = leaf_at
        sibling = stack[depth] + delta
        stack[depth] = sibling
        seg_begin = seg_end
        if seg_begin > t_limit:
            break
        if use_ropes:
            assert hop.any()
            rope_target, record = int(ropes[leaf_at, face, 0]), ropes[leaf_at, face, 1:]
        else:
            node_at, record = sibling, nodes[sibling]
    return visited, fallbacks


@pytest.mark.parametrize('grid,n_rays', [((120, 60), 250), ((300, 150), 120)])
def test_ropes_visit_the_leaves_of_the_reference_descent(grid, n_rays):
    knot = knot_tree(*grid)
    nodes = np.asarray(knot._native['flat_nodes']).reshape(-1, 2)
    ropes, parents = octree_ropes(nodes)
    root_origin, root_size = np.asarray(knot.root.origin, f32), f32(knot.root.size)
    rng = np.random.RandomState(3)
    total_leaves = total_fallbacks = 0
    for k in range(n_rays):
        # from outside towards the tree, from inside it, and along cell edges (the rare way)
        centre = root_origin + f32(0.5) * root_size
        if k % 3 == 0:
            origin = centre + rng.normal(size=3) * float(root_size)
            ray = centre + rng.uniform(-0.4, 0.4, 3) * float(root_size) - origin
        elif k % 3 == 1:
            origin = root_origin + rng.uniform(0.05, 0.95, 3) * float(root_size)
            ray = rng.normal(size=3)
        else:
            # starts on a corner of the finest grid and runs along an axis: on cell boundaries all the way
            origin = root_origin + rng.randint(1, 63, 3) * float(root_size) / 64
            ray = np.zeros(3)
            ray[rng.randint(3)] = rng.choice([-1.0, 1.0])
            ray = ray + rng.normal(size=3) * 1e-9
        ray = (ray / np.linalg.norm(ray)).astype(f32)
        want, _ = leaf_sequence(nodes, ropes, parents, root_origin, root_size, origin, ray, f32(1e3), False)
        margin = f32(float.fromhex(knot.rope_margin_literal.rstrip('f')))
        got, fallbacks = leaf_sequence(nodes, ropes, parents, root_origin, root_size, origin, ray, f32(1e3), True,
                                       margin=margin)
        assert got == want, (k, origin, ray)
        total_leaves += len(want)
        total_fallbacks += fallbacks
    assert total_leaves > 10 * n_rays
    print('%d leaves visited, %d hops took the fallback; shortcut: %s' % (total_leaves, total_fallbacks, SHORTCUT))
    assert SHORTCUT['sure'] > 5 * SHORTCUT['checked'] > 0


def test_two_level_records_describe_the_same_tree():
    """objects/meshes.py: octree_two_level - breadth-first records in which an inner node names the
    children blocks of its inner children.  Every root-to-leaf path of child numbers ends in the
    same (triangle count, first triangle record) as in the builder's records, reached through the
    two-level rule of the pooled traversal (hip_templates/shapes/meshes.hip), and the child's own
    record names the block the parent's record promises."""
    from raytracer_amd.objects.meshes import octree_two_level
    plain = np.asarray(knot_tree(120, 60, two_level=False)._native['flat_nodes']).reshape(-1, 2)
    wide = octree_two_level(plain)
    assert wide.shape == plain.shape and wide.dtype == np.int32

    def leaves_plain():
        todo = [(0, ())]
        while todo:
            at, path = todo.pop()
            head, payload = (int(v) for v in plain[at])
            if head >= 0:
                yield path, (head, payload)
            else:
                todo += [(payload + c, path + (c,)) for c in range(8)]

    def leaves_wide():
        todo = [(0, ())]
        while todo:
            at, path = todo.pop()
            head, payload = (int(v) for v in wide[at])
            if head >= 0:
                yield path, (head, payload)
                continue
            first_child, inner_children = head & 0x7fffff, (head >> 23) & 255
            for c in range(8):
                if (inner_children >> c) & 1:
                    block = payload + 8 * bin(inner_children & ((1 << c) - 1)).count('1')
                    assert wide[first_child + c, 0] < 0 and (int(wide[first_child + c, 0]) & 0x7fffff) == block
                    todo += [(block + c2, path + (c, c2)) for c2 in range(8)]
                else:
                    assert wide[first_child + c, 0] >= 0
                    todo.append((first_child + c, path + (c,)))

    a, b = dict(leaves_plain()), dict(leaves_wide())
    assert len(a) > 1000 and a == b
