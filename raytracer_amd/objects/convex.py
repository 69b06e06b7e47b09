"""Convex bodies as intersections of components.

ConvexIntersection and its component classes (half-space, slab, sphere, infinite cylinder,
semi-infinite cone), the literal-baking Unique/Fixed variants, and the shapes built from them:
Cylinder, Cone, Parallelepiped and the Platonic solids.  Same constructors as the reference's
objects/convex_intersection.py:5-120, planes.py:23-54, sphere.py:39-49, cylinder.py:5-39,
cone.py:5-43 and polyhedra.py:5-102."""
import math
import numpy
from .base import Tracer, next_unique_id
from ..geometry import normalize_tuple, vec_norm
from ..geometry import Affine, rotation_aligning_vectors

def float_literal(value):
    """A C float literal for a baked-in scalar."""
    text = repr(float(value))
    if 'e' not in text and '.' not in text and 'inf' not in text and 'nan' not in text:
        text += '.0'
    return text + 'f'


def vec_literal(vec):
    return "make_vec3(%s)" % ", ".join(float_literal(c) for c in numpy.ravel(vec))


class ConvexIntersection(Tracer):
    """Intersection of convex bodies; run-time parameters, one function per component list."""
    speculation_friendly = True

    class Component(Tracer):
        def __init__(self, pos=(0, 0, 0)):
            Tracer.__init__(self, position=pos)

    def __init__(self, origin, components):
        Tracer.__init__(self, position=origin)
        self.components = components
        self.unique_tracer_id = '_' + '_'.join(c.__class__.__name__ for c in components)

    def template_name(self):
        return 'ConvexIntersection'

    def make_functions(self, template_env):
        funcs = dict(Tracer.make_functions(self, template_env))
        for component in self.components:
            funcs.update(component.make_functions(template_env))
        return funcs

    @property
    def convex(self):
        return True

    # ---- template helpers
    def face_ranges(self):
        """[(component index, component, first face id, end face id)]: the faces ("subobjects")
        of the components are numbered consecutively from 1, 0 meaning "no face"."""
        ranges, first = [], 0
        for i, c in enumerate(self.components):
            ranges.append((i, c, first, first + c.n_subobjects))
            first += c.n_subobjects
        return ranges

    def component_position_expr(self, component_idx, base):
        return '%s - _pos_%d' % (base, component_idx)

    def component_parameter_string(self, component_idx):
        names = self.components[component_idx].parameter_names()
        return ''.join(', %s_%d' % (n, component_idx) for n in names)

    def component_tracer_call_params(self, component_idx):
        return '%s, ray, &c_enter, &c_exit, &c_face, inside%s' % (
            self.component_position_expr(component_idx, 'rel'),
            self.component_parameter_string(component_idx))

    def component_normal_call_params(self, component_idx, subobject_offset):
        return '%s, subobject - %d, p_normal%s' % (
            self.component_position_expr(component_idx, 'p'), subobject_offset,
            self.component_parameter_string(component_idx))

    # ---- run-time parameters: component position + the component's own parameters
    def parameter_declarations(self):
        decls = []
        for i, c in enumerate(self.components):
            decls.append('float3 _pos_%d' % i)
            decls += ['%s_%d' % (d, i) for d in c.parameter_declarations()]
        return decls

    def parameter_values(self):
        values = []
        for c in self.components:
            values.append(c.position)
            values += c.parameter_values()
        return values


CULL_BALL_FACTOR = 1.25      # radius of the culling ball / radius of the ball that holds the body
CULL_BALL_SLACK = 1.05       # on the squared radius: rounding of the test itself
CULL_MIN_MARGIN = 0.05       # least distance of the centre from a bounding plane / radius of the body


def plane_body_cull_ball(components):
    """(centre, squared radius of the culling ball) in tracer coordinates for a body bounded by
    planes only (LayerComponent / HalfSpaceComponent), or None.  A ray that stays outside that ball
    on the stretch the tracer looks at is known to end the reference's component tests without a
    hit, whatever their roundings (clr_ball_may_be_hit)."""
    import itertools
    planes = []
    for c in components:
        if not isinstance(c, HalfSpaceComponent):
            return None
        n = numpy.ravel(numpy.asarray(c.normal, dtype=float))
        if abs(numpy.linalg.norm(n) - 1.0) > 1e-6:
            return None
        offset = float(numpy.dot(numpy.ravel(numpy.asarray(c.position, dtype=float)), n))
        planes.append((n, offset + c.h))
        if isinstance(c, LayerComponent):
            planes.append((-n, -offset))
    normals = numpy.array([n for n, _ in planes])
    heights = numpy.array([d for _, d in planes])
    scale = max(1.0, numpy.abs(heights).max())
    vertices = []
    for i, j, k in itertools.combinations(range(len(planes)), 3):
        m = normals[[i, j, k]]
        if abs(numpy.linalg.det(m)) < 1e-9:
            continue
        x = numpy.linalg.solve(m, heights[[i, j, k]])
        if (normals.dot(x) <= heights + 1e-9 * scale).all():
            vertices.append(x)
    if len(vertices) < 4:
        return None
    vertices = numpy.array(vertices)
    centre = 0.5 * (vertices.min(axis=0) + vertices.max(axis=0))
    radius = float(numpy.linalg.norm(vertices - centre, axis=1).max()) * (1.0 + 1e-6)
    margins = heights - normals.dot(centre)
    if radius <= 0 or margins.min() < CULL_MIN_MARGIN * radius:
        return None
    # The vertices say nothing about a body that is unbounded: it is bounded iff each coordinate
    # has a finite maximum and minimum over it (a direction of recession has a non-zero component).
    try:
        from scipy.optimize import linprog
    except ImportError:
        return None
    for axis in range(3):
        for sign in (1.0, -1.0):
            cost = numpy.zeros(3)
            cost[axis] = -sign
            result = linprog(cost, A_ub=normals, b_ub=heights, bounds=[(None, None)] * 3, method='highs')
            if result.status != 0 or abs(result.x[axis] - centre[axis]) > radius * (1.0 + 1e-6):
                return None
    return centre, (CULL_BALL_FACTOR * radius) ** 2 * CULL_BALL_SLACK


class UniqueConvexIntersection(ConvexIntersection):
    """All component parameters baked into the generated source as literals."""

    def __init__(self, origin, components):
        ConvexIntersection.__init__(self, origin, components)
        self.unique_tracer_id = next_unique_id()

    def parameter_declarations(self):
        return []

    def parameter_values(self):
        return []

    def component_position_expr(self, component_idx, base):
        pos = self.components[component_idx].position
        if numpy.linalg.norm(pos) < 1e-10:
            return base
        return '%s - %s' % (base, vec_literal(pos))

    def component_parameter_string(self, component_idx):
        parts = []
        for val in self.components[component_idx].parameter_values():
            parts.append(vec_literal(val) if numpy.size(val) > 1 else float_literal(val))
        return ''.join(', ' + p for p in parts)


class FixedConvexIntersection(UniqueConvexIntersection):
    """Baked literals shared by every instance of the subclass (pose via the affine only)."""

    def __init__(self, *args, **kwargs):
        UniqueConvexIntersection.__init__(self, *args, **kwargs)
        self.unique_tracer_id = ''

    @property
    def can_hit_expression(self):
        """Bodies bounded by planes: the ray against a ball around the body (literals of the class)."""
        if not hasattr(self, '_cull_ball'):
            self._cull_ball = plane_body_cull_ball(self.components)
        if self._cull_ball is None:
            return None
        centre, rho2 = self._cull_ball
        where = 'local_pos' if numpy.linalg.norm(centre) < 1e-12 else 'local_pos - %s' % vec_literal(centre)
        return '(within || clr_ball_may_be_hit(%s, local_ray, %s, isec_dist))' % (where, float_literal(rho2))


class PolyhedronLight:
    """Light interface (surface_area, random_surface_point_and_normal,
    center_and_min_sampling_distance) for convex bodies bounded by planes only: any pose, also
    sheared or stretched.  The reference has the interface for Sphere alone
    (objects/sphere.py:12-25); SURVEY section 8 row f4.  The boundary is found once per call from
    the half-spaces of the components: vertices = triple plane intersections that satisfy every
    half-space, faces = their convex polygons, cut into fans of triangles in world coordinates;
    a point is drawn uniformly by area (triangle by its share, then uniform in the triangle)."""

    def _half_spaces(self):
        """[(n, d)]: the body is {x : n.x <= d for all} in tracer coordinates."""
        planes = []
        for c in self.components:
            n = numpy.ravel(numpy.asarray(c.normal, dtype=float))
            offset = float(numpy.dot(numpy.ravel(numpy.asarray(c.position, dtype=float)), n))
            if isinstance(c, LayerComponent):
                planes += [(n, offset + c.h), (-n, -offset)]
            elif isinstance(c, HalfSpaceComponent):
                planes.append((n, offset + c.h))
            else:
                raise RuntimeError("light interface: %s is not bounded by planes" % type(c).__name__)
        return planes

    def _world_triangles(self):
        """[(a, b, c, outward unit normal, area)] in world coordinates."""
        import itertools
        planes = self._half_spaces()
        normals = numpy.array([n for n, _ in planes])
        heights = numpy.array([d for _, d in planes])
        scale = max(1.0, numpy.abs(heights).max())
        vertices = []
        for i, j, k in itertools.combinations(range(len(planes)), 3):
            m = normals[[i, j, k]]
            if abs(numpy.linalg.det(m)) < 1e-12:
                continue
            x = numpy.linalg.solve(m, heights[[i, j, k]])
            if (normals.dot(x) <= heights + 1e-9 * scale).all() and \
                    not any(numpy.linalg.norm(x - v) < 1e-9 * scale for v in vertices):
                vertices.append(x)
        if len(vertices) < 4:
            raise RuntimeError("light interface: the half-spaces do not bound a body")
        vertices = numpy.array(vertices)
        to_world = self.global_to_tracer_coordinate_transform()
        world = vertices.dot(numpy.asarray(to_world.linear, dtype=float).T) + \
            numpy.ravel(numpy.asarray(to_world.translation, dtype=float))
        centre = world.mean(axis=0)
        triangles = []
        for n, d in planes:
            on_face = numpy.flatnonzero(numpy.abs(vertices.dot(n) - d) < 1e-7 * scale)
            if len(on_face) < 3:
                continue
            corners = world[on_face]
            middle = corners.mean(axis=0)
            normal = None
            for a, b in itertools.combinations(range(len(corners)), 2):
                normal = numpy.cross(corners[a] - middle, corners[b] - middle)
                if numpy.linalg.norm(normal) > 1e-12:
                    break
            normal = normal / numpy.linalg.norm(normal)
            if numpy.dot(middle - centre, normal) < 0:
                normal = -normal
            u = corners[0] - middle
            u = u / numpy.linalg.norm(u)
            v = numpy.cross(normal, u)
            order = numpy.argsort(numpy.arctan2((corners - middle).dot(v), (corners - middle).dot(u)))
            ring = corners[order]
            for t in range(1, len(ring) - 1):
                area = 0.5 * numpy.linalg.norm(numpy.cross(ring[t] - ring[0], ring[t + 1] - ring[0]))
                triangles.append((ring[0], ring[t], ring[t + 1], normal, area))
        return triangles

    def surface_area(self):
        return float(sum(t[4] for t in self._world_triangles()))

    def random_surface_point_and_normal(self, rng=numpy.random):
        triangles = self._world_triangles()
        pick, a, b = rng.rand(3)
        cumulative = numpy.cumsum([t[4] for t in triangles])
        index = min(int(numpy.searchsorted(cumulative, pick * cumulative[-1], side='right')), len(triangles) - 1)
        p0, p1, p2, normal, _ = triangles[index]
        if a + b > 1.0:                      # fold the unit square onto the triangle
            a, b = 1.0 - a, 1.0 - b
        return p0 + a * (p1 - p0) + b * (p2 - p0), normal

    def center_and_min_sampling_distance(self):
        corners = numpy.array([p for t in self._world_triangles() for p in t[:3]])
        centre = 0.5 * (corners.min(axis=0) + corners.max(axis=0))
        return centre, 2.0 * float(numpy.linalg.norm(corners - centre, axis=1).max())


class RoundLight:
    """Light interface shared by Cylinder and Cone: discs and a lateral surface of revolution
    about the local z axis, posed by rotation and translation only (a stretched or sheared pose
    would make uniform sampling by area a different problem; refused like a scaled Sphere light)."""

    def _check_no_distortion(self):
        if not self.coordinates.is_orthogonal():
            raise RuntimeError("illegal operation for a stretched or sheared %s" % type(self).__name__)

    def _frame(self):
        """World position of the local origin and the unit vectors of the local x, y, z axes."""
        self._check_no_distortion()
        rotation = numpy.asarray(self.coordinates.linear, dtype=float).dot(
            numpy.asarray(rotation_aligning_vectors((0, 0, 1), self.axis), dtype=float))
        return numpy.ravel(numpy.asarray(self.coordinates.translation, dtype=float)), rotation[:, 0], rotation[:, 1], rotation[:, 2]

    def _draw(self, parts, rng):
        """parts: [(area, sampler(u, v) -> (point, normal))]; one part by area, then a point on it."""
        pick, u, v = rng.rand(3)
        cumulative = numpy.cumsum([p[0] for p in parts])
        index = min(int(numpy.searchsorted(cumulative, pick * cumulative[-1], side='right')), len(parts) - 1)
        return parts[index][1](u, v)

    def surface_area(self):
        return float(sum(p[0] for p in self._parts()))

    def random_surface_point_and_normal(self, rng=numpy.random):
        return self._draw(self._parts(), rng)


# ---------------------------------------------------------------- components

class HalfSpaceComponent(ConvexIntersection.Component):
    n_subobjects = 1

    def __init__(self, normal, h):
        ConvexIntersection.Component.__init__(self)
        self.normal = normalize_tuple(normal)
        self.h = h

    def parameter_declarations(self):
        return ['float3 normal', 'float h']


class LayerComponent(HalfSpaceComponent):
    """Infinite slab of thickness h along ``axis`` (h defaults to |axis|)."""
    n_subobjects = 2

    def __init__(self, axis, h=None):
        HalfSpaceComponent.__init__(self, axis, vec_norm(axis) if h is None else h)


class ZLayerComponent(LayerComponent):
    def __init__(self):
        LayerComponent.__init__(self, (0, 0, 1), 1.0)

    def parameter_declarations(self):
        return []


class SphereComponent(ConvexIntersection.Component):
    n_subobjects = 1

    def __init__(self, pos, R):
        ConvexIntersection.Component.__init__(self, pos)
        self.R = R

    def parameter_declarations(self):
        return ['float R']


class CylinderComponent(ConvexIntersection.Component):
    n_subobjects = 1

    def __init__(self, axis, R):
        ConvexIntersection.Component.__init__(self)
        self.axis = normalize_tuple(axis)
        self.R = R

    def parameter_declarations(self):
        return ['float3 axis', 'float R']


class ZCylinderComponent(CylinderComponent):
    def __init__(self):
        CylinderComponent.__init__(self, (0, 0, 1), 1.0)

    def parameter_declarations(self):
        return []


class ConeComponent(ConvexIntersection.Component):
    n_subobjects = 1

    def __init__(self, pos, axis, slope):
        ConvexIntersection.Component.__init__(self, pos)
        self.axis = normalize_tuple(axis)
        self.slope = slope

    def parameter_declarations(self):
        return ['float3 axis', 'float slope']


class ZConeComponent(ConeComponent):
    def __init__(self):
        ConeComponent.__init__(self, (0, 0, 0), (0, 0, 1), 1.0)

    def parameter_declarations(self):
        return []


# ---------------------------------------------------------------- composed shapes

class Cylinder(RoundLight, FixedConvexIntersection):
    """Unit z-slab ∩ unit z-cylinder, posed by rotation(axis) ∘ scale(R, R, height)."""

    def _parts(self):
        origin, ex, ey, ez = self._frame()
        R, h = float(self.R), float(self.height)

        def disc(z, outward):
            def sample(u, v):
                rho, phi = R * math.sqrt(u), 2.0 * math.pi * v
                return origin + z * ez + rho * (math.cos(phi) * ex + math.sin(phi) * ey), outward
            return sample

        def side(u, v):
            phi = 2.0 * math.pi * v
            radial = math.cos(phi) * ex + math.sin(phi) * ey
            return origin + u * h * ez + R * radial, radial
        return [(math.pi * R * R, disc(0.0, -ez)), (math.pi * R * R, disc(h, ez)), (2.0 * math.pi * R * h, side)]

    def center_and_min_sampling_distance(self):
        origin, ex, ey, ez = self._frame()
        return origin + 0.5 * self.height * ez, 2.0 * math.hypot(self.R, 0.5 * self.height)

    def __init__(self, bottom_center, axis, height, R):
        self.axis = axis
        self.height = height
        self.R = R
        FixedConvexIntersection.__init__(self, bottom_center,
                                         [ZLayerComponent(), ZCylinderComponent()])

    def tracer_coordinate_system(self):
        rotation = Affine(linear=rotation_aligning_vectors((0, 0, 1), self.axis))
        return rotation(Affine(scaling=(self.R, self.R, self.height)))


class Cone(RoundLight, FixedConvexIntersection):
    """{z <= 1} ∩ unit z-cone with its tip at the origin, posed like Cylinder."""

    def _parts(self):
        tip, ex, ey, ez = self._frame()
        R, h = float(self.R), float(self.height)
        slant = math.hypot(R, h)

        def base(u, v):
            rho, phi = R * math.sqrt(u), 2.0 * math.pi * v
            return tip + h * ez + rho * (math.cos(phi) * ex + math.sin(phi) * ey), ez

        def side(u, v):
            # area grows linearly with the distance from the tip: that distance is slant * sqrt(u)
            f, phi = math.sqrt(u), 2.0 * math.pi * v
            radial = math.cos(phi) * ex + math.sin(phi) * ey
            return tip + f * (h * ez + R * radial), (h * radial - R * ez) / slant
        return [(math.pi * R * R, base), (math.pi * R * slant, side)]

    def center_and_min_sampling_distance(self):
        tip, ex, ey, ez = self._frame()
        centre = tip + 0.5 * self.height * ez
        return centre, 2.0 * max(0.5 * self.height, math.hypot(self.R, 0.5 * self.height))

    def __init__(self, tip, axis, height, R):
        self.axis = axis
        self.height = height
        self.R = R
        FixedConvexIntersection.__init__(self, tip, [
            HalfSpaceComponent(normal=(0, 0, 1), h=1.0), ZConeComponent()])

    def tracer_coordinate_system(self):
        rotation = Affine(linear=rotation_aligning_vectors((0, 0, 1), self.axis))
        return rotation(Affine(scaling=(self.R, self.R, self.height)))


PHI = 0.5 * (1 + math.sqrt(5))  # golden ratio


class Parallelepiped(ConvexIntersection):
    """Three slabs 0 <= x.n_i <= |ax_i| with n_i = ax_i / |ax_i|, measured from `origin`
    (a box when the axes are orthogonal)."""

    def __init__(self, origin, ax1, ax2, ax3):
        ConvexIntersection.__init__(self, origin, [LayerComponent(ax) for ax in (ax1, ax2, ax3)])
        self.unique_tracer_id = ''

    # ---- light interface (the reference has it for Sphere only, objects/sphere.py:12-25;
    # SURVEY §8 row f4).  A box can then be a next-event light: `obj.bidirectional_light = True`.
    def _world_frame(self):
        """A corner and the three edge vectors, in world coordinates."""
        normals = numpy.array([numpy.ravel(c.normal) for c in self.components], dtype=float)
        heights = numpy.array([c.h for c in self.components], dtype=float)
        # column j of `edges`: the displacement that raises slab coordinate j from 0 to h_j only
        edges = numpy.linalg.inv(normals) * heights[None, :]
        linear = numpy.asarray(self.coordinates.linear, dtype=float)
        return numpy.ravel(self.coordinates.translation).astype(float), linear.dot(edges)

    def _faces(self):
        """[(corner, edge u, edge v, outward unit normal, area)] of the six faces."""
        corner, edges = self._world_frame()
        centre = corner + edges.sum(axis=1) / 2.0
        faces = []
        for i in range(3):
            j, k = (i + 1) % 3, (i + 2) % 3
            u, v = edges[:, j], edges[:, k]
            normal = numpy.cross(u, v)
            area = numpy.linalg.norm(normal)
            normal = normal / area
            for start in (corner, corner + edges[:, i]):
                outward = normal if numpy.dot(start + (u + v) / 2.0 - centre, normal) > 0 else -normal
                faces.append((start, u, v, outward, area))
        return faces

    def surface_area(self):
        return float(sum(f[4] for f in self._faces()))

    def random_surface_point_and_normal(self, rng=numpy.random):
        faces = self._faces()
        pick, a, b = rng.rand(3)
        cumulative = numpy.cumsum([f[4] for f in faces])
        start, u, v, outward, _ = faces[min(int(numpy.searchsorted(cumulative, pick * cumulative[-1], side='right')), 5)]
        return start + a * u + b * v, outward

    def center_and_min_sampling_distance(self):
        corner, edges = self._world_frame()
        signs = numpy.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=float)
        half_diagonal = max(numpy.linalg.norm(edges.dot(sg)) for sg in signs) / 2.0
        return corner + edges.sum(axis=1) / 2.0, 2.0 * half_diagonal


class HyperplaneRepresentation(PolyhedronLight, FixedConvexIntersection):
    """Polyhedron given by outward plane vectors p (plane at distance |p| along p)."""

    def __init__(self, center, planes, R):
        FixedConvexIntersection.__init__(
            self, center, [HalfSpaceComponent(p, vec_norm(p)) for p in planes])
        self.linear_transform(scaling=R)


class SymmetricDualPolyhedron(PolyhedronLight, FixedConvexIntersection):
    """Dual of the point-symmetric vertex set {±v}: one slab per listed vertex."""

    def __init__(self, center, vertices, R):
        slabs = []
        for v in vertices:
            slab = LayerComponent(tuple(-x for x in v), vec_norm(v) * 2.0)
            slab.position = v
            slabs.append(slab)
        FixedConvexIntersection.__init__(self, center, slabs)
        self.linear_transform(scaling=R)


class Tetrahedron(HyperplaneRepresentation):
    def __init__(self, center, R):
        q = 1.0 / math.sqrt(2)
        HyperplaneRepresentation.__init__(
            self, center, [(1, 0, -q), (-1, 0, -q), (0, 1, q), (0, -1, q)], R / (q + 1 / q))


class Cube(SymmetricDualPolyhedron):
    def __init__(self, center, side=None, R=None):
        if side is None:
            # as in the reference, calling with neither argument is a TypeError (R is None)
            side = R / math.sqrt(3) * 2.0
        SymmetricDualPolyhedron.__init__(
            self, center, [(1, 0, 0), (0, 1, 0), (0, 0, 1)], side / 2.0)


class Octahedron(SymmetricDualPolyhedron):
    def __init__(self, center, R):
        SymmetricDualPolyhedron.__init__(
            self, center, [(1, 1, 1), (1, -1, 1), (-1, 1, 1), (-1, -1, 1)], R / 3.0)


class Dodecahedron(SymmetricDualPolyhedron):
    def __init__(self, center, R):
        SymmetricDualPolyhedron.__init__(
            self, center,
            [(0, 1, PHI), (1, PHI, 0), (PHI, 0, 1), (0, -1, PHI), (-1, PHI, 0), (PHI, 0, -1)],
            R / (PHI + 1.0 / PHI))


class Icosahedron(SymmetricDualPolyhedron):
    def __init__(self, center, R):
        q = 1 / PHI
        SymmetricDualPolyhedron.__init__(
            self, center,
            [(1, 1, 1), (-1, 1, 1), (1, -1, 1), (-1, -1, 1), (0, q, PHI), (0, -q, PHI),
             (q, PHI, 0), (-q, PHI, 0), (PHI, 0, q), (PHI, 0, -q)],
            R / (PHI + 1 / PHI ** 3))
