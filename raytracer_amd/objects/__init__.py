"""Shape classes (same names and constructors as /root/reference/objects/*.py)."""
from .convex_intersection import (ConvexIntersection, UniqueConvexIntersection,
                                  FixedConvexIntersection)
from .planes import HalfSpace, HalfSpaceComponent, LayerComponent, ZLayerComponent
from .sphere import Sphere, SphereComponent
from .cone import Cone, ConeComponent, ZConeComponent
from .cylinder import Cylinder, CylinderComponent, ZCylinderComponent
from .polyhedra import (Parallelepiped, HyperplaneRepresentation, SymmetricDualPolyhedron,
                        Tetrahedron, Cube, Octahedron, Dodecahedron, Icosahedron)
from .implicit_surface import ImplicitSurface
from .julia_sets import QuaternionJuliaSet, QuaternionJuliaSet2
from .triangle_mesh import TriangleMesh
from .octree import Octree
from .distance_field import DistanceField
