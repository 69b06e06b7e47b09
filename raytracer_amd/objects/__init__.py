"""Shape classes (same names and constructors as the reference's objects package)."""
from .primitives import Sphere, HalfSpace
from .convex import (ConvexIntersection, UniqueConvexIntersection, FixedConvexIntersection,
                     HalfSpaceComponent, LayerComponent, ZLayerComponent, SphereComponent,
                     CylinderComponent, ZCylinderComponent, ConeComponent, ZConeComponent,
                     Cylinder, Cone, Parallelepiped, HyperplaneRepresentation,
                     SymmetricDualPolyhedron, Tetrahedron, Cube, Octahedron, Dodecahedron,
                     Icosahedron)
from .marching import ImplicitSurface, QuaternionJuliaSet, QuaternionJuliaSet2, DistanceField
from .meshes import TriangleMesh, Octree
