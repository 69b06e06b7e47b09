"""Large triangle mesh in the box room (BASELINE config C4): a displaced torus-knot tube,
1000 x 500 quads = 1 000 000 triangles by default, inside an Octree built by the native
builder in the flattened layout.  The reference has no runnable counterpart at this size
(SURVEY F9); the scene follows SURVEY §8(d)."""
import numpy as np
from raytracer_amd.scenes.default_scenes import BoxScene
from raytracer_amd.objects import TriangleMesh, Octree


def torus_knot_mesh(n_u=1000, n_v=500, tube=0.45, bump=0.05):
    """(vertices (n_u*n_v, 3) float32, faces (2*n_u*n_v, 3) int32) of a (2,3) torus knot tube."""
    u = np.linspace(0.0, 2 * np.pi, n_u, endpoint=False)
    v = np.linspace(0.0, 2 * np.pi, n_v, endpoint=False)

    def curve(t):
        r = 2.0 + np.cos(3 * t)
        return np.stack((r * np.cos(2 * t), r * np.sin(2 * t), np.sin(3 * t)), axis=-1)
    c = curve(u)
    tangent = curve(u + 1e-4) - curve(u - 1e-4)
    tangent /= np.linalg.norm(tangent, axis=1, keepdims=True)
    side = np.cross(tangent, np.array([0.0, 0.0, 1.0]))
    side /= np.linalg.norm(side, axis=1, keepdims=True)
    up = np.cross(side, tangent)
    U, V = np.meshgrid(u, v, indexing='ij')
    radius = tube + bump * np.sin(13 * U) * np.sin(7 * V)
    pts = (c[:, None, :] + radius[..., None] * (np.cos(V)[..., None] * side[:, None, :]
                                                 + np.sin(V)[..., None] * up[:, None, :]))
    vertices = pts.reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(n_u), np.arange(n_v), indexing='ij')
    a = (i * n_v + j).ravel()
    b = (((i + 1) % n_u) * n_v + j).ravel()
    cc = (((i + 1) % n_u) * n_v + (j + 1) % n_v).ravel()
    d = (i * n_v + (j + 1) % n_v).ravel()
    faces = np.concatenate((np.stack((a, b, cc), 1), np.stack((a, cc, d), 1))).astype(np.int32)
    return vertices, faces


def make_scene(n_u=1000, n_v=500, layout='flat', max_depth=None, max_faces_per_leaf=8,
               max_steps=2000, material='white'):
    s = BoxScene()
    s.get_object('light').bidirectional_light = True
    vertices, faces = torus_knot_mesh(n_u, n_v)
    if max_depth is None:       # about one level per factor of 8 in the face count
        max_depth = int(min(10, max(3, np.ceil(np.log2(max(len(faces), 8)) / 3.0) + 2)))
    mesh = TriangleMesh(vertices, faces, auto_scale=True, center=(0, 0, 1), scale=1.0,
                        auto_flip_normal=True)
    s.add_object(Octree(mesh, max_depth=max_depth, max_faces_per_leaf=max_faces_per_leaf,
                        builder='native', layout=layout, max_steps=max_steps), material, 'knot')
    s.image.size = (1920, 1080)
    s.samples_per_pixel = 16
    s.max_bounces = 4
    s.min_bounces = 2
    s.camera_position = (1.5, -4.2, 2.2)
    s.direct_camera_towards((0, 0, 0.9))
    return s


# building the full mesh takes a few seconds: only when loaded as a scene file
scene = make_scene() if __name__ in ('__main__', '<run_path>') else None
