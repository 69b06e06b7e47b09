"""Host-side geometry (float64 numpy): small vector helpers, affine maps used to place objects,
and the camera model (orientation matrix, per-pixel base rays, sub-pixel tilt, depth-of-field and
quasi-random direction samples).

One module for what the reference spreads over utils.py:3-15, transformations.py:5-129 and
camera.py:4-108; the public names are the same, and `install_flat_aliases()` maps the
reference's module names `utils`, `transformations` and `camera` onto it.  Functions that draw
random numbers take the ``numpy.random.RandomState`` to draw from, so the order of draws
 is explicit and replayable.
"""
import numpy as np

# ---------------------------------------------------------------- vector helpers
def normalize(vecs):
    """Scale the vectors along the last axis to unit length (zero vectors stay zero)."""
    vecs = np.asarray(vecs, dtype=float)
    norms = np.sqrt(np.sum(vecs * vecs, axis=-1, keepdims=True))
    inv = np.zeros_like(norms)
    np.divide(1.0, norms, out=inv, where=norms > 0)
    return vecs * inv


def normalize_tuple(vec):
    return tuple(normalize(np.array(tuple(vec), dtype=float)))


def vec_norm(vec):
    return float(np.sqrt(sum(float(c) ** 2 for c in vec)))


# ---------------------------------------------------------------- affine maps
EPSILON = 1e-8


class Affine:
    """x -> linear @ x + translation.

    Constructor keywords follow the reference: an optional axis/angle rotation is
    applied to ``linear`` first, then ``scaling`` (a scalar or a per-row factor).
    """

    def __init__(self, linear=None, translation=None, rotation_axis=None,
                 rotation_deg=None, scaling=None):
        mat = np.identity(3) if linear is None else np.asarray(linear, dtype=float)
        shift = np.zeros(3) if translation is None else np.ravel(translation).astype(float)
        if rotation_axis is not None:
            mat = rotation_matrix(rotation_axis, rotation_deg) @ mat
        if scaling is not None:
            factors = np.ravel(scaling).astype(float)
            if factors.size == 3:
                mat = factors.reshape(3, 1) * mat
            else:
                mat = factors * mat
        self._linear = mat
        self._translation = shift

    def __call__(self, other):
        if isinstance(other, Affine):
            return Affine(self._linear @ other._linear,
                          self._linear @ other._translation + self._translation)
        return self._linear @ np.ravel(other) + self._translation

    __mul__ = __call__

    def inverse(self):
        inv = np.linalg.inv(self._linear)
        return Affine(inv, -(inv @ self._translation))

    def transpose(self):
        return Affine(self._linear.T, -self._translation)

    def is_identity(self, epsilon=EPSILON):
        dev = np.linalg.norm(self._linear - np.identity(3), 'fro')
        return dev + np.linalg.norm(self._translation) < epsilon

    def is_orthogonal(self, epsilon=EPSILON):
        return self.transpose()(self).is_identity()

    @property
    def linear(self):
        return self._linear.copy()

    @property
    def translation(self):
        return self._translation.copy()

    @staticmethod
    def identity():
        return Affine()

    def __str__(self):
        return "AffineTransformation(\n%s * x +\n%s)" % (self._linear, self._translation)


def normalize_vector(vec):
    vec = np.ravel(vec).astype(float)
    length = np.linalg.norm(vec)
    if length < EPSILON:
        raise RuntimeError('normalizing near-zero vector')
    return vec * (1.0 / length)


_AXES = {'x': (1, 0, 0), 'y': (0, 1, 0), 'z': (0, 0, 1)}


def axis_symbol_to_vector(letter):
    try:
        return _AXES[letter.lower()]
    except KeyError:
        raise RuntimeError("invalid axis letter %s" % letter)


def rotation_matrix(axis, angle_deg=None, angle_rad=None):
    """Right-handed rotation about ``axis`` (Rodrigues' formula, column by column)."""
    if angle_rad is None:
        angle_rad = np.pi / 180.0 * angle_deg
    elif angle_deg is not None:
        raise RuntimeError('cannot have both angle_rad and angle_deg')
    if isinstance(axis, str):
        axis = axis_symbol_to_vector(axis)
    axis = normalize_vector(axis)
    c, s = np.cos(angle_rad), np.sin(angle_rad)
    columns = []
    for basis in np.identity(3):
        columns.append(c * basis + s * np.cross(axis, basis) + axis * np.dot(axis, basis) * (1.0 - c))
    return np.vstack(columns).T


def rotation_aligning_vectors(vec_to_rotate, vec_to_align_with):
    a = normalize_vector(vec_to_rotate)
    b = normalize_vector(vec_to_align_with)
    if np.linalg.norm(a - b) < EPSILON:
        return np.eye(3)
    if np.linalg.norm(a + b) < EPSILON:
        return -np.eye(3)  # point mirror, as in the reference
    return rotation_matrix(np.cross(a, b), angle_rad=np.arccos(np.dot(a, b)))


# ---------------------------------------------------------------- camera
def camera_rotmat(direction, up=(0, 0, 1)):
    """Columns = (right, -up', forward), each unit length."""
    direction = np.array(direction, dtype=float)
    up = np.array(up, dtype=float)
    right = np.cross(direction, up)
    true_up = np.cross(right, direction)
    return normalize(np.vstack((right, -true_up, direction))).transpose()


def rotmat_tilt_camera(xa, ya):
    """Small rotation by xa about the camera's y axis followed by ya about its x axis."""
    cx, sx = np.cos(xa), np.sin(xa)
    cy, sy = np.cos(ya), np.sin(ya)
    about_x = np.array([[1.0, 0.0, 0.0], [0.0, cy, -sy], [0.0, sy, cy]])
    about_y = np.array([[cx, 0.0, -sx], [0.0, 1.0, 0.0], [sx, 0.0, cx]])
    return about_x @ about_y


def camera_rays(wh, flat=False, wfov=60, direction=(0, 1, 0), up=(0, 0, 1)):
    """(h, w, 3) float64 array of un-jittered primary ray directions."""
    w, h = wh
    wfov = wfov / 180.0 * np.pi
    aspect = h / float(w)
    hfov = wfov * aspect
    rotmat = camera_rotmat(direction, up)
    if flat:
        ra = np.tan(wfov * 0.5)
        X, Y = np.meshgrid(np.linspace(-ra, ra, w), np.linspace(-ra * aspect, ra * aspect, h))
        Z = np.ones(X.shape)
    else:
        pixel_angle = float(wfov) / w
        xa = (np.arange(0, w) + 0.5) * pixel_angle - wfov / 2.0
        ya = (np.arange(0, h) + 0.5) * pixel_angle - hfov / 2.0
        Xa, Ya = np.meshgrid(xa, ya)
        X = np.sin(Xa) * np.cos(Ya)
        Z = np.cos(Xa) * np.cos(Ya)
        Y = np.sin(Ya)
    vecs = np.dstack((X, Y, Z)).reshape(w * h, 3).transpose()
    vecs = np.dot(rotmat, vecs).transpose().reshape(h, w, 3)
    if flat:
        vecs = normalize(vecs)
    return vecs


def quasi_random_direction_sample(n, hemisphere=True):
    """Fermat-spiral points on the sphere; with ``hemisphere`` the first n of 2n."""
    total = 2 * n if hemisphere else n
    golden_angle = np.pi * (3 - np.sqrt(5))
    theta = golden_angle * np.arange(total)
    z = np.linspace(1 - 1.0 / total, 1.0 / total - 1, total)
    radius = np.sqrt(1 - z * z)
    points = np.column_stack((radius * np.cos(theta), radius * np.sin(theta), z))
    return points[:n, :] if hemisphere else points


def random_dof_sample(rng=np.random):
    """Uniform point in the unit disc by rejection (two draws per attempt)."""
    while True:
        dofx = (rng.rand() - 0.5) * 2.0
        dofy = (rng.rand() - 0.5) * 2.0
        if dofx ** 2 + dofy ** 2 < 1.0:
            return dofx, dofy
