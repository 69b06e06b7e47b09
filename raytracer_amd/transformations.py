"""Affine maps used to place objects (host-side set-up math, float64 numpy).

API-compatible with the reference's ``transformations.py``
(/root/reference/transformations.py:5-129): ``Affine`` composes by call or ``*``,
``inverse()``, ``is_identity()``, ``is_orthogonal()``; ``rotation_matrix`` and
``rotation_aligning_vectors`` build rotations.
"""
import numpy as np

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
