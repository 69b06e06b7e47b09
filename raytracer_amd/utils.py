"""Small vector helpers shared by the scene-description layer.

Mirrors the helper surface of the reference's ``utils.py`` (normalize,
normalize_tuple, vec_norm; /root/reference/utils.py:3-15).
"""
import numpy as np


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
