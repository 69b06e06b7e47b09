"""Camera geometry: orientation matrix, per-pixel base rays, DOF and quasi-random samples.

Host-side counterpart of /root/reference/camera.py:4-108.  All functions that draw
random numbers take the ``numpy.random.RandomState`` to draw from so that the order of
draws is explicit and replayable.
"""
import numpy as np
from .utils import normalize


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
