"""Sphere-traced signed distance fields given as a code string (reference objects/distance_field.py:3-17).

``tracer_code`` is a C statement that assigns ``dist`` from the local coordinates
``x, y, z`` (e.g. ``"dist = sqrt(x*x + y*y + z*z) - 0.5"``); ``normal_code`` may assign
``normal``; otherwise central differences are used.
"""
from ..tracer import Tracer, next_unique_id


class DistanceField(Tracer):

    def __init__(self, tracer_code, normal_code=None, center=(0, 0, 0), bndR=None,
                 max_itr=1000, precision=1e-5, self_intersection=True):
        Tracer.__init__(self, position=center)
        self.max_itr = max_itr
        self.precision = precision
        self.bndR = bndR
        self.no_self_intersection = not self_intersection
        self.tracer_code = tracer_code.strip()
        self.normal_code = normal_code
        self.unique_tracer_id = next_unique_id()
