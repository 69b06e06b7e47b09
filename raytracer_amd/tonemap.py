"""8-bit image straight from the device accumulator.

Device counterpart of ``imgutils.Image._to_24bit`` (reference imgutils.py:60-80) for the
settings the shipped scenes use: low normalisation = minimum (or none), brightness reference =
mean, no flares.  Anything else raises NotImplementedError and the caller keeps the host path.
The pipeline is evaluated in binary64 like numpy's; the scaling is applied as one factor
(brightness / reference) instead of a division followed by a multiplication, and the mean is
summed per block, so a value whose 8-bit quantisation boundary lies within ~1e-15 relative can
differ by one count from the host result (tests allow 1 count on < 1e-5 of the values).
"""
import ctypes
import os

import numpy as np

from . import program_cache
from .accelerator import _check, DeviceArray
from .hip_compiler import TEMPLATE_DIR


class DeviceToneMapper:

    def __init__(self, renderer):
        self.renderer = renderer
        acc = self.acc = renderer.acc
        source = open(os.path.join(TEMPLATE_DIR, 'tonemap.hip')).read()
        arch = acc.device_info()['arch'].split(':')[0]
        binary, _ = program_cache.get_binary(source, ['-O3'], arch)
        self.prog = ctypes.c_void_p()
        _check(acc.lib.clr_program_load(acc.ctx, binary, len(binary), ctypes.byref(self.prog)))
        n = renderer.n_pixels
        self.n_blocks = (n + 255) // 256
        self.order = acc.to_device(np.ascontiguousarray(renderer.image_order, dtype=np.int32))
        self.block_min = acc.new_array((self.n_blocks,), np.float64)
        self.block_sum = acc.new_array((self.n_blocks,), np.float64)
        self.out = acc.new_array((n * 3,), np.uint8)
        self.base = None

    def set_base(self, data):
        """Earlier sums (H, W, 3) to add before encoding (the --append mode of clray.py)."""
        if data is None:
            self.base = None
            return
        r = self.renderer
        per_ray = np.asarray(data, dtype=np.float64)[r.image_order[:, 0], r.image_order[:, 1]]
        self.base = self.acc.to_device(np.ascontiguousarray(per_ray))

    def to_24bit(self, settings):
        """(H, W, 3) uint8 of the renderer's current accumulator."""
        r, acc = self.renderer, self.acc
        lo_norm = settings.low_normalization
        no_floor = lo_norm is None or (isinstance(lo_norm, int) and lo_norm == 0)
        if settings.flares or settings.brightness_reference is not np.mean or not (no_floor or lo_norm is np.min):
            raise NotImplementedError("device tone mapping covers min / mean normalisation without flares")
        n = r.n_pixels
        acc.call('clr_tm_reduce', n, (r.img, self.base), value_args=(np.int32(n), self.block_min, self.block_sum),
                 prog=self.prog)
        floor = 0.0 if no_floor else float(self.block_min.get().min())
        ref = float(self.block_sum.get().sum()) / n - floor
        height, width = r.img_shape
        acc.call('clr_tm_encode', n, (r.img, self.base, self.order),
                 value_args=(np.int32(n), np.int32(width), np.float64(floor),
                             np.float64(settings.brightness / ref), np.float64(1.0 / settings.gamma), self.out),
                 prog=self.prog)
        return self.out.get().reshape(height, width, 3)
