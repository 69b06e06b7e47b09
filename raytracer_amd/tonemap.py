"""8-bit image straight from the device accumulator.

Device counterpart of ``imgutils.Image._to_24bit`` and ``flares`` (reference imgutils.py:60-80,
121-148) for low normalisation = minimum (or none) and brightness reference = mean, with or
without flares.  Other reduction functions raise NotImplementedError and the caller keeps the
host path.
The pipeline is evaluated in binary64 like numpy's; the scaling is applied as one factor
(brightness / reference) instead of a division followed by a multiplication, and the mean is
summed per block, so a value whose 8-bit quantisation boundary lies within ~1e-15 relative can
differ by one count from the host result (tests allow 1 count on < 1e-5 of the values).
The flare filters repeat scipy.ndimage's correlate1d term by term on binary64 planes (the
reference's arrays are binary32 there, stored between the filter passes: same allowance).
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
        binary, _ = program_cache.get_binary(source, ['-O3', '-ffp-contract=off'], arch)
        self.prog = ctypes.c_void_p()
        _check(acc.lib.clr_program_load(acc.ctx, binary, len(binary), ctypes.byref(self.prog)))
        n = renderer.n_pixels
        self.n_blocks = (n + 255) // 256
        self.order = acc.to_device(np.ascontiguousarray(renderer.image_order, dtype=np.int32))
        self.block_min = acc.new_array((self.n_blocks,), np.float64)
        self.block_sum = acc.new_array((self.n_blocks,), np.float64)
        self.out = acc.new_array((n * 3,), np.uint8)
        self.base = None
        self.planes = None       # flares: three (H, W, 3) binary64 work planes, made on first use

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
        if settings.brightness_reference is not np.mean or not (no_floor or lo_norm is np.min):
            raise NotImplementedError("device tone mapping covers min / mean normalisation only")
        n = r.n_pixels
        acc.call('clr_tm_reduce', n, (r.img, self.base), value_args=(np.int32(n), self.block_min, self.block_sum),
                 prog=self.prog)
        floor = 0.0 if no_floor else float(self.block_min.get().min())
        ref = float(self.block_sum.get().sum()) / n - floor
        height, width = r.img_shape
        if settings.flares:
            return self._with_flares(floor, settings.brightness / ref, 1.0 / settings.gamma)
        acc.call('clr_tm_encode', n, (r.img, self.base, self.order),
                 value_args=(np.int32(n), np.int32(width), np.float64(floor),
                             np.float64(settings.brightness / ref), np.float64(1.0 / settings.gamma), self.out),
                 prog=self.prog)
        return self.out.get().reshape(height, width, 3)

    # ---- flares (reference imgutils.py:121-148)

    @staticmethod
    def flare_filters():
        """(gaussian taps, tail taps as correlate1d applies them, reach of the tail).

        The gaussian is scipy.ndimage.gaussian_filter's for sigma 1 (radius int(4 sigma + 0.5));
        the tail is the reference's 198-tap kernel after convolve1d's reversal, and ``reach`` the
        number of taps before the output position (even length, origin 0: length // 2 - 1)."""
        radius = 4
        x = np.arange(-radius, radius + 1)
        gauss = np.exp(-0.5 / 1.0 * x ** 2)
        gauss = gauss / gauss.sum()
        tail = np.exp(-np.arange(0, 100, dtype=np.float32) * 0.2)
        kernel = np.concatenate((tail[-1:1:-1], tail))
        kernel /= kernel.sum()
        taps = np.ascontiguousarray(kernel[::-1], dtype=np.float64)
        return gauss, taps, len(taps) // 2 - 1

    def _with_flares(self, floor, scale, inv_gamma):
        r, acc = self.renderer, self.acc
        height, width = r.img_shape
        n = r.n_pixels
        elements = np.int64(n * 3)
        if self.planes is None:
            self.planes = [acc.new_array((n * 3,), np.float64) for _ in range(3)]
            gauss, tail, self.tail_reach = self.flare_filters()
            self.gauss, self.tail, self.tail_taps = acc.to_device(gauss), acc.to_device(tail), len(tail)
        plane, tmp_a, tmp_b = self.planes
        row, pixel = np.int64(width * 3), np.int64(3)
        acc.call('clr_tm_linear', n, (r.img, self.base, self.order),
                 value_args=(np.int32(n), np.int32(width), np.float64(floor), np.float64(scale),
                             np.float64(inv_gamma), plane), prog=self.prog)
        # gaussian of what exceeds 1, along the rows axis and then along the columns axis
        acc.call('clr_tm_correlate', int(elements), (plane, tmp_a),
                 value_args=(elements, row, np.int32(height), self.gauss, np.int32(9), np.int32(4), np.int32(1),
                             np.float64(1.0)), prog=self.prog)
        acc.call('clr_tm_correlate', int(elements), (tmp_a, tmp_b),
                 value_args=(elements, pixel, np.int32(width), self.gauss, np.int32(9), np.int32(4), np.int32(1),
                             np.float64(-1.0)), prog=self.prog)
        acc.call('clr_tm_merge', int(elements), (plane, tmp_b), value_args=(elements, np.float64(1.0)),
                 prog=self.prog)
        # exponential tail of what exceeds 2, along the rows axis
        acc.call('clr_tm_correlate', int(elements), (plane, tmp_a),
                 value_args=(elements, row, np.int32(height), self.tail, np.int32(self.tail_taps),
                             np.int32(self.tail_reach), np.int32(0), np.float64(2.0)), prog=self.prog)
        acc.call('clr_tm_encode_plane', int(elements), (plane, tmp_a),
                 value_args=(elements, np.float64(2.0), self.out), prog=self.prog)
        return self.out.get().reshape(height, width, 3)
