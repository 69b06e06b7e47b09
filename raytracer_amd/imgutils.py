"""Tone mapping and image output (host post-process; reference imgutils.py:8-148).

The preview window (SDL2 / pygame) of the reference is out of scope;
``Image.show`` is a no-op unless such a back end is installed.
"""
import numpy as np


class ColorEncodingSettings:
    """Pipeline: subtract ``low_normalization`` of the lightness, scale so that
    ``brightness_reference`` of the lightness maps to ``brightness``, gamma, optional flares,
    clamp to [0, 1], 8 bit."""

    def __init__(self):
        self.low_normalization = np.min
        self.brightness_reference = np.mean
        self.brightness = 0.4
        self.gamma = 1.8
        self.flares = False


class Image:

    def __init__(self, npy_filename=None, data=None, settings=None):
        self.data = data if npy_filename is None else np.load(npy_filename)
        self.settings = settings if settings is not None else ColorEncodingSettings()

    def _sum(self, imgdata):
        if self.data is None:
            return imgdata
        return self.data * 1 if imgdata is None else imgdata + self.data

    def save_raw(self, filename, imgdata=None):
        # np.save appends '.npy' only when missing, so 'out.raw.npy' stays as is
        np.save(filename, self._sum(imgdata))

    def save_png(self, filename, imgdata=None):
        from PIL import Image as PilImage
        PilImage.fromarray(self._to_24bit(imgdata), 'RGB').save(filename)

    def _to_24bit(self, imgdata=None):
        settings = self.settings
        imgdata = np.nan_to_num(np.array(self._sum(imgdata), dtype=float))
        lightness = np.ravel(np.mean(imgdata, 2))
        lo_norm = settings.low_normalization
        if lo_norm is not None and not (isinstance(lo_norm, int) and lo_norm == 0):
            floor = lo_norm(lightness)
            imgdata = imgdata - floor
            lightness = lightness - floor
        ref = settings.brightness_reference(lightness)
        imgdata = np.clip(imgdata / ref * settings.brightness, 0, None)
        imgdata = np.power(imgdata, 1.0 / settings.gamma)
        if settings.flares:
            imgdata = flares(imgdata)
        return (np.clip(imgdata, 0, 1.0) * 255).astype(np.uint8)

    def show(self, imgdata=None):
        pass


def flares(imgdata):
    """Bleed over-exposed values into their neighbourhood (gaussian, then a long vertical tail)."""
    from scipy import ndimage
    visible = np.clip(imgdata, 0, 1)
    excess = imgdata - visible
    for c in range(3):
        excess[:, :, c] = ndimage.gaussian_filter(excess[:, :, c], 1.0)
    imgdata = visible + excess
    visible = np.clip(imgdata, 0, 2)
    excess = imgdata - visible
    tail = np.exp(-np.arange(0, 100, dtype=np.float32) * 0.2)
    kernel = np.concatenate((tail[-1:1:-1], tail))
    kernel /= kernel.sum()
    excess = ndimage.convolve1d(excess, kernel, 0, None, 'constant', 0, 0)
    return visible + excess
