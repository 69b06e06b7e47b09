"""Wavelength grid, CIE-1931 colour matching and a few model spectra.

Follows /root/reference/spectrum.py:12-87.  The reference loads the CIE 1931 2-degree
table from ``data/cie-1931.csv``, a file it does not ship; this build uses that
file when present and otherwise the analytic multi-lobe fit of Wyman, Sloan & Shirley (2013)
to the same observer, so spectral parity with the reference is unpinned.
"""
import os
import numpy as np


def interp1d(x, data_x, data_y):
    data_y = np.asarray(data_y)
    if data_y.ndim == 1:
        data_y = data_y[np.newaxis, :]
    return np.array([np.interp(x, data_x, row) for row in data_y])


def cie_1931_xyz_fit(wavelength_nm):
    """Analytic approximation of the CIE 1931 colour matching functions (3 x N)."""
    w = np.asarray(wavelength_nm, dtype=float)

    def lobe(mu, s_lo, s_hi):
        sigma = np.where(w < mu, s_lo, s_hi)
        return np.exp(-0.5 * ((w - mu) / sigma) ** 2)

    x = 1.056 * lobe(599.8, 37.9, 31.0) + 0.362 * lobe(442.0, 16.0, 26.7) - 0.065 * lobe(501.1, 20.4, 26.2)
    y = 0.821 * lobe(568.8, 46.9, 40.5) + 0.286 * lobe(530.9, 16.3, 31.1)
    z = 1.217 * lobe(437.0, 11.8, 36.0) + 0.681 * lobe(459.0, 26.0, 13.8)
    return np.vstack((x, y, z))


class Spectrum:

    def __init__(self, wavelength_range_nm=None, resolution=None):
        if wavelength_range_nm is None:
            wavelength_range_nm = (360, 830)
        if resolution is None:
            resolution = wavelength_range_nm[1] - wavelength_range_nm[0] + 1
        self.resolution = resolution
        self.wavelength_range = wavelength_range_nm
        self.wavelengths = np.linspace(*wavelength_range_nm, num=resolution)

    def map_left(self, wavelengths, y):
        """Resample y given on ``wavelengths`` onto this grid."""
        return interp1d(self.wavelengths, wavelengths, y)

    def map_right(self, wavelengths, y):
        """Evaluate y given on this grid at ``wavelengths``."""
        return interp1d(wavelengths, self.wavelengths, y)

    def cie_1931_xyz(self):
        if not hasattr(self, 'cie_xyz'):
            table = os.path.join('data', 'cie-1931.csv')
            if os.path.exists(table):
                cie_data = np.genfromtxt(table, delimiter=',')
                self.cie_xyz = self.map_left(cie_data[:, 0], cie_data[:, 1:].T)
            else:
                self.cie_xyz = cie_1931_xyz_fit(self.wavelengths)
        return self.cie_xyz

    def cie_1931_rgb(self):
        xyz_from_rgb = 1.0 / 0.17697 * np.array([[0.49, 0.31, 0.20],
                                                 [0.17697, 0.81240, 0.01063],
                                                 [0.00, 0.01, 0.99]])
        return np.linalg.solve(xyz_from_rgb, self.cie_1931_xyz())

    def visible_intensity(self):
        intensity = np.sqrt(np.sum(self.cie_1931_rgb() ** 2, axis=0))
        return intensity / np.sum(intensity)

    def black_body(self, T, normalized=True):
        """Planck's law on the wavelength grid, T in kelvin."""
        from scipy.constants import h, c, k
        lam = self.wavelengths * 1e-9
        density = 2 * h * c ** 2 / lam ** 5 / (np.exp(h * c / (lam * k * T)) - 1.0)
        return density / np.max(density) if normalized else density

    def linear_dispersion_ior(self, n_d, abbe_number):
        slope = (n_d - 1.0) / ((486.1 - 656.3) * abbe_number)
        return (self.wavelengths - 589.3) * slope + n_d

    def gaussian(self, mean, stdev):
        return np.exp(-0.5 * ((self.wavelengths - mean) / stdev) ** 2)

    def get_color(self, density, brightness=None):
        c = np.dot(self.cie_1931_rgb(), np.ravel(density).T)
        if brightness is not None:
            c = c / np.max(c) * brightness
            if isinstance(brightness, int):
                c = c.astype(int)
        return tuple(c)
