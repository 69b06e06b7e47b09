"""Materials and colour: the wavelength grid with CIE-1931 colour matching (Spectrum), the two
colour representations behind one interface (RgbShader, SpectrumShader) and the default
material tables.

One module for the reference's spectrum.py:12-87, shader.py:4-194 and the material part of
scenes/default_scenes.py:9-78; `install_flat_aliases()` maps the module names `spectrum` and
`shader` onto it.  Material tables on the device are property-major: entry
``prop_offset + object_id`` with id 0 = the ambient ('default') material and
``prop_offset = k * (n_objects + 1)`` for the k-th property of a set.

The reference loads the CIE 1931 2-degree table from ``data/cie-1931.csv``, a file it does not
ship; that file is used when present, otherwise the analytic multi-lobe fit of
Wyman, Sloan & Shirley (2013) to the same observer.  Everything downstream of the table is
pinned against the reference's own code run on this table (tests/golden/spectral-*.npz,
DESIGN.md section 4); only the table values themselves are this package's.
"""
import os
import numpy as np

# ---------------------------------------------------------------- spectra
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


# ---------------------------------------------------------------- shaders
class Shader:

    def __init__(self, scene):
        self.scene = scene

    def each_object_material(self, property_list):
        """Yield (object id, property index, value); id 0 is the default material."""
        materials = self.scene.materials
        for p_idx, pname in enumerate(property_list):
            default = np.array(materials['default'][pname])
            yield 0, p_idx, default
            for i, obj in enumerate(self.scene.objects):
                yield i + 1, p_idx, materials[obj.material].get(pname, default)

    def get_material_property_offsets(self):
        stride = len(self.scene.objects) + 1
        offsets = []
        for property_list in self.material_property_sets:
            offsets += [(p.upper(), k * stride) for k, p in enumerate(property_list)]
        return offsets

    def init_sample(self, renderer):
        pass


class RgbShader(Shader):
    """Colours are RGB triples (float4 rows on the device); no dispersion."""

    def __init__(self, scene):
        Shader.__init__(self, scene)
        self.rgb = True
        self.color_cl_type = 'float3'
        self.color_hip_type = 'vec3'
        self.material_property_sets = [
            ['diffuse', 'emission', 'reflection', 'transparency', 'volume_absorption'],
            ['ior', 'volume_scattering', 'volume_scattering_blur', 'reflection_blur',
             'transparency_blur'],
        ]

    def host_material_tables(self):
        """(colors (5*(N+1), 4) float32, scalars (5*(N+1),) float32)."""
        stride = len(self.scene.objects) + 1
        colors = np.zeros((stride * len(self.material_property_sets[0]), 4))
        scalars = np.zeros((stride * len(self.material_property_sets[1]), 1))
        for table, props, width in ((colors, self.material_property_sets[0], 3),
                                    (scalars, self.material_property_sets[1], 1)):
            for obj_id, p_idx, value in self.each_object_material(props):
                value = np.ravel(np.array(value, dtype=float))
                if value.size == 1 and width == 3:
                    value = np.repeat(value, 3)
                table[p_idx * stride + obj_id, :value.size] = value
        return colors.astype(np.float32), scalars[:, 0].astype(np.float32)

    def initialize_material_buffers(self, acc):
        colors, scalars = self.host_material_tables()
        self.material_buffers = [acc.new_const_buffer(colors), acc.new_const_buffer(scalars)]


class SpectrumShader(Shader):
    """One wavelength per sample pass; every colour is a scalar at that wavelength."""

    def __init__(self, scene):
        Shader.__init__(self, scene)
        self.rgb = False
        self.color_cl_type = 'float'
        self.color_hip_type = 'float'
        self.material_property_sets = [[
            'diffuse', 'emission', 'reflection', 'transparency', 'volume_scattering',
            'volume_absorption', 'reflection_blur', 'transparency_blur',
            'volume_scattering_blur', 'ior']]

    def prepare_host_tables(self):
        spectrum = self.scene.spectrum
        self.color_responses = spectrum.cie_1931_rgb()
        self.color_intensity_pdf = spectrum.visible_intensity()
        self.color_intensity_cdf = np.cumsum(self.color_intensity_pdf)
        rows = []
        for _, __, value in self.each_object_material(self.material_property_sets[0]):
            y = np.ravel(np.array(value, dtype=float))
            x = np.linspace(*spectrum.wavelength_range, num=y.size)
            rows.append(spectrum.map_left(x, y))
        self.host_mat_y = np.vstack(rows).astype(np.float32)

    def initialize_material_buffers(self, acc):
        self.prepare_host_tables()
        self.device_material_buffer = acc.new_const_buffer(self.host_mat_y[:, 0])
        self.material_buffers = [self.device_material_buffer]

    def sample_wavelength(self, omega):
        """Importance-sample a wavelength; returns (1/pdf, material row, colour mask)."""
        spectrum = self.scene.spectrum
        wavelength = np.interp([omega], self.color_intensity_cdf, spectrum.wavelengths)
        prob = spectrum.map_right([wavelength], self.color_intensity_pdf)
        at = lambda y: spectrum.map_right([wavelength], y).astype(np.float32)
        return (np.float32(1.0 / np.ravel(prob)[0]), np.ravel(at(self.host_mat_y)),
                np.ravel(at(self.color_responses)))


# ---------------------------------------------------------------- default materials
def default_rgb_materials():
    ambient = {          # "air": also supplies every property a material leaves out
        'diffuse': (1, 0, 1), 'emission': 0.0, 'reflection': 0.0, 'transparency': 0.0,
        'ior': 1.0, 'volume_absorption': 0.0, 'volume_scattering': 0.0,
        'volume_scattering_blur': 1.0, 'transparency_blur': 0.0, 'reflection_blur': 0.0,
    }
    wax = {
        'transparency': 0.9, 'transparency_blur': 0.7,
        'volume_scattering': 0.5, 'volume_scattering_blur': 1.0,
        'volume_absorption': (1.0 - np.array([1, .8, .3])) * 4.0,
        'reflection': 0.05, 'reflection_blur': 0.3, 'diffuse': 0.02,
    }
    return {
        'default': ambient,
        'white': {'diffuse': 0.8},
        'green': {'diffuse': (0.4, 0.9, 0.4)},
        'red': {'diffuse': (.7, .4, .4)},
        'mirror': {'diffuse': 0.2, 'reflection': 0.7},
        'light': {'diffuse': 1.0, 'emission': (4, 2, .7)},
        'sky': {'diffuse': 0.0, 'emission': (.5, .5, .7)},
        'glass': {'diffuse': 0.1, 'transparency': 0.7, 'reflection': 0.2, 'ior': 1.5},
        'brushed metal': {'diffuse': 0.2, 'reflection': 0.7, 'reflection_blur': 0.1},
        'wax': wax,
    }


def default_spectrum_materials(spectrum):
    materials = default_rgb_materials()
    materials['green'] = {'diffuse': spectrum.gaussian(540, 30) * 0.65 + 0.3}
    materials['red'] = {'diffuse': spectrum.gaussian(670, 30) * 0.6 + 0.2}
    materials['light'] = {'diffuse': 1.0, 'emission': spectrum.black_body(3200) * 7.0}
    materials['sky'] = {'diffuse': 1.0, 'emission': spectrum.black_body(10000)}
    materials['wax']['volume_absorption'] = (1.0 - spectrum.gaussian(670, 100)) * 4.0
    materials['glass']['ior'] = spectrum.linear_dispersion_ior(1.5, 60.0)
    return materials
