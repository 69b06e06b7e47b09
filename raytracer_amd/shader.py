"""Material model: RGB or spectral colour representation behind one interface.

Mirrors /root/reference/shader.py:4-194.  The tables are property-major: entry
``prop_offset + object_id`` with id 0 = the ambient ('default') material and
``prop_offset = k * (n_objects + 1)`` for the k-th property of a set.
"""
import numpy as np


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
