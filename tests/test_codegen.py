"""Code generation: every shipped scene renders to one HIP translation unit that hiprtc
compiles for gfx950 (no device needed), and the generator's de-duplication / clash rules."""
import pytest

from raytracer_amd import accelerator, hip_compiler
from raytracer_amd.renderer import Renderer
from raytracer_amd import objects as shapes
from conftest import load_scene


@pytest.mark.parametrize('name', ['scene-dev', 'scene-three-objects', 'scene-test', 'scene-dev-spectral',
                                  'scene-stairs-small'])
def test_scene_program_compiles_for_gfx950(name):
    r = Renderer(load_scene(name, (64, 48)), device=False, seed=0)
    code = r.program_code
    assert 'extern "C" __global__' in code and 'clr_render' in code and 'clr_first_hit' in code
    assert accelerator.compile_only(code, r.build_options, 'gfx950') > 10000


def test_one_function_per_class_not_per_instance():
    r = Renderer(load_scene('scene-test', (16, 8)), device=False, seed=0)
    code = r.program_code
    # two Octahedron instances share one tracer; the two generic ConvexIntersections do not
    assert code.count('void Octahedron_tracer(') == 2          # declaration + definition
    assert 'ConvexIntersection_CylinderComponent_SphereComponent_LayerComponent_tracer' in code
    assert code.count('void LayerComponent_tracer(') == 2


def test_name_clash_is_an_error():
    class Impostor(shapes.Sphere):
        def template_name(self):
            return 'HalfSpace'

        @property
        def tracer_function_name(self):
            return 'Sphere_tracer'

    scene = load_scene('scene-dev', (16, 8))
    scene.add_object(Impostor((0, 0, 0), 1.0), 'white')
    with pytest.raises(RuntimeError):
        Renderer(scene, device=False, seed=0)


def test_parity_options_disable_contraction():
    assert '-ffp-contract=off' in hip_compiler.PARITY_OPTIONS


def test_empty_scene_still_generates_a_program():
    from raytracer_amd.scenes.default_scenes import BoxScene
    scene = BoxScene()
    scene.objects[:] = []
    scene.image.size = (16, 8)
    r = Renderer(scene, device=False, seed=0)
    assert accelerator.compile_only(r.program_code, r.build_options) > 1000


def test_large_scenes_keep_parameters_in_constant_memory():
    """More than 48 objects: the parameter table is filled at run time instead of baked."""
    r = Renderer(load_scene('scene-stairs-small', (16, 8)), device=False, seed=0)
    assert r.bake_parameters and '0x1' in r.program_code
    big = Renderer(load_scene('scene-stairs-small', (16, 8)), device=False, seed=0, bake_parameters=False)
    assert '__constant__ float4 clr_params[' in big.program_code
    assert accelerator.compile_only(big.program_code, big.build_options) > 1000


def test_tone_mapping_program_compiles_for_gfx950():
    import os
    from raytracer_amd import accelerator
    from raytracer_amd.hip_compiler import TEMPLATE_DIR
    source = open(os.path.join(TEMPLATE_DIR, 'tonemap.hip')).read()
    assert accelerator.compile_only(source, ['-O3']) > 1000
