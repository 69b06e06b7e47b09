import os
import runpy
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENE_DIR = os.path.join(ROOT, 'raytracer_amd', 'scenes')
GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an AMD GPU (run on the MI355X box)")


def load_scene(name, size=None, **overrides):
    """A fresh copy of a shipped scene ('scene-dev', 'scene-three-objects', 'scene-test')."""
    if name == 'scene-dev-spectral':
        scene = runpy.run_path(os.path.join(SCENE_DIR, 'scene-dev.py'))['make_scene'](spectral=True)
    elif name == 'scene-stairs-small':     # 12 of the 65 steps: same classes, quicker to compile
        scene = runpy.run_path(os.path.join(SCENE_DIR, 'scene-stairs.py'))['make_scene'](12)
    elif name.startswith('variant-') or name.startswith('spectral-'):
        import types
        import golden_scenes
        from raytracer_amd import objects as shapes
        from raytracer_amd.geometry import Affine
        from raytracer_amd.shading import SpectrumShader
        from raytracer_amd.scenes.default_scenes import BoxScene, DefaultScene
        api = types.SimpleNamespace(BoxScene=BoxScene, DefaultScene=DefaultScene, SpectrumShader=SpectrumShader,
                                    Affine=Affine, Cube=shapes.Cube, Sphere=shapes.Sphere,
                                    TriangleMesh=shapes.TriangleMesh, Octree=shapes.Octree,
                                    Cylinder=shapes.Cylinder, HalfSpace=shapes.HalfSpace)
        scene = golden_scenes.VARIANTS[name](api)
    elif name.startswith('scene-mesh'):
        # 'scene-mesh:<n_u>x<n_v>:<layout>:<overlap>', layout 'flat-lanes' = flat, not pooled, 'flat-ropes', 'flat-two';
        # overlap 'reference' (default here: the reference's tree) or 'sat'
        parts = name.split(':')[1:]
        grid, layout, overlap = parts + ['120x60', 'flat', 'reference'][len(parts):]
        n_u, n_v = (int(v) for v in grid.split('x'))
        make = runpy.run_path(os.path.join(SCENE_DIR, 'scene-mesh.py'), run_name='lib')['make_scene']
        if layout == 'flat-lanes':
            scene = make(n_u, n_v, layout='flat', pooled=False, overlap=overlap)
        elif layout == 'flat-ropes':        # pooled tests, next leaf through neighbour links
            scene = make(n_u, n_v, layout='flat', pooled=True, ropes=True, overlap=overlap)
        elif layout == 'flat-two':          # pooled tests, node records for a two-level descent
            scene = make(n_u, n_v, layout='flat', pooled=True, two_level=True, overlap=overlap)
        elif layout == 'flat':              # pooled tests, plain node records
            scene = make(n_u, n_v, layout='flat', pooled=True, two_level=False, overlap=overlap)
        else:
            scene = make(n_u, n_v, layout=layout, overlap=overlap)
    else:
        scene = runpy.run_path(os.path.join(SCENE_DIR, name + '.py'))['scene']
    if size is not None:
        scene.image.size = tuple(size)
    for key, value in overrides.items():
        setattr(scene, key, value)
    return scene


@pytest.fixture
def scene_loader():
    return load_scene
