"""raytracer_amd - MI355X-native back end for the clray path tracer's per-pixel loop."""
from .scene import Scene  # noqa: F401
from .renderer import Renderer  # noqa: F401


# reference module name -> module of this package that provides the same public names
FLAT_ALIASES = {
    'utils': 'geometry', 'transformations': 'geometry', 'camera': 'geometry',
    'tracer': 'objects.base', 'objects': 'objects', 'mesh_formats': 'objects.meshes',
    'shader': 'shading', 'spectrum': 'shading', 'scene': 'scene', 'imgutils': 'imgutils',
    'scenes': 'scenes', 'scenes.default_scenes': 'scenes.default_scenes', 'renderer': 'renderer',
}


def install_flat_aliases():
    """Make the reference's flat module names (``objects``, ``scenes.default_scenes``,
    ``shader``, ``transformations`` ...) resolve to this package, so scene files written for
    the reference import unchanged."""
    import importlib
    import sys
    for name, target in FLAT_ALIASES.items():
        if name not in sys.modules:
            sys.modules[name] = importlib.import_module('raytracer_amd.' + target)
