"""raytracer_amd - MI355X-native back end for the clray path tracer's per-pixel loop."""
from .scene import Scene  # noqa: F401
from .renderer import Renderer  # noqa: F401


def install_flat_aliases():
    """Make the reference's flat module names (``objects``, ``scenes.default_scenes``,
    ``shader``, ``transformations`` ...) resolve to this package, so scene files written for
    the reference import unchanged."""
    import importlib
    import sys
    for name in ('utils', 'transformations', 'camera', 'tracer', 'objects', 'shader', 'spectrum',
                 'scene', 'imgutils', 'mesh_formats', 'scenes', 'scenes.default_scenes', 'renderer'):
        if name not in sys.modules:
            sys.modules[name] = importlib.import_module('raytracer_amd.' + name)
