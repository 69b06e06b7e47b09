"""Names the reference's scene files import from `scenes.default_scenes`."""
from .library import (default_settings, make_world_box, DefaultScene, BoxScene,  # noqa: F401
                      default_rgb_materials, default_spectrum_materials)
