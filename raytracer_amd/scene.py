"""Scene = camera + objects + materials + render settings (reference scene.py:5-58)."""
import numpy as np
from . import geometry as camera
from .imgutils import ColorEncodingSettings


class Scene:

    class Object:
        """A shape (Tracer) with a material name; ``bidirectional_light`` marks it as a
        next-event-estimation light (the Tracer must implement the light interface)."""

        def __init__(self, tracer, material, name=None):
            self.tracer = tracer
            self.material = material
            self.name = name
            self.bidirectional_light = False

    class ImageSettings(ColorEncodingSettings):
        pass

    def add_object(self, tracer, material, name=None):
        obj = Scene.Object(tracer, material, name)
        self.objects.append(obj)
        return obj

    def get_objects(self, name):
        return [obj for obj in self.objects if obj.name == name]

    def get_object(self, name):
        found = self.get_objects(name)
        if len(found) == 1:
            return found[0]
        if not found:
            raise KeyError("No object named '%s'" % name)
        raise KeyError("Multiple objects in the scene are called '%s'" % name)

    def delete_objects(self, name):
        self.objects[:] = [obj for obj in self.objects if obj.name != name]

    def direct_camera_towards(self, target):
        self.camera_direction = np.array(target) - np.array(self.camera_position)
        self.camera_sharp_distance = np.linalg.norm(self.camera_direction)

    def get_camera_rotmat(self):
        return camera.camera_rotmat(self.camera_direction, self.camera_up)

    def get_camera_rays(self):
        return camera.camera_rays(self.image.size, self.camera_flat_ccd, self.camera_fov,
                                  self.camera_direction, self.camera_up)

    def get_number_of_camera_rays(self):
        return self.image.size[0] * self.image.size[1]
