"""Spiral staircase of thin cuboids and small spheres under a dome; spectral shading
with depth of field.  Values: /root/reference/scenes/scene-stairs.py:8-60."""
from raytracer_amd.objects import Sphere, Cube, HalfSpace
from raytracer_amd.scenes.default_scenes import DefaultScene
from raytracer_amd.shader import SpectrumShader
from raytracer_amd.transformations import Affine


class DomeScene(DefaultScene):

    def set_up_objects_and_camera(self):
        self.materials['dome'] = {'diffuse': 0.0}
        self.materials['red'] = {'diffuse': self.spectrum.gaussian(670, 30) * 0.9 + 0.05,
                                 'reflection': 0.1, 'reflection_blur': 0.05}
        sky_distance = 100
        dome = self.add_object(Sphere((0, 0, 0), sky_distance), 'dome', 'dome')
        self.add_object(HalfSpace((0, 0, -1), sky_distance * 0.7), 'sky', 'sky')
        self.add_object(HalfSpace((0, 0, 1), 0.0), 'white', 'floor')
        self.root_object = dome

        self.camera_position = (1.5, -7.5, 1.5)
        self.direct_camera_towards((-0.5, 0, 1.5))


def make_scene(n_steps=65):
    s = DomeScene(SpectrumShader)
    s.camera_dof_fstop = 0.05
    s.image.size = (1920 // 2, 1080 // 2)
    s.samples_per_pixel = 100000

    step_pose = Affine(translation=(1, 0, 0))
    ball_pose = Affine(translation=(0, 1.5, 0))
    advance = Affine(rotation_axis=(-0.5, -1, 3.1), rotation_deg=10, translation=(0, 0, 0.07))

    for j in range(n_steps):
        step = Cube((0, 0, 0), 1.0)
        step.coordinates = step_pose * Affine(scaling=(0.5, 0.2, 0.02))
        s.add_object(step, {8: 'light', 0: 'red'}.get(j % 16, 'glass'))

        ball = Sphere((0, 0, 0), 0.1)
        ball.coordinates = ball_pose
        s.add_object(ball, 'red')

        step_pose = advance * step_pose
        ball_pose = advance * ball_pose
    return s


scene = make_scene()
