"""Code generator: renders the Jinja2 HIP templates into one scene-specialised translation unit.

Takes the place of the reference's ``cl_compiler.py`` (/root/reference/cl_compiler.py:6-62):
same template environment conventions (``###`` line statements, per-object
``make_functions`` / ``make_kernels`` hooks, de-duplication by body, name-clash check) and
the same render context (``shader``, ``renderer``, ``objects``, ``n_objects``, ``functions``).
The output is compiled at run time with hiprtc (see accelerator.py).
"""
import os
import jinja2

TEMPLATE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hip_templates')

# parity mode: no FMA contraction and IEEE division / square root, so that the device
# arithmetic can be followed step by step by the CPU oracle
# -fno-slp-vectorize changes no result: the SLP vectoriser pairs neighbouring scalar FP32
# operations into v_pk_* instructions, and assembling the register pairs costs more moves and
# registers than the pairing saves (measured: 14 % to 2x slower with it)
PARITY_OPTIONS = ['-O3', '-ffp-contract=off', '-fno-slp-vectorize']
FAST_OPTIONS = ['-O3', '-ffp-contract=fast', '-fno-slp-vectorize']


def make_template_env():
    return jinja2.Environment(loader=jinja2.FileSystemLoader(TEMPLATE_DIR),
                              line_statement_prefix='###', undefined=jinja2.StrictUndefined)


class Compiler:

    def __init__(self, renderer):
        self.scene = renderer.scene
        self.shader = renderer.shader
        self.renderer = renderer

    def make_program(self):
        env = make_template_env()
        kernels, functions = collect_tracer_kernels(self.scene.objects, env)
        declarations = [body[:body.find('{')].rstrip() + ';' for body in functions]
        return env.get_template('main.hip').render({
            'shader': self.shader,
            'renderer': self.renderer,
            'objects': self.scene.objects,
            'n_objects': len(self.scene.objects),
            'functions': {'declarations': declarations, 'definitions': functions,
                          'kernels': kernels},
        })


def collect_tracer_kernels(objects, template_env):
    """Distinct shape functions and per-class steps of a scene, in a deterministic order."""
    seen = {}

    def push(store, name, body):
        body = body.strip()
        if name in seen:
            if seen[name] != body:
                raise RuntimeError("function name clash: two different bodies for %s" % name)
            return
        seen[name] = body
        store.append(body)

    functions, kernels = [], []
    for obj in objects:
        for name, body in sorted(obj.tracer.make_functions(template_env).items()):
            push(functions, name, body)
        for name, body in sorted(obj.tracer.make_kernels(template_env).items()):
            push(kernels, name, body)
    return kernels, functions
