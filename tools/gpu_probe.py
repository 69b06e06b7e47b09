"""Ad-hoc GPU probe used during development: HIP vs oracle on small sizes + a timing sweep."""
import runpy, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd.renderer import Renderer
from oracle import oracle

def load(name, size):
    scene = runpy.run_path('raytracer_amd/scenes/%s.py' % name)['scene']
    scene.image.size = size
    return scene

def compare(name, size=(64, 48), passes=8):
    r = Renderer(load(name, size), seed=1)
    print(r.acc.device_info())
    osc = oracle.OracleScene(r)
    tape = r.make_tape(0, 1)
    ids, t, nrm = r.first_hit(tape)
    oids, ot, onrm = osc.first_hit(tape)
    print(name, 'first-hit id mismatches', int((ids != oids).sum()), 't bit-mismatch', int((t.view(np.uint32) != ot.view(np.uint32)).sum()),
          'normal bit-mismatch', int((nrm.view(np.uint32) != onrm.view(np.uint32)).sum()))
    ms, dump = r.render_tape(tape, debug=True)
    oimg, odump, _ = osc.render(tape, debug=True)
    for b in range(dump.shape[0]):
        d = (dump[b].view(np.uint32) != odump[b].view(np.uint32))
        print('  segment', b, 'records differing', int(d.any(axis=(1, 2)).sum()), 'of', dump.shape[1])
    img = r.img.get()
    print('  1-pass image bit-mismatch', int((img.view(np.uint32) != oimg.view(np.uint32)).sum()))
    tape = r.make_tape(1, passes)
    for regen in (0, 1):
        r.img.fill(0)
        r.regenerate = bool(regen)
        r.render_tape(tape)
        img = r.img.get()
        oimg, _, _ = osc.render(tape)
        diff = np.abs(img - oimg)
        print('  regen', regen, passes, 'passes: bit-mismatch', int((img.view(np.uint32) != oimg.view(np.uint32)).sum()),
              'max abs diff', float(diff.max()), 'mean', float(oimg.mean()))

def timing(name, size=(1920, 1080), passes=(1, 4, 16)):
    r = Renderer(load(name, size), seed=1)
    for n in passes:
        tape = r.make_tape(0, n)
        for regen in (0, 1):
            r.regenerate = bool(regen)
            r.render_tape(tape)
            best = min(r.render_tape(tape)[0] for _ in range(3))
            print(name, size, 'passes', n, 'regen', regen, '%.3f ms' % best, '%.1f Msamples/s' % (r.n_pixels * n / best / 1e3))

if __name__ == '__main__':
    for name in sys.argv[1:] or ['scene-dev', 'scene-three-objects']:
        compare(name)
        timing(name)
