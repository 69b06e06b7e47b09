"""Development check (container only): character-level similarity of our sources against the
reference's files of the same name or of similar size.  Prints the pairs above 0.5."""
import difflib
import glob
import os
import sys

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources():
    mine = []
    for pattern in ('raytracer_amd/**/*.py', 'raytracer_amd/hip_templates/**/*.hip',
                    'raytracer_amd/hip_templates/*.inl', '*.py', 'tests/*.py', 'tools/*.py',
                    'oracle/*.py', 'oracle/*.c', 'oracle/*.h', 'raytracer_amd/csrc/*.cpp'):
        mine += glob.glob(os.path.join(ROOT, pattern), recursive=True)
    ref = []
    for pattern in ('**/*.py', 'cl_templates/**/*.cl'):
        ref += glob.glob(os.path.join(REF, pattern), recursive=True)
    return sorted(set(mine)), sorted(set(ref))


def main():
    mine, ref = sources()
    ref_text = [(a, open(a, errors='replace').read()) for a in ref]
    found = []
    for b in mine:
        tb = open(b, errors='replace').read()
        if len(tb) < 200:
            continue
        stem_b = os.path.splitext(os.path.basename(b))[0]
        for a, ta in ref_text:
            if len(ta) < 200:
                continue
            same_name = os.path.splitext(os.path.basename(a))[0] == stem_b
            if not (same_name or 0.5 < len(ta) / float(len(tb)) < 2.0):
                continue
            sm = difflib.SequenceMatcher(None, ta, tb, autojunk=False)
            if sm.quick_ratio() < 0.5:
                continue
            r = sm.ratio()
            if r > 0.5:
                found.append((r, os.path.relpath(b, ROOT), os.path.relpath(a, REF)))
    for r, b, a in sorted(found, reverse=True):
        print('%.2f  %s  <->  %s' % (r, b, a))
    return 1 if found and max(found)[0] > 0.6 else 0


if __name__ == '__main__':
    sys.exit(main())
