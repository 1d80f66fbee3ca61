#!/usr/bin/env python3
"""Build an EXPERIMENT variant of the library from a patched copy of one source file (the tree is left untouched):

    python tools/exp/build_variant.py <name> <file.hip> 'old text' 'new text' ['old2' 'new2' ...]

-> tools/probe/libjrr_<name>.so (git-ignored, travels to the GPU box); tools/exp/ab_libs.sh times it against the in-tree build
(bench.py refuses JRR_LIB unless --allow_experiment_lib is given, and prints the library's hash and every JRR_* knob).  The variant's
object is compiled into a scratch directory: nothing is written inside the package."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, 'joint-regressor-refinement_amd')


def main():
    name, fname, pairs = sys.argv[1], sys.argv[2], sys.argv[3:]
    # the other objects the variant links against must be those of the CURRENT tree: bring the in-tree build up to date first
    sys.path.insert(0, PKG)
    import build as _build
    _build.build(verbose=False)
    tmp = tempfile.mkdtemp(prefix='jrr_var_')
    csrc = os.path.join(tmp, 'pkg', 'csrc')
    shutil.copytree(os.path.join(PKG, 'csrc'), csrc)
    os.makedirs(os.path.join(tmp, 'include'))
    shutil.copy(os.path.join(ROOT, 'include', 'jrr.h'), os.path.join(tmp, 'include', 'jrr.h'))
    src = open(os.path.join(csrc, fname)).read()
    for old, new in zip(pairs[0::2], pairs[1::2]):
        assert src.count(old) >= 1, f'pattern not found: {old[:60]!r}'
        src = src.replace(old, new)
    open(os.path.join(csrc, fname), 'w').write(src)
    obj = os.path.join(tmp, fname.replace('.hip', f'.{name}.o'))          # scratch: the package's build/ holds the objects of SOURCES only
    subprocess.check_call([_build._hipcc()] + _build.FLAGS + ['-c', os.path.join(csrc, fname), '-o', obj])
    objs = [os.path.join(PKG, 'build', s.replace('.hip', '.o')) for s in _build.SOURCES if s != fname] + [obj]
    out = os.path.join(ROOT, 'tools', 'probe', f'libjrr_{name}.so')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([_build._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out])
    shutil.rmtree(tmp)
    print(out)


if __name__ == '__main__':
    main()
