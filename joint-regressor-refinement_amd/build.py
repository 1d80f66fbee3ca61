"""Build the gfx950 shared library (csrc/*.hip -> libjrr_hip.so) with hipcc, in-tree.

    python joint-regressor-refinement_amd/build.py [--force] [--report] [--clean]

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels with the
repo snapshot to the GPU box.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'libjrr_hip.so')
SOURCES = ['api.hip', 'prep.hip', 'lbs.hip', 'gemm.hip', 'disc.hip', 'eval.hip', 'fold.hip', 'sil.hip', 'sup.hip']
HEADERS = ['jrr_common.h', 'kernels.h', 'dconv.h', 'supk.h', 'proj.h', os.path.join('..', '..', 'include', 'jrr.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function']


def _hipcc() -> str:
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    return 'hipcc'


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def resource_rows(sources=None):
    """Per-kernel resource usage of the CURRENT sources (compiled into a scratch directory with
    -Rpass-analysis=kernel-resource-usage; the in-tree objects are not touched): a list of dicts with the demangled kernel
    name and integer vgpr / agpr / scratch / vspill / occ / lds.  tests/test_host_logic.py fails on any spilled register."""
    import tempfile
    tmp = tempfile.mkdtemp(prefix='jrr_report_')
    jobs = [[_hipcc()] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', os.path.join(tmp, src.replace('.hip', '.o')),
                                  '-Rpass-analysis=kernel-resource-usage'] for src in (sources or SOURCES)]
    with ThreadPoolExecutor(max_workers=min(len(jobs), 8)) as ex:
        outs = list(ex.map(lambda cmd: subprocess.run(cmd, capture_output=True, text=True), jobs))
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    rows = []
    for p in outs:
        if p.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + p.stderr[-2000:])
        rows += _parse(p.stdout + p.stderr)
    return rows


def _parse(out: str):
    import re
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r'remark: Function Name: (\S+)', line)
        if m:
            name = m.group(1)
            try:
                name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
            except OSError:
                pass
            cur = {'name': name.split('(')[0]}
            rows.append(cur)
            continue
        for key, pat in (('vgpr', r'remark:\s+VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'),
                         ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)'),
                         ('vspill', r'VGPRs Spill: (\d+)')):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return [r for r in rows if 'vgpr' in r]


def _summarise(out: str) -> str:
    """Condense -Rpass-analysis=kernel-resource-usage remarks to one line per kernel."""
    import re
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r'remark: Function Name: (\S+)', line)
        if m:
            name = m.group(1)
            try:
                name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
            except OSError:
                pass
            cur = {'name': name.split('(')[0][:70]}
            rows.append(cur)
            continue
        for key, pat in (('vgpr', r'remark:\s+VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'),
                         ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)'),
                         ('sgpr', r'remark:\s+SGPRs: (\d+)'), ('vspill', r'VGPRs Spill: (\d+)')):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = m.group(1)
        if 'warning' in line or 'error' in line:
            rows.append({'name': line})
    txt = ''
    for r in rows:
        if 'vgpr' in r:
            txt += (f"  {r['name']:<72s} vgpr={r.get('vgpr')} agpr={r.get('agpr')} sgpr={r.get('sgpr')} scratch={r.get('scratch')} "
                    f"vspill={r.get('vspill')} occ={r.get('occ')} lds={r.get('lds')}\n")
        else:
            txt += r['name'] + '\n'
    return txt


def clean(verbose: bool = True) -> list:
    """Remove every file of the object directory that is not the object of a source in SOURCES (experiment variants built by hand or by
    older tools): they would otherwise travel to the GPU box with every snapshot.  Returns the removed names."""
    keep = {s.replace('.hip', '.o') for s in SOURCES}
    gone = []
    if os.path.isdir(OBJ):
        for name in sorted(os.listdir(OBJ)):
            if name not in keep:
                os.remove(os.path.join(OBJ, name))
                gone.append(name)
    if verbose and gone:
        print(f'[jrr build] removed {len(gone)} stale objects from {OBJ}')
    return gone


def build(force: bool = False, report: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace('.hip', '.o'))
        if force or report or _stale(o, [s] + hdrs):
            cmd = [_hipcc()] + FLAGS + ['-c', s, '-o', o]
            if report:
                cmd.append('-Rpass-analysis=kernel-resource-usage')
            jobs.append((src, cmd))

    def run(job):
        src, cmd = job
        p = subprocess.run(cmd, capture_output=True, text=True)
        return src, p.returncode, p.stdout + p.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), 8)) as ex:
            for src, rc, out in ex.map(run, jobs):
                if rc != 0:
                    sys.stderr.write(out)
                elif report:
                    sys.stderr.write(_summarise(out))
                if rc != 0:
                    raise RuntimeError(f'hipcc failed on {src}')
                if verbose:
                    print(f'[jrr build] compiled {src}')
    objs = [os.path.join(OBJ, s.replace('.hip', '.o')) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            sys.stderr.write(p.stdout + p.stderr)
            raise RuntimeError('hipcc link failed')
        if verbose:
            print(f'[jrr build] linked {LIB}')
    return LIB


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--force', action='store_true')
    ap.add_argument('--report', action='store_true', help='print per-kernel VGPR/LDS/occupancy')
    ap.add_argument('--clean', action='store_true', help='remove objects that do not belong to SOURCES (stale experiment variants)')
    a = ap.parse_args()
    if a.clean:
        clean()
    build(force=a.force, report=a.report)
