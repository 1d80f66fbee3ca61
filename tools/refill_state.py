#!/usr/bin/env python3
"""Re-fill the measured tables of DESIGN.md (section 0) and README.md ("Measured") from a bench line, using the @PLACEHOLDER@ templates kept
in git (the commit that introduced them, a565059) and tools/fill_docs.py's mapping -- so that the tables quote the committed profiles/ line:

    python tools/refill_state.py [profiles/bench_r06_default.json]
"""
import importlib.util
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEMPLATE_COMMIT = 'a565059'
bench = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'profiles', 'bench_r06_default.json')


def section(text, start, end):
    i = text.index(start)
    return i, text.index(end, i)


def template(path):
    return subprocess.run(['git', '-C', ROOT, 'show', f'{TEMPLATE_COMMIT}:{path}'], capture_output=True, text=True, check=True).stdout


# the value map of fill_docs.py, without letting it rewrite the files
src = open(os.path.join(ROOT, 'tools', 'fill_docs.py')).read()
src = src[:src.index("for name in ('DESIGN.md', 'README.md'):")]
ns = {'__file__': os.path.join(ROOT, 'tools', 'fill_docs.py')}
sys.argv = ['fill_docs.py', bench]
exec(compile(src, 'fill_docs.py', 'exec'), ns)
val = ns['val']
for path, start, end in (('DESIGN.md', '| quantity | value | round 5 |', '(Boxes differ by'), ('README.md', '## Measured (one MI355X', 'RCCL with more than one rank')):
    cur = open(os.path.join(ROOT, path)).read()
    tpl = template(path)
    a0, a1 = section(tpl, start, end)
    filled = re.sub(r'@([A-Z0-9]+)@', lambda m: val.get(m.group(1), m.group(0)), tpl[a0:a1])
    c0, c1 = section(cur, start, end)
    open(os.path.join(ROOT, path), 'w').write(cur[:c0] + filled + cur[c1:])
    print(path, 'table refreshed from', os.path.relpath(bench, ROOT))
