#!/usr/bin/env python3
"""Condensed control-flow / instruction-mix trace of one kernel in a hipcc -save-temps .s file.
usage: asm_trace.py file.s kernel_name_substring"""
import sys
S, key = sys.argv[1], sys.argv[2]
lines = open(S).read().split('\n')
start = [i for i, l in enumerate(lines) if key in l.split(':')[0] and ': ' in l and l[0] not in ' \t.;'][0]
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end]
print(len(body), 'lines')
keys = ['mfma', 'sst', 'sld', 'gl', 'gs', 'dsr', 'dsw', 'valu', 'wait', 'bar']
cnt = {k: 0 for k in keys}
for l in body:
    t = l.strip()
    if t.startswith('.LBB') or t.startswith('s_cbranch') or t.startswith('s_branch'):
        print({k: v for k, v in cnt.items() if v}, t.split()[0:2])
        cnt = {k: 0 for k in keys}
    elif t.startswith('v_mfma'): cnt['mfma'] += 1
    elif t.startswith('scratch_store'): cnt['sst'] += 1
    elif t.startswith('scratch_load'): cnt['sld'] += 1
    elif t.startswith('global_load') or t.startswith('buffer_load'): cnt['gl'] += 1
    elif t.startswith('global_store') or t.startswith('buffer_store'): cnt['gs'] += 1
    elif t.startswith('ds_read') or t.startswith('ds_load'): cnt['dsr'] += 1
    elif t.startswith('ds_write') or t.startswith('ds_store'): cnt['dsw'] += 1
    elif t.startswith('s_waitcnt'): cnt['wait'] += 1
    elif t.startswith('s_barrier'): cnt['bar'] += 1
    elif t.startswith('v_'): cnt['valu'] += 1
print({k: v for k, v in cnt.items() if v})
