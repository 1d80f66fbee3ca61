#!/usr/bin/env python3
"""Substitute the @PLACEHOLDER@ figures of DESIGN.md / README.md from a bench line (profiles/bench_r06_steps20.json by default):
    python tools/fill_docs.py [bench.json]
so that the state tables quote the measured line itself, not a transcription."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
j = json.load(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'profiles', 'bench_r06_steps20.json')))
k, kr, rf = j['kernels_ms'], j['kernels_roofline'], j['roofline']
small = k['k_prep_fwd'] + k['k_joints_loss'] + k['k_prep_bwd'] + k.get('k_shape_disc', 0.0)
do = j.get('driver_outer_batch') or {}
c1r = (j.get('cadence1_host_driven') or {}).get('with_rccl_one_rank_allreduce') or {}
def f(x, n=4): return ('%.' + str(n) + 'f') % x if x is not None else 'n/a'
val = {
    'HEAD': f(j['value'], 1), 'HEADMS': f(j['ms_per_step']), 'INNER': f(rf['whole_step']['inner_only_ms_per_step']), 'WSF': f(rf['whole_step']['frac'], 3),
    'FWD': f(k['k_lbs_fwd']), 'FWDF': f(rf['frac'], 3), 'BWD': f(k['k_lbs_bwd']), 'BWDF': f(kr['k_lbs_bwd']['frac'], 3),
    'ADJ': f(k['k_gemm_tn_blend_adjoint']), 'ADJF': f(kr['k_blend_adjoint']['frac'], 3),
    'DISC': f(k['pose_disc_gemms']), 'DISCF': f(kr['pose_disc_gemms (4 launches)']['frac'], 3), 'SMALL': f(small, 3),
    'C1': f(j['cadence1']['ms_per_step'], 3), 'C1H': f(j['cadence1_host_driven']['ms_per_step'], 3), 'C1R': f(c1r.get('ms_per_step'), 3),
    'ST': f(j['support_tiles']['ms_per_step'], 3), 'STG': f(j['support_tiles']['kernels_ms'].get('pose_disc_gemms'), 3),
    'STK': f(j['support_tiles']['kernels_ms'].get('k_sup_step'), 3),
    'STC2': f((j['support_tiles'].get('config2_batch1024_joint_loss_only') or {}).get('ms_per_step'), 3),
    'BF': f((j.get('bf16x3_mode') or {}).get('ms_per_step'), 3), 'C2': f(j['config2']['ms_per_step'], 3), 'C2F': f(j['config2']['whole_step']['frac'], 3),
    'C5': f(j['config5']['ms_per_step'], 2), 'RAS': f(j['config5']['kernels_ms']['silhouette_fwd_bwd'], 2),
    'DOA': f(do.get('all_vertex_tiles', {}).get('seconds_per_outer_batch', 0) * 1e3, 1), 'DOAS': '%.1f %%' % (100 * do.get('all_vertex_tiles', {}).get('share_not_inner_loop_or_outer_step', 0)),
    'DOS': f(do.get('support_tiles_default', {}).get('seconds_per_outer_batch', 0) * 1e3, 1), 'DOSS': '%.1f %%' % (100 * do.get('support_tiles_default', {}).get('share_not_inner_loop_or_outer_step', 0)),
    'RD': f((j.get('reference_default') or {}).get('seconds_per_outer_batch', 0) * 1e3, 1),
    'CPU': f(j['cpu_baseline']['value'], 2), 'CPUT': str(j['cpu_baseline']['cores']),
}
for name in ('DESIGN.md', 'README.md'):
    p = os.path.join(ROOT, name)
    s = open(p).read()
    s2 = re.sub(r'@([A-Z0-9]+)@', lambda m: val.get(m.group(1), m.group(0)), s)
    left = sorted(set(re.findall(r'@([A-Z0-9]+)@', s2)))
    open(p, 'w').write(s2)
    print(name, 'filled;', 'unresolved: %s' % left if left else 'no placeholder left')
