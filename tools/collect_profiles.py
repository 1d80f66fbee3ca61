#!/usr/bin/env python3
"""Copy the evidence of `tools/final.sh <tag>` from gpurun_out/ (scratch) into profiles/ (tracked) under the round's names:

    python tools/collect_profiles.py <tag> <round>        e.g.  python tools/collect_profiles.py r06 r06

bench lines -> profiles/bench_<round>_{steps20,default,config2_b1024,config5}.json, the kernel trace -> rocprof_<round>_kernel_stats.csv,
the trace + PMC summary -> rocprof_<round>_summary.json, and profiles/pmc_traffic.json (the static counts bench.py prints beside the live
durations: HBM bytes per launch, issued matrix instructions, the rasteriser's SQ counters) refreshed from the same run."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
G, P = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')
for src, dst in (('bench_steps20.json', 'steps20'), ('bench_default.json', 'default'), ('bench_config2.json', 'config2_b1024'), ('bench_config5.json', 'config5')):
    line = [l for l in open(os.path.join(G, tag, src)) if l.startswith('{')][-1]
    json.loads(line)
    open(os.path.join(P, f'bench_{rnd}_{dst}.json'), 'w').write(line)
shutil.copy(os.path.join(G, f'prof_{tag}', 'trace', 'trace_kernel_stats.csv'), os.path.join(P, f'rocprof_{rnd}_kernel_stats.csv'))
shutil.copy(os.path.join(G, f'prof_{tag}', 'summary.json'), os.path.join(P, f'rocprof_{rnd}_summary.json'))
new = json.load(open(os.path.join(G, f'prof_{tag}', 'pmc_traffic.json')))
ras = os.path.join(G, f'pmc_{tag}c5', 'sil_raster_pmc.json')
if os.path.exists(ras):
    m = {k: int(v) for k, v in json.load(open(ras)).items()}
    new['k_sil_raster_adj_pmc_per_launch_b4096'] = m
    new['k_sil_raster_adj_note'] = (f"tools/prof_c5.sh {tag}c5: SQ_WAIT_ANY / SQ_WAVE_CYCLES = {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f}, "
                                    f"SQ_INSTS_VALU {m['SQ_INSTS_VALU']:.3g} (the rasteriser is unchanged since round 5)")
new['recorded'] = f'{rnd}: tools/final.sh {tag} (tools/prof.sh + tools/prof_c5.sh on one box, final build)'
json.dump(new, open(os.path.join(P, 'pmc_traffic.json'), 'w'), indent=1)
for f in ('gpu_suite.txt', 'gpu_suite_variants.txt'):
    s = os.path.join(G, tag, f)
    if os.path.exists(s):
        shutil.copy(s, os.path.join(P, f.replace('gpu_suite', f'gpu_suite_{rnd}')))
print('profiles/ refreshed from', tag)
