#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof.sh <tag>
# kernel-trace stats + separate PMC passes (never combined with tracing domains), summaries -> gpurun_out/prof_<tag>/
TAG=${1:-run}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# kernel trace: the default bench command itself (100 timed steps); PMC passes: a short run (counters serialise kernels)
FULL="python3 $PWD/bench.py --no_cpu_baseline --min_timed_ms 600 --no_config2 --no_rccl_one_rank --no_driver_blocks"
CMD="python3 $PWD/bench.py --steps 5 --warmup 2 --no_cpu_baseline --min_timed_ms 1 --no_skin_variants --no_folded --no_config2 --no_rccl_one_rank --no_support_tiles --no_driver_blocks"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $FULL > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o pmc -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob, collections, json, os
out = {}
for f in glob.glob('trace/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    out['kernel_stats'] = [{k: r[k] for k in r} for r in rows[:25]]
for name in ('pmc_sq', 'pmc_fetch', 'pmc_write'):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(name + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:60]
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[(k, r['Counter_Name'])] += 1
    out[name] = {k: {c: v / cnt[(k, c)] for c, v in d.items()} for k, d in agg.items()}
# HBM traffic of the dominant kernel, per launch, corrected as MI355X_MICROARCH.md section HBM prescribes:
# FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane)
# coalesced reads (all of this kernel's reads are 16 B/lane LDS-DMA), WRITE_SIZE is exact.
def pick(d, key):
    for k, v in d.items():
        if key in k:
            return v
    return {}
f = pick(out['pmc_fetch'], 'k_lbs_fwd<true, false').get('FETCH_SIZE')
w = pick(out['pmc_write'], 'k_lbs_fwd<true, false').get('WRITE_SIZE')
if f is not None and w is not None:
    out['traffic'] = {'k_lbs_fwd_hbm_bytes_per_launch': int((2 * f + w) * 1024), 'FETCH_SIZE_KB': f, 'WRITE_SIZE_KB': w,
                      'correction': 'bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)'}
    for key, name in (('k_lbs_bwd', 'k_lbs_bwd'), ('k_blend_adjoint', 'k_blend_adjoint'), ('k_sil_raster<true', 'k_sil_raster_adj')):
        ff, ww = pick(out['pmc_fetch'], key).get('FETCH_SIZE'), pick(out['pmc_write'], key).get('WRITE_SIZE')
        if ff is not None and ww is not None:
            out['traffic'][name + '_hbm_bytes_per_launch'] = int((2 * ff + ww) * 1024)
            out['traffic'][name + '_FETCH_SIZE_KB'] = ff
            out['traffic'][name + '_WRITE_SIZE_KB'] = ww
    # ISSUED matrix work per launch: SQ_INSTS_MFMA x FLOP per instruction (32x32x2: 4096; 16x16x4 and the four-block 16x16x1:
    # 2048; k_lbs_fwd mixes 378 + 48 per tile and wave: 330 blend steps incl. 3 against the zero rows 218, 219, 222, 223 of the
    # K-quad layout, 48 skinning, 48 four-block regressor), printed by bench.py beside every algorithmic figure
    sq = out.get('pmc_sq', {})
    def mf(key):
        return pick(sq, key).get('SQ_INSTS_MFMA')
    issued = {}
    n = mf('k_lbs_fwd<true, false, 8, false, false>')
    if n: issued['k_lbs_fwd'] = {'sq_insts_mfma': int(n), 'flop_per_launch': int(n * (378 * 4096 + 48 * 2048) / 426)}
    n = mf('k_lbs_bwd16<0, 8, false, false, 1>')
    if n: issued['k_lbs_bwd'] = {'sq_insts_mfma': int(n), 'flop_per_launch': int(n * 2048)}
    n = mf('k_blend_adjoint')
    if n: issued['k_blend_adjoint'] = {'sq_insts_mfma': int(n), 'flop_per_launch': int(n * 4096)}
    nd = [v.get('SQ_INSTS_MFMA') for k, v in sq.items() if 'k_disc_gemm' in k]
    if nd and all(nd): issued['pose_disc_gemms'] = {'sq_insts_mfma': int(sum(nd)), 'flop_per_launch': int(sum(nd) * 4096), 'launches': len(nd)}
    out['traffic']['mfma_issued_b4096'] = issued
    for key, name in (('k_lbs_fwd<true, false, 8, false, false>', 'k_lbs_fwd'), ('k_lbs_bwd16<0, 8, false, false, 1>', 'k_lbs_bwd'), ('k_blend_adjoint', 'k_blend_adjoint')):
        d = pick(sq, key)
        if d: out['traffic'][name + '_sq_per_launch'] = {c: int(v) for c, v in d.items()}
    json.dump(out['traffic'], open('pmc_traffic.json', 'w'), indent=1)
json.dump(out, open('summary.json', 'w'), indent=1)
for r in out.get('kernel_stats', []):
    print(r.get('Name', '')[:70], r.get('Calls'), r.get('TotalDurationNs'), r.get('AverageNs'), r.get('Percentage'))
PY
ls -la $OUT
