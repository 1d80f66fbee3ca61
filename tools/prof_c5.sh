#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof_c5.sh <tag>
# PMC pass over a short `bench.py --config 5` run: instruction counts of the fused rasteriser (config 5's issue-rate line)
TAG=${1:-c5}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT -o pmc -- python3 $ROOT/bench.py --config 5 --steps 3 --warmup 1 --no_cpu_baseline --no_folded --no_skin_variants --no_driver_blocks --no_rccl_one_rank --min_timed_ms 1 > $OUT/bench.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:48]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
    if 'sil_raster<true' in k:
        m = {n: v / cnt[(k, n)] for n, v in d.items()}
        print(k, json.dumps(m))
        json.dump(m, open('sil_raster_pmc.json', 'w'), indent=1)
PY
