#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_tiles.sh <tag>
# HBM bytes per launch of the LIST instantiations (support-tile iterations; tools/exp/support_tiles_check.py), separate PMC passes
TAG=${1:-tiles}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 $PWD/tools/exp/support_tiles_check.py 4096 5"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob, collections, json
res = {}
for name, ctr in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    agg = collections.defaultdict(list)
    for f in glob.glob(name + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != ctr:
                continue
            agg[(r['Kernel_Name'].split('(')[0], r.get('Grid_Size', ''))].append(float(r['Counter_Value']))
    for (k, grid), v in agg.items():
        if any(t in k for t in ('k_lbs_fwd<true, false, 8, false', 'k_lbs_bwd16<0, 8, false', 'k_blend_adjoint')):
            res.setdefault(k, {})[ctr + '_KB_per_launch'] = sorted(set(round(x) for x in v))[:6]
json.dump(res, open('tiles_traffic.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
