#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_rev.sh <tag>
# FETCH_SIZE / WRITE_SIZE of the three LBS kernels with k_lbs_bwd16 walking its tiles in the forward kernel's order (default) and in
# REVERSE (JRR_BWD16_REV=1: the tiles k_lbs_fwd wrote last are read first) -- DESIGN.md section 8, stage B.  The variable is exported
# (never `env` behind rocprofv3's `--`).
TAG=${1:-rev}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 $PWD/bench.py --steps 5 --warmup 2 --no_cpu_baseline --min_timed_ms 1 --no_skin_variants --no_folded --no_config2 --no_config5 --no_bf16x3 --no_rccl_one_rank --no_support_tiles --no_driver_blocks"
cd /tmp
for mode in fwd_order reversed; do
  if [ $mode = reversed ]; then export JRR_BWD16_REV=1; else unset JRR_BWD16_REV; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${mode}_fetch -o pmc -- $CMD > $OUT/${mode}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${mode}_write -o pmc -- $CMD > $OUT/${mode}_write.log 2>&1
done
unset JRR_BWD16_REV
cd $OUT
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for mode in ('fwd_order', 'reversed'):
    for kind, ctr in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
        agg, cnt = collections.defaultdict(float), collections.Counter()
        for f in glob.glob(f'{mode}_{kind}/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                if r['Counter_Name'] != ctr:
                    continue
                k = r['Kernel_Name'].split('(')[0][:48]
                agg[k] += float(r['Counter_Value']); cnt[k] += 1
        for k in agg:
            if any(s in k for s in ('k_lbs_fwd', 'k_lbs_bwd16', 'k_blend_adjoint')):
                out.setdefault(k, {})[f'{mode}_{ctr}_KB_per_launch'] = round(agg[k] / cnt[k], 1)
json.dump(out, open('rev_traffic.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
