#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof_pmc.sh <tag> "<counters>" [bench flags]
# one rocprofv3 --pmc pass (never combined with tracing domains) over a 5-step bench; prints per-kernel counter means
TAG=${1:-p}; CTRS=${2:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"}; shift; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o pmc -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_folded --no_config5 --min_timed_ms 1 "$@" > $OUT/bench.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:48]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
names = sorted({c for d in agg.values() for c in d})
print('kernel'.ljust(50), ' '.join(n[-14:].rjust(15) for n in names))
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', kv[1].get(names[0], 0)) / max(1, cnt[(kv[0], names[0])]))[:14]:
    print(k.ljust(50), ' '.join(f"{d.get(n, 0) / max(1, cnt[(k, n)]):15.4g}" for n in names))
PY
tail -c 300 $OUT/bench.log
