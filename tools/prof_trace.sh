#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof_trace.sh <tag> [bench flags]
# kernel trace only (rocprofv3 --kernel-trace --stats) of a short bench run; prints the per-kernel averages
TAG=${1:-t}; shift
OUT=$PWD/gpurun_out/trace_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_folded --no_config5 "$@" > $OUT/bench.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob
for f in glob.glob('**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:40]:
        print(f"{r.get('Name','')[:90]:<90s} calls={r.get('Calls'):>5s} avg_us={float(r.get('AverageNs',0))/1e3:9.2f} pct={r.get('Percentage')}")
PY
tail -c 400 $OUT/bench.log
