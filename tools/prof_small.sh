#!/bin/bash
# per-kernel averages of one bench run, all kernels (usage on the GPU box: bash tools/prof_small.sh)
OUT=$PWD/gpurun_out/prof_small
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 $PWD/bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_folded"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob('$OUT/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(r['Name'].split('(')[0][:46].ljust(46), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
