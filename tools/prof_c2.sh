#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_c2.sh <tag>   -- kernel-trace stats of BASELINE configs[1] (batch 1024, joint loss only)
TAG=${1:-c2}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $OLDPWD/bench.py --config 2 --batch 1024 --steps 100 --warmup 10 --no_cpu_baseline --min_timed_ms 300 --no_folded --no_config5 --no_config2 --no_skin_variants --no_support_tiles --no_driver_blocks --no_rccl_one_rank > $OUT/trace.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob
for f in glob.glob('trace/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:16]:
        print('%-90s calls %6s avg %9.1f ns  total %6.2f %%' % (r['Name'][:90], r['Calls'], float(r['AverageNs']), float(r['Percentage'])))
PY
tail -1 $OUT/trace.log | cut -c1-600
