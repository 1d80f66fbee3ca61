#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_sup.sh <batch>
# kernel trace (rocprofv3 --kernel-trace --stats) of tools/exp/support_tiles_check.py at one batch size: the default (support-vertex)
# iteration's launches -- the discriminator GEMM tile variants (template arguments WM, WAVES_M, WAVES_N, EPI, BTR, KS) and k_sup_step
B=${1:-512}
OUT=$PWD/gpurun_out/trace_sup$B
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 $ROOT/tools/exp/support_tiles_check.py $B 100 > $OUT/run.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob
for f in glob.glob('**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:30]:
        print(f"{r.get('Name','')[:100]:<100s} calls={r.get('Calls'):>5s} avg_us={float(r.get('AverageNs',0))/1e3:9.2f} pct={r.get('Percentage')}")
PY
grep -E "ms/iteration" $OUT/run.log
