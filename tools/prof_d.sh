#!/bin/bash
# kernel breakdown of the default bench (incl. the separately timed outer-step work)
OUT=$PWD/gpurun_out/prof_d
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 $PWD/bench.py --steps 5 --warmup 1 --no_cpu_baseline --no_folded"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
