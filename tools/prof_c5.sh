#!/bin/bash
OUT=$PWD/gpurun_out/prof_c5
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 $PWD/bench.py --config 5 --steps 3 --warmup 1 --no_cpu_baseline"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
head -12 $OUT/trace/trace_kernel_stats.csv | cut -c1-60,150-260
