#!/bin/bash
# usage (GPU box, repo root): bash tools/exp/ab_variants.sh <tag> name ...   -- like ab_libs.sh (variants selected through JRR_LIB, the
# in-tree library untouched), printing the skin_variants block
TAG=$1; shift
mkdir -p gpurun_out/$TAG
FLAGS="--allow_experiment_lib --no_cpu_baseline --no_folded --no_config5 --no_config2 --no_rccl_one_rank --no_support_tiles --no_driver_blocks --min_timed_ms 800"
for round in 1 2; do
  for which in base "$@"; do
    if [ $which = base ]; then unset JRR_LIB; else export JRR_LIB=$PWD/tools/probe/libjrr_$which.so; fi
    python bench.py $FLAGS > gpurun_out/$TAG/${which}_${round}.json 2>/dev/null
    python - <<PY
import json
j = json.load(open('gpurun_out/$TAG/${which}_${round}.json'))
print('%-8s %d  head %.4f  fwd %.4f |' % ('$which', $round, j['ms_per_step'], j['kernels_ms']['k_lbs_fwd']), {k: v['ms_per_step'] for k, v in j['skin_variants'].items()})
PY
  done
done
unset JRR_LIB
