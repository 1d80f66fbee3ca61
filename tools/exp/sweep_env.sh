#!/bin/bash
# usage (GPU box, repo root): bash tools/exp/sweep_env.sh <tag> <ENV_NAME> "<bench flags>" v1 v2 ...
# one short bench per value of an environment knob, two rounds interleaved; prints the per-kernel times
TAG=$1; NAME=$2; BF=$3; shift; shift; shift
mkdir -p gpurun_out/$TAG
FLAGS="--no_cpu_baseline --no_folded --no_config5 --no_skin_variants --no_config2 --no_rccl_one_rank --no_support_tiles --min_timed_ms 1000 $BF"
for round in 1 2; do
  for val in "$@"; do
    env $NAME=$val python bench.py $FLAGS > gpurun_out/$TAG/${val}_${round}.json 2>/dev/null
    python - <<PY
import json
j = json.load(open('gpurun_out/$TAG/${val}_${round}.json')); k = j['kernels_ms']
print('$NAME=%-6s %d  %.4f ms  inner %.4f  fwd %.4f bwd %.4f adj %.4f disc %.4f' % ('$val', $round, j['ms_per_step'], j['roofline']['whole_step']['inner_only_ms_per_step'], k['k_lbs_fwd'], k['k_lbs_bwd'], k['k_gemm_tn_blend_adjoint'], k.get('pose_disc_gemms', 0)))
PY
  done
done
