#!/bin/bash
# usage (GPU box, repo root): bash tools/exp/sweep_env_c2.sh <tag> <ENV_NAME> v1 v2 ...  -- BASELINE configs[1] (batch 1024, joint loss only) per value
TAG=$1; NAME=$2; shift; shift
mkdir -p gpurun_out/$TAG
FLAGS="--config 2 --batch 1024 --no_cpu_baseline --no_folded --no_config5 --no_skin_variants --no_config2 --no_rccl_one_rank --no_support_tiles --no_driver_blocks --min_timed_ms 500"
for round in 1 2; do
  for val in "$@"; do
    env $NAME=$val python bench.py $FLAGS > gpurun_out/$TAG/${val}_${round}.json 2>/dev/null
    python - <<PY
import json
j = json.load(open('gpurun_out/$TAG/${val}_${round}.json')); k = j['kernels_ms']
print('$NAME=%-6s %d  %.4f ms ' % ('$val', $round, j['ms_per_step']), {a: round(b, 4) for a, b in k.items()})
PY
  done
done
