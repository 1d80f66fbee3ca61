#!/bin/bash
# usage (GPU box, repo root): bash tools/exp/ab_lib.sh <other.so> <tag> [bench flags]
# A/B of two builds of the library on ONE box: bench with the in-tree library, with <other.so> copied over it, and again with the first
OTHER=$1; TAG=$2; shift; shift
LIB=joint-regressor-refinement_amd/libjrr_hip.so
mkdir -p gpurun_out/$TAG
cp $LIB /tmp/lib_a.so
FLAGS="--no_cpu_baseline --no_folded --no_config5 --no_skin_variants --no_config2 --min_timed_ms 1500 $@"
for round in 1 2; do
  for which in a b; do
    if [ $which = a ]; then cp /tmp/lib_a.so $LIB; else cp $OTHER $LIB; fi
    python bench.py $FLAGS > gpurun_out/$TAG/${which}${round}.json 2>/dev/null
    python - <<PY
import json
j = json.load(open('gpurun_out/$TAG/${which}${round}.json'))
print('$which$round', j['value'], j['ms_per_step'], 'inner', j['roofline']['whole_step']['inner_only_ms_per_step'], {k: v for k, v in j['kernels_ms'].items()}, 'c1', j['cadence1']['ms_per_step'], 'c1h', j.get('cadence1_host_driven', {}).get('ms_per_step'))
PY
  done
done
cp /tmp/lib_a.so $LIB
