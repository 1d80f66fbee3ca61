#!/bin/bash
# usage (GPU box, repo root): bash tools/exp/ab_libs.sh <tag> "<bench flags>" name1 name2 ...
# times the in-tree library and tools/probe/libjrr_<name>.so variants on ONE box, two rounds, interleaved.  The variant is selected
# through JRR_LIB (honoured by _lib.load()): the in-tree library is never overwritten.
TAG=$1; BF=$2; shift; shift
mkdir -p gpurun_out/$TAG
FLAGS="--allow_experiment_lib --no_cpu_baseline --no_folded --no_config5 --no_skin_variants --no_config2 --no_rccl_one_rank --no_support_tiles --no_driver_blocks --min_timed_ms 1200 $BF"
for round in 1 2; do
  for which in base "$@"; do
    if [ $which = base ]; then unset JRR_LIB; else export JRR_LIB=$PWD/tools/probe/libjrr_$which.so; fi
    python bench.py $FLAGS > gpurun_out/$TAG/${which}_${round}.json 2>gpurun_out/$TAG/${which}_${round}.err || tail -3 gpurun_out/$TAG/${which}_${round}.err
    python - <<PY
import json
try:
    j = json.load(open('gpurun_out/$TAG/${which}_${round}.json'))
    k = j['kernels_ms']
    print('%-14s %d  %.4f ms  inner %.4f  fwd %.4f bwd %.4f adj %.4f disc %.4f prep_b %.4f sil %.4f | c1 %.4f' % ('$which', $round, j['ms_per_step'], j['roofline']['whole_step']['inner_only_ms_per_step'], k['k_lbs_fwd'], k['k_lbs_bwd'], k['k_gemm_tn_blend_adjoint'], k.get('pose_disc_gemms', 0), k['k_prep_bwd'], k.get('silhouette_fwd_bwd', 0), j['cadence1']['ms_per_step']))
except Exception as e:
    print('$which', $round, 'failed', e)
PY
  done
done
unset JRR_LIB
