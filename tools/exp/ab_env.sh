#!/bin/bash
# usage (GPU box, repo root): bash tools/exp/ab_env.sh <tag> "<bench flags>" "ENV1=a ENV2=b" "ENV3=c" ...
# one short bench per environment SETTING (each argument is a space-separated list of assignments; "-" = none), two rounds interleaved
TAG=$1; BF=$2; shift; shift
mkdir -p gpurun_out/$TAG
FLAGS="--no_cpu_baseline --no_folded --no_config5 --no_skin_variants --no_config2 --no_rccl_one_rank --no_driver_blocks --min_timed_ms 1000 $BF"
for round in 1 2; do
  i=0
  for setting in "$@"; do
    i=$((i+1))
    if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
    env $envs python bench.py $FLAGS > gpurun_out/$TAG/s${i}_${round}.json 2>/dev/null
    python - <<PY
import json
j = json.load(open('gpurun_out/$TAG/s${i}_${round}.json')); k = j['kernels_ms']
st = j.get('support_tiles') or {}
print('%-34s %d  %.4f ms  inner %.4f  fwd %.4f bwd %.4f adj %.4f disc %.4f small %.4f | support tiles %s' % ('$setting', $round, j['ms_per_step'], j['roofline']['whole_step']['inner_only_ms_per_step'], k['k_lbs_fwd'], k['k_lbs_bwd'], k['k_gemm_tn_blend_adjoint'], k.get('pose_disc_gemms', 0), k['k_prep_fwd'] + k['k_joints_loss'] + k['k_prep_bwd'], st.get('ms_per_step')))
PY
  done
done
