#!/bin/bash
# round 6: the forward kernel's vertex-chunk cap at small batches (api.hip plan_geometry, JRR_FWD_CHUNK_CAP): BASELINE configs[4] at 256 / 512 poses and
# the headline mode at 512 poses, per cap (GPU box, repo root)
for b in 256 512; do for cap in 54 108 216; do
JRR_FWD_CHUNK_CAP=$cap python bench.py --config 5 --batch $b --steps 20 --warmup 5 --no_driver_blocks --no_cpu_baseline --no_folded --no_skin_variants --no_rccl_one_rank --no_bf16x3 --no_support_tiles --no_config2 --min_timed_ms 300 2>/dev/null | python -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=p['kernels_ms']
print('config5 B=$b cap=$cap  %.4f ms  fwd %.4f loss %.4f bwd %.4f adj %.4f' % (p['ms_per_step'], k['k_lbs_fwd'], k['k_joints_loss'], k['k_lbs_bwd'], k['k_gemm_tn_blend_adjoint']))"
done; done
for cap in 54 108; do JRR_FWD_CHUNK_CAP=$cap python bench.py --config 3 --batch 512 --steps 20 --warmup 5 --no_driver_blocks --no_cpu_baseline --no_folded --no_skin_variants --no_rccl_one_rank --no_bf16x3 --no_support_tiles --no_config2 --no_config5 --min_timed_ms 300 2>/dev/null | python -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=p['kernels_ms']
print('config3 B=512 cap=$cap  %.4f ms  fwd %.4f loss %.4f' % (p['ms_per_step'], k['k_lbs_fwd'], k['k_joints_loss']))"
done
