#!/bin/bash
# round 6: k_tail_step (slab sum + ONE launch for MLP adjoint, chain adjoint + Adam, the next iteration's chain forward + MLP forward) against the
# stand-alone launches, per batch size (GPU box, repo root): BASELINE configs[1] (joint loss only) and configs[2] (+ pose discriminator), all tiles
for cfg in 2 3; do for b in 256 512 1024 2048; do for t in 0 1; do
JRR_TAIL_STEP=$t python bench.py --config $cfg --batch $b --steps 20 --warmup 5 --no_driver_blocks --no_cpu_baseline --no_folded --no_skin_variants --no_rccl_one_rank --no_bf16x3 --no_support_tiles --no_config2 --no_config5 --min_timed_ms 300 2>/dev/null | python -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=p['kernels_ms']
print('config %d B=%4d tail=$t  %.4f ms  prep_fwd %.4f loss %.4f prep_bwd %.4f' % ($cfg, $b, p['ms_per_step'], k['k_prep_fwd'], k['k_joints_loss'], k['k_prep_bwd']))"
done; done; done
