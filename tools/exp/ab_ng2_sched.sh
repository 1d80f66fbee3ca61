mkdir -p gpurun_out/r5k
F="--no_cpu_baseline --no_folded --no_config5 --no_skin_variants --no_config2 --no_rccl_one_rank --no_support_tiles --no_driver_blocks --min_timed_ms 1000 --steps 20 --warmup 5"
run() { tag=$1; shift; env "$@" python bench.py $F > gpurun_out/r5k/$tag.json 2>/dev/null; python - <<PY
import json
j=json.load(open('gpurun_out/r5k/$tag.json')); k=j['kernels_ms']
print('%-12s %.4f ms  bwd %.4f  fwd %.4f adj %.4f disc %.4f' % ('$tag', j['ms_per_step'], k['k_lbs_bwd'], k['k_lbs_fwd'], k['k_gemm_tn_blend_adjoint'], k['pose_disc_gemms']))
PY
}
for r in 1 2; do
run ng1_$r X=1
run ng2_$r JRR_BWD16_NG=2
for v in ng2s13 ng2s14 ng2s26; do run ${v}_$r JRR_BWD16_NG=2 JRR_LIB=$PWD/tools/probe/libjrr_$v.so; done
done
bash tools/exp/ab_libs.sh r5k "--config 5 --steps 20 --warmup 5" silpf
