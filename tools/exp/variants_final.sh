mkdir -p gpurun_out/r06w
{
  echo "GPU suite (pytest -m gpu) of the FINAL build (lib c90048adf51e12ca) under the kernel-selection variants that interact with the launch geometry, one MI355X box, $(date +%F):"
  for v in JRR_SKIN_JOINTS=12 JRR_DENSE_SKINNING=1 JRR_BWD16=0 JRR_VERTEX_ORDER=sorted JRR_SUPPORT_FUSED=0; do
    echo "== $v"
    env $v python -m pytest tests -m gpu -q 2>&1 | tail -1
  done
} > gpurun_out/r06w/gpu_suite_variants_final.txt 2>&1
cat gpurun_out/r06w/gpu_suite_variants_final.txt
