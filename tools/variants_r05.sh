#!/bin/bash
# the GPU suite of the final round-5 build under each forced kernel variant (one box)
mkdir -p gpurun_out/r5v
{
  echo "GPU suite (pytest -m gpu) of the final round-5 build under each forced kernel variant, one MI355X box, $(date +%F):"
  for v in JRR_SKIN_JOINTS=12 JRR_DENSE_SKINNING=1 JRR_BWD16=0 JRR_VERTEX_ORDER=sorted JRR_BWD16_NG=2 JRR_DISC_KS=1; do
    echo "== $v"
    env $v python -m pytest tests -m gpu -q 2>&1 | tail -1
  done
} > gpurun_out/r5v/gpu_suite_variants.txt 2>&1
cat gpurun_out/r5v/gpu_suite_variants.txt
