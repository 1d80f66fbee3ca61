#!/bin/bash
# The end-of-round evidence of ONE box, in one place (replaces the per-round final_r05*.sh scripts):
#
#     bash tools/final.sh <tag> [variants]        (GPU box, repo root; e.g. gpurun -- 'bash tools/final.sh r06')
#
# writes gpurun_out/<tag>/: the kernel trace + PMC passes of the headline command (tools/prof.sh), the rasteriser's counters
# (tools/prof_c5.sh), the two bench lines the driver produces (--steps 20 --warmup 5, and the defaults), BASELINE configs[1] and [4] as
# their own lines, the GPU suite and smoke().  `variants` runs INSTEAD the GPU suite under each forced kernel-selection variant
# (13 x ~6.5 minutes: more than one gpurun call of 60 minutes -- VARIANTS="A=1 B=2" selects a subset).
# Copy what DESIGN.md / profiles/README.md cite into profiles/ afterwards -- gpurun_out/ is scratch.
TAG=${1:?usage: bash tools/final.sh <tag> [variants]}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "$2" = variants ]; then
  {
    echo "GPU suite (pytest -m gpu) under each forced kernel variant, one MI355X box, $(date +%F):"
    for v in ${VARIANTS:-JRR_SKIN_JOINTS=12 JRR_DENSE_SKINNING=1 JRR_BWD16=0 JRR_VERTEX_ORDER=sorted JRR_BWD16_NG=2 JRR_DISC_KS=1 JRR_DISC_NARROW=0 JRR_DISC_NARROW=1 \
             JRR_BWD16_REV=1 JRR_SUPPORT_FUSED=0 JRR_SUPPORT_FUSED=2 JRR_SUP_OVERLAP=0 JRR_SUP_OVERLAP=1}; do
      echo "== $v"
      env $v python -m pytest tests -m gpu -q 2>&1 | tail -1
    done
  } > $OUT/gpu_suite_variants.txt 2>&1
  cat $OUT/gpu_suite_variants.txt
  exit 0
fi
bash tools/prof.sh $TAG > $OUT/prof.log 2>&1
bash tools/prof_c5.sh ${TAG}c5 > $OUT/prof_c5.log 2>&1
python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; echo "rc $?" >> $OUT/bench_steps20.err
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc $?" >> $OUT/bench_default.err
python bench.py --config 2 --batch 1024 --steps 20 --warmup 5 --no_driver_blocks > $OUT/bench_config2.json 2>/dev/null
python bench.py --config 5 --steps 20 --warmup 5 --no_driver_blocks > $OUT/bench_config5.json 2>/dev/null
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $OUT/gpu_suite.txt 2>&1
tail -3 $OUT/prof.log; tail -2 $OUT/prof_c5.log | cut -c1-400; cat $OUT/gpu_suite.txt; tail -1 $OUT/bench_default.err
