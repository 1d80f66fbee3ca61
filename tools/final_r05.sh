#!/bin/bash
# round-5 final evidence on ONE box (repo root): kernel trace + PMC passes, the rasteriser's counters, the four bench lines, the GPU suite
# under each forced kernel variant
mkdir -p gpurun_out/r5z
bash tools/prof.sh r05 > gpurun_out/r5z/prof.log 2>&1
bash tools/prof_c5.sh r05c5 > gpurun_out/r5z/prof_c5.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r5z/bench_steps20.json 2> gpurun_out/r5z/bench_steps20.err; echo "rc $?" >> gpurun_out/r5z/bench_steps20.err
python bench.py > gpurun_out/r5z/bench_default.json 2> gpurun_out/r5z/bench_default.err; echo "rc $?" >> gpurun_out/r5z/bench_default.err
python bench.py --config 2 --batch 1024 --steps 20 --warmup 5 --no_driver_blocks > gpurun_out/r5z/bench_config2.json 2>/dev/null
python bench.py --config 5 --steps 20 --warmup 5 --no_driver_blocks > gpurun_out/r5z/bench_config5.json 2>/dev/null
{
  echo "GPU suite (pytest -m gpu) of the final round-5 build under each forced kernel variant, one MI355X box, $(date +%F):"
  for v in JRR_SKIN_JOINTS=12 JRR_DENSE_SKINNING=1 JRR_BWD16=0 JRR_VERTEX_ORDER=sorted JRR_BWD16_NG=2 JRR_DISC_KS=1; do
    echo "== $v"
    env $v python -m pytest tests -m gpu -q 2>&1 | tail -1
  done
} > gpurun_out/r5z/gpu_suite_variants.txt 2>&1
tail -3 gpurun_out/r5z/prof.log; tail -2 gpurun_out/r5z/prof_c5.log | cut -c1-300; cat gpurun_out/r5z/gpu_suite_variants.txt
