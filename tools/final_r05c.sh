#!/bin/bash
# the two bench lines the driver produces + the GPU suite + smoke on ONE box, final round-5 build (after the split-bf16 side mode was added)
mkdir -p gpurun_out/r5x
python bench.py --steps 20 --warmup 5 > gpurun_out/r5x/bench_steps20.json 2> gpurun_out/r5x/bench_steps20.err; echo "rc $?" >> gpurun_out/r5x/bench_steps20.err
python bench.py > gpurun_out/r5x/bench_default.json 2> gpurun_out/r5x/bench_default.err; echo "rc $?" >> gpurun_out/r5x/bench_default.err
python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r5x/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r5x/gpu_suite.txt 2>&1
cat gpurun_out/r5x/gpu_suite.txt; tail -1 gpurun_out/r5x/bench_default.err
