#!/bin/bash
# round-5 final evidence after the late rasteriser work, ONE box (repo root): kernel trace + PMC passes, the rasteriser's counters, the
# four bench lines, the default GPU suite (the forced-variant suites of tools/final_r05.sh do not touch the rasteriser: not repeated)
mkdir -p gpurun_out/r5y
bash tools/prof.sh r05 > gpurun_out/r5y/prof.log 2>&1
bash tools/prof_c5.sh r05c5 > gpurun_out/r5y/prof_c5.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r5y/bench_steps20.json 2> gpurun_out/r5y/bench_steps20.err; echo "rc $?" >> gpurun_out/r5y/bench_steps20.err
python bench.py > gpurun_out/r5y/bench_default.json 2> gpurun_out/r5y/bench_default.err; echo "rc $?" >> gpurun_out/r5y/bench_default.err
python bench.py --config 2 --batch 1024 --steps 20 --warmup 5 --no_driver_blocks > gpurun_out/r5y/bench_config2.json 2>/dev/null
python bench.py --config 5 --steps 20 --warmup 5 --no_driver_blocks > gpurun_out/r5y/bench_config5.json 2>/dev/null
python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r5y/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r5y/gpu_suite.txt 2>&1
tail -3 gpurun_out/r5y/prof.log; tail -2 gpurun_out/r5y/prof_c5.log | cut -c1-400; cat gpurun_out/r5y/gpu_suite.txt
