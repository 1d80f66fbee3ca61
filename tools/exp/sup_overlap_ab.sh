#!/bin/bash
# usage (GPU box, repo root): bash tools/exp/sup_overlap_ab.sh   -- JRR_SUP_OVERLAP=1 against the composed kernel, same box: time and bits
mkdir -p gpurun_out/supov
for b in 256 512 1024 4096; do
  timeout 150 python tools/exp/sup_overlap_check.py $b 100 gpurun_out/supov/base_$b.npy 2>/dev/null | tail -1
  JRR_SUP_OVERLAP=1 timeout 150 python tools/exp/sup_overlap_check.py $b 100 gpurun_out/supov/ov_$b.npy 2>/dev/null | tail -1
  python -c "
import numpy as np
a, b = np.load('gpurun_out/supov/base_$b.npy'), np.load('gpurun_out/supov/ov_$b.npy')
print('  bits equal:', np.array_equal(a, b), ' max abs diff', float(np.abs(a - b).max()), ' step', a[-1], b[-1])"
  rm -f gpurun_out/supov/*.npy
done
