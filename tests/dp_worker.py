"""Rank process of the data-parallel product-path test (tests/test_gpu_dp.py).

Runs the real driver `optimize_pose_refiner()` of the package on the HIP engine -- under
`python -m torch.distributed.run` for world size 2 (both ranks on cuda:0 over gloo: `--single_device
--dist_backend gloo`, the 1-GPU stand-in for one rank per GPU over RCCL) or directly for world size 1 --
and saves what the test compares: refined poses of the rank's shard, the shared parameters after the
outer step and the log record.

    python tests/dp_worker.py OUT_PREFIX [driver flags ...]
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'joint-regressor-refinement_amd'


def main():
    out, flags = sys.argv[1], sys.argv[2:]
    argsmod = importlib.import_module(PKG + '.args')
    argsmod._LazyArgs._ns = argsmod.get_args(flags)
    opt = importlib.import_module(PKG + '.optimize')
    records = []
    res = opt.optimize_pose_refiner(log=records.append)
    rank = int(os.environ.get('RANK', '0'))
    lo, hi = res['shard']
    np.savez(f'{out}.rank{rank}.npz', J=res['J_regressor'].cpu().numpy(), disc=res['disc_flat'].cpu().numpy(),
             sdisc=res['sdisc_flat'].cpu().numpy(), x6d=res['x6d'].cpu().numpy(), betas=res['betas'].cpu().numpy(), cam=res['cam'].cpu().numpy(),
             lo=lo, hi=hi, history=json.dumps(res['history']))
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
