"""Data parallelism for the pose-refinement path (new in this build).

Poses shard naturally: every pose owns its parameters, Adam state and loss terms; the inner loop
needs NO collective (the MSE normaliser uses the GLOBAL batch, a constant).  The only exchange is
one sum-all-reduce of the shared-parameter gradients at each shared-parameter step: the
J_regressor gradient (17 x 6890 fp32 = 468 520 B) and, at the outer step, the discriminator
gradients.  One process per GPU; backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" for CPU tests.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def init(backend: Optional[str] = None):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1, unless JRR_DIST_SINGLE_RANK=1
    asks for the collectives' call sites to run with a one-rank group: the way a 1-GPU box executes them over RCCL)."""
    import torch.distributed as dist
    rank, local_rank, world = env_rank_world()
    if world == 1 and os.environ.get('JRR_DIST_SINGLE_RANK', '0') != '1':
        return None
    if not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl':     # RCCL: one rank per GPU
            torch.cuda.set_device(local_rank)
            kw['device_id'] = torch.device('cuda', local_rank)
        dist.init_process_group(backend, **kw)
    return dist


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous batch shard [lo, hi) of rank `rank` (sizes differ by at most one)."""
    return n * rank // world, n * (rank + 1) // world


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():      # (a one-rank group too: JRR_DIST_SINGLE_RANK)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def agree_max(value: int, device=None) -> int:
    """MAX of a small integer over the ranks (the identity without a process group): how the ranks agree on a per-batch
    outcome -- e.g. "this batch failed to load on some rank" -- before they enter the next collective together."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(value)
    on_gpu = dist.get_backend() == 'nccl'
    t = torch.tensor([int(value)], dtype=torch.int32, device=device if on_gpu else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


class JStepExchange:
    """The J step between two segments of the inner loop under data parallelism (scripts/optimize.py:300-312 sharded):
    local gradient -> ONE sum-all-reduce -> replicated Adam + re-normalisation, no allocation, nothing read back.

    The gradient w.r.t. the raw regressor is exactly zero outside the regressor's positive support (ReLU'), and every rank
    holds the same regressor: when the support fits the engine's device-side lists (asked ONCE, at construction: J steps
    only shrink it) the ranks exchange the (17,128) support values -- 8 704 bytes instead of 468 520 -- through
    jrr_j_regressor_grad_support / jrr_j_step_apply_support; otherwise the dense (17,6890) pair.  `reduce` is the
    collective (default: the all-reduce of this module; bench.py passes a no-op to time the host-driven sequence at
    world size 1)."""

    def __init__(self, eng, dense_buf: torch.Tensor, compact: bool = True, reduce=None):
        self.eng = eng
        self.dense = dense_buf
        self.reduce = reduce if reduce is not None else all_reduce_sum_
        self.compact = False
        if compact:
            _, fits = eng.j_support_info()
            self.compact = bool(fits)
        self.buf = torch.zeros(17, eng.J_SUPPORT_CAP, device=dense_buf.device) if self.compact else None
        self.nbytes = (self.buf if self.compact else self.dense).numel() * 4

    def step(self, J, opt_m, opt_v, opt_step, lr, x6d, betas, gt, mask=None) -> None:
        if self.compact:
            self.eng.j_regressor_grad_support(x6d, betas, gt, out=self.buf)
            self.reduce(self.buf)
            self.eng.j_step_apply_support(J, self.buf, opt_m, opt_v, opt_step, lr, mask=mask)
        else:
            self.eng.j_regressor_grad(x6d, betas, gt, out=self.dense)
            self.reduce(self.dense)
            self.eng.j_step_apply(J, self.dense, opt_m, opt_v, opt_step, lr, mask=mask)
