"""Joint-regression operators of the reference (/root/reference/scripts/utils.py) on the HIP path.

  find_joints       scripts/utils.py:85-103   fused SMPL + J_regressor contraction, differentiable
                                               w.r.t. shape, orient, pose and J_regressor
  move_pelvis       scripts/utils.py:106-114
  evaluate          scripts/utils.py:117-145  (k_evaluate: on-device Procrustes, row f3 of SURVEY.md)
  find_j_reg_mask   scripts/utils.py:182-187  (reproduces the reference's all-ones mask)
  rot6d_to_rotmat   scripts/utils.py:190-204  (row-wise cross product for every N)
  set_seed          scripts/utils.py:207-215
"""
from __future__ import annotations

import random

import numpy as np
import torch

from . import engine as _engine


class _Rot6dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.detach().contiguous().float()
        ctx.save_for_backward(x)
        return _engine.rot6d_forward(x)

    @staticmethod
    def backward(ctx, dR):
        (x,) = ctx.saved_tensors
        return _engine.rot6d_backward(x, dR.contiguous()).view_as(x)


def rot6d_to_rotmat(x: torch.Tensor) -> torch.Tensor:
    """(N*6,) / (N,6) / (...,6) -> (N,3,3); columns of R are b1, b2, b3."""
    return _Rot6dFn.apply(x.reshape(-1, 6))


class _FindJointsFn(torch.autograd.Function):
    """The adjoint kernels read the engine's state of the most recent forward; autograd defers backward, so the
    forward generation is saved and, if another call has used the engine since, the forward is re-run from the
    saved inputs (and the saved J / mask) before the adjoint."""

    @staticmethod
    def forward(ctx, R, betas, J, mask, eng):
        R, betas = R.detach().contiguous(), betas.detach().contiguous()
        Jd = J.detach().clone()
        md = None if mask is None else mask.detach().clone()
        eng.set_j_regressor(Jd, md)
        joints = eng.find_joints_forward(betas, R=R)
        ctx.eng, ctx.gen = eng, eng.generation
        ctx.need_dJ = J.requires_grad
        ctx.has_mask = md is not None
        ctx.save_for_backward(R, betas, Jd, *([md] if md is not None else []))
        return joints

    @staticmethod
    def backward(ctx, dj):
        R, betas, Jd = ctx.saved_tensors[:3]
        md = ctx.saved_tensors[3] if ctx.has_mask else None
        eng = ctx.eng
        if eng.generation != ctx.gen:          # the engine has been used since: restore this call's forward state
            eng.set_j_regressor(Jd, md)
            eng.find_joints_forward(betas, R=R)
            ctx.gen = eng.generation
        dR, db, dJ = eng.find_joints_backward(betas, dj.contiguous(), R=R, want_dJ=ctx.need_dJ)
        return dR, db, dJ, None, None


def find_joints(smpl, shape, orient, pose, J_regressor, mask=None, return_verts=False):
    """J*mask -> ReLU -> row-normalise -> SMPL vertices -> batched J.V product.  `smpl` is a
    joint-regressor-refinement_amd.smpl.SMPL; orient (B,1,3,3), pose (B,23,3,3), shape (B,10)."""
    if return_verts:
        # composition exactly as written in the reference (utils.py:87-101), on the HIP SMPL operator
        Jm = J_regressor * mask if mask is not None else J_regressor
        Jn = torch.relu(Jm)
        Jn = Jn / torch.sum(Jn, dim=1).unsqueeze(1).expand(Jn.shape)
        verts = smpl(global_orient=orient, body_pose=pose, betas=shape, pose2rot=False).vertices
        return torch.matmul(Jn[None].expand(verts.shape[0], -1, -1), verts), verts
    B = shape.shape[0]
    R = torch.cat([orient.reshape(B, 1, 3, 3), pose.reshape(B, 23, 3, 3)], dim=1).float()
    eng = smpl.engine(B)      # one engine per batch size, shared with smpl(...): the generation counter keeps deferred
    #                           backward passes correct when several forwards interleave
    return _FindJointsFn.apply(R, shape.float(), J_regressor, mask, eng)


def move_pelvis(j3ds: torch.Tensor) -> torch.Tensor:
    return j3ds - j3ds[:, [0], :]


def find_j_reg_mask(j_reg: torch.Tensor) -> torch.Tensor:
    """The reference builds both branches of torch.where from torch.ones (utils.py:183-184), so its
    mask is identically 1; reproduced for parity (sparsity is preserved by ReLU'(0) = 0 anyway)."""
    ones = torch.ones_like(j_reg)
    return torch.where(j_reg == 0, ones, ones)


def evaluate(pred_j3ds: torch.Tensor, target_j3ds: torch.Tensor):
    """MPJPE and PA-MPJPE in mm (pred in m, target in mm), scripts/utils.py:117-145, on device
    (k_evaluate: pelvis centring, per-pose Procrustes with an in-register 3x3 SVD)."""
    with torch.no_grad():
        err, err_pa = _engine.evaluate(pred_j3ds.detach().float(), target_j3ds.detach().float())
        return float(err.mean().item()) * 1000, float(err_pa.mean().item()) * 1000


def evaluate_sums(pred_j3ds: torch.Tensor, target_j3ds: torch.Tensor):
    """Sums over the batch of the per-pose joint error and PA joint error in METRES (device tensors): the pieces a
    data-parallel run all-reduces before dividing by the global batch."""
    with torch.no_grad():
        err, err_pa = _engine.evaluate(pred_j3ds.detach().float(), target_j3ds.detach().float())
        return err.sum(), err_pa.sum()


def set_seed(seed: int):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)
