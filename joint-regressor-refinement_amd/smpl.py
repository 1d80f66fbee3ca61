"""SMPL operator with the reference's calling convention, evaluated by the HIP kernels.

Mirrors the wrapper the reference constructs as `SMPL("SPIN/data/smpl", batch_size=1).to(device)`
(/root/reference/scripts/optimize.py:96-99; scripts/smpl.py:61-85) and calls as
    smpl(global_orient=(B,1,3,3), body_pose=(B,23,3,3), betas=(B,10), pose2rot=False).vertices
(scripts/utils.py:94-95, scripts/optimize.py:78-79, scripts/renderer.py:32-33).  Only `.vertices`
is consumed on this path; the wrapper's 49 extra regressed joints are dead code there and are not
produced.  `.vertices` is differentiable w.r.t. all three inputs (analytic adjoint kernels).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import engine as _engine
from . import smpl_model as _smpl_model


class SMPLOutput:
    def __init__(self, vertices, global_orient=None, body_pose=None, betas=None):
        self.vertices = vertices
        self.global_orient = global_orient
        self.body_pose = body_pose
        self.betas = betas
        self.joints = None


class _SMPLVerticesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, R, betas, eng):
        R, betas = R.detach().contiguous(), betas.detach().contiguous()
        _, verts = eng.find_joints_forward(betas, R=R, return_verts=True)
        ctx.eng, ctx.gen = eng, eng.generation
        ctx.save_for_backward(R, betas)
        return verts

    @staticmethod
    def backward(ctx, dverts):
        R, betas = ctx.saved_tensors
        eng = ctx.eng
        if eng.generation != ctx.gen:          # engine state overwritten by a later forward: re-run this one
            eng.find_joints_forward(betas, R=R)
            ctx.gen = eng.generation
        dR, db = eng.smpl_vertices_backward(betas, dverts.contiguous(), R=R)
        return dR, db, None


class _RodriguesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, aa):
        aa = aa.detach().contiguous().float()
        ctx.save_for_backward(aa)
        return _engine.rodrigues_forward(aa)

    @staticmethod
    def backward(ctx, dR):
        (aa,) = ctx.saved_tensors
        return _engine.rodrigues_backward(aa, dR.contiguous()).view_as(aa)


def batch_rodrigues(rot_vecs: torch.Tensor) -> torch.Tensor:
    """smplx lbs.batch_rodrigues on the HIP kernel: (N,3) axis-angle -> (N,3,3), differentiable."""
    return _RodriguesFn.apply(rot_vecs.reshape(-1, 3))


class SMPL:
    """smpl = SMPL(model_dir, batch_size=1).to(device)"""

    def __init__(self, model_path: Optional[str] = None, batch_size: int = 1, model: Optional[Dict] = None,
                 allow_synthetic: bool = True, **_):
        self.model_np = model if model is not None else _smpl_model.load_smpl_model(model_path, allow_synthetic)
        self.provenance = str(self.model_np.get('provenance', 'caller-supplied arrays'))
        self.faces = self.model_np.get('faces')
        self.device = None
        self.device_model = None
        self._engines: Dict = {}

    def to(self, device, hint_vertices=None):
        """hint_vertices (optional, this build's): the vertices an H36M regressor reads (its positive columns) -- a hint for the
        library's internal vertex order (engine.DeviceModel); results and every index at the API are unaffected"""
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError('the HIP SMPL operator has no CPU path (device must be a ROCm "cuda" device)')
        if self.device_model is None or self.device != device or hint_vertices is not None:
            self.device = device
            self.device_model = _engine.DeviceModel(self.model_np, device, hint_vertices=hint_vertices)
            self._engines = {}
        return self

    def engine(self, batch: int, flags: int = _engine.FLAG_KEEP_VERTS) -> '_engine.RefineEngine':
        if self.device_model is None:
            self.to('cuda:0')
        key = (batch, flags)
        if key not in self._engines:
            eng = _engine.RefineEngine(self.device_model, batch, flags=flags)
            eng.set_j_regressor(torch.ones(_engine.NUM_H36M, _engine.NUM_VERTS, device=self.device))
            self._engines[key] = eng
        return self._engines[key]

    def release_engines(self):
        """drop the cached per-batch-size engines (each owns a workspace of ~0.4 MB per pose with vertex buffers)"""
        self._engines = {}

    def __call__(self, global_orient=None, body_pose=None, betas=None, pose2rot=True, **_):
        """smplx.SMPL.forward's signature and default: pose2rot=True takes axis-angle global_orient (B,3) and
        body_pose (B,69) and converts them with batch_rodrigues (HIP kernel); the reference's hot path passes
        rotation matrices with pose2rot=False (scripts/utils.py:94-95)."""
        B = betas.shape[0]
        if pose2rot:
            aa = torch.cat([global_orient.reshape(B, 1, 3), body_pose.reshape(B, 23, 3)], dim=1).float()
            R = batch_rodrigues(aa.reshape(-1, 3)).view(B, 24, 3, 3)
        else:
            R = torch.cat([global_orient.reshape(B, 1, 3, 3), body_pose.reshape(B, 23, 3, 3)], dim=1).float()
        eng = self.engine(B)
        verts = _SMPLVerticesFn.apply(R, betas.float(), eng)
        return SMPLOutput(verts, global_orient, body_pose, betas)
