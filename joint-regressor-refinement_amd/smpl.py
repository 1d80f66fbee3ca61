"""SMPL operator with the reference's calling convention, evaluated by the HIP kernels.

Mirrors the wrapper the reference constructs as `SMPL("SPIN/data/smpl", batch_size=1).to(device)`
(/root/reference/scripts/optimize.py:96-99; scripts/smpl.py:61-85) and calls as
    smpl(global_orient=(B,1,3,3), body_pose=(B,23,3,3), betas=(B,10), pose2rot=False).vertices
(scripts/utils.py:94-95, scripts/optimize.py:78-79, scripts/renderer.py:32-33).  Only `.vertices`
is consumed on this path.  `.vertices` is differentiable w.r.t. all three inputs (analytic adjoint kernels).
`.joints` (dead on the path: no caller of the reference reads it) holds the 24 posed joints of the kinematic chain
(jrr_smpl_posed_joints) and -- when the wrapper's `J_regressor_extra.npy` is found -- the reference's 49 re-mapped joints
(scripts/smpl.py:61-84); like smplx's it is differentiable w.r.t. all three inputs: the posed joints through the chain adjoint
(jrr_smpl_posed_joints_backward), the vertex-derived joints through `.vertices`.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import engine as _engine
from . import smpl_model as _smpl_model


# scripts/smpl.py:12-51 as data: [JOINT_MAP[name] for name in JOINT_NAMES] -- indices into cat(24 posed joints, smplx's 21 vertex
# joints, 9 extra regressed joints) for the 49 joints the wrapper returns
JOINT_MAP_49 = (24, 12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4, 7, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 8, 5, 45, 46, 4, 7, 21, 19, 17, 16,
                18, 20, 47, 48, 49, 50, 51, 52, 53, 24, 26, 25, 28, 27)
# smplx 0.1.26 VertexJointSelector for SMPL (vertex_ids['smplh']; [3P-memory], like SURVEY.md Appendix A): nose, right / left eye,
# right / left ear; left big toe, small toe, heel, right big toe, small toe, heel; left thumb .. pinky tips, right thumb .. pinky tips
SMPL_VERTEX_JOINTS = (332, 6260, 2800, 4071, 583, 3216, 3226, 3387, 6617, 6624, 6787, 2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905,
                      6016, 6133)
JOINT_REGRESSOR_TRAIN_EXTRA = 'data/vibe_data/J_regressor_extra.npy'      # scripts/smpl.py:54


class SMPLOutput:
    def __init__(self, vertices, global_orient=None, body_pose=None, betas=None, joints=None):
        self.vertices = vertices
        self.global_orient = global_orient
        self.body_pose = body_pose
        self.betas = betas
        self.joints = joints


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


class _PosedJointsFn(torch.autograd.Function):
    """the 24 posed joints of the forward `_SMPLVerticesFn` has just run on `eng` (same R, betas)"""

    @staticmethod
    def forward(ctx, R, betas, eng):
        R, betas = R.detach().contiguous(), betas.detach().contiguous()
        joints = eng.posed_joints(betas)
        ctx.eng, ctx.gen = eng, eng.generation
        ctx.save_for_backward(R, betas)
        return joints

    @staticmethod
    def backward(ctx, dj):
        R, betas = ctx.saved_tensors
        eng = ctx.eng
        if eng.generation != ctx.gen:          # engine state overwritten by a later forward: re-run this one
            eng.find_joints_forward(betas, R=R)
            ctx.gen = eng.generation
        dR, db = eng.posed_joints_backward(betas, dj.contiguous(), R=R)
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
                 allow_synthetic: bool = True, joint_regressor_extra: Optional[str] = None, **_):
        self.model_np = model if model is not None else _smpl_model.load_smpl_model(model_path, allow_synthetic)
        # scripts/smpl.py:66-69: the (9,6890) extra joint regressor; without the file `.joints` is the 24 posed joints only
        self.J_regressor_extra = None
        import os
        for cand in (joint_regressor_extra, JOINT_REGRESSOR_TRAIN_EXTRA,
                     os.path.join(model_path, 'J_regressor_extra.npy') if model_path else None):
            if cand and os.path.isfile(cand):
                import numpy as np
                self.J_regressor_extra = torch.from_numpy(np.load(cand).astype('float32'))
                break
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
        joints = _PosedJointsFn.apply(R, betas.float(), eng)                 # the chain's 24 posed joints of THIS forward
        if self.J_regressor_extra is not None:                               # scripts/smpl.py:75-78
            extra = torch.einsum('jv,bvc->bjc', self.J_regressor_extra.to(verts.device), verts)
            joints = torch.cat([joints, verts[:, list(SMPL_VERTEX_JOINTS)], extra], dim=1)[:, list(JOINT_MAP_49)]
        return SMPLOutput(verts, global_orient, body_pose, betas, joints)
