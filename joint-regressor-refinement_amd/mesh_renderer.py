"""Soft-silhouette renderer with the reference's module interface
(/root/reference/scripts/mesh_renderer.py:23-79, used by render_mesh at scripts/optimize.py:77-85) -- SURVEY.md
section 8 row f2.  pytorch3d 0.3.0 (MeshRasterizer 224^2, blur_radius 0, faces_per_pixel 1 + SoftSilhouetteShader,
sigma 1e-4) is restated by the HIP kernels in csrc/sil.hip (parity unpinned: pytorch3d is absent)."""
from __future__ import annotations

import torch
from torch import nn

from . import engine as _engine


class _SilhouetteFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, cam, eng):
        verts, cam = verts.detach().contiguous().float(), cam.detach().contiguous().float()
        alpha = eng.silhouette_forward(verts, cam)
        ctx.eng, ctx.gen = eng, eng.generation
        ctx.save_for_backward(verts, cam)
        return alpha

    @staticmethod
    def backward(ctx, galpha):
        eng = ctx.eng
        if eng.generation != ctx.gen:      # the rasteriser's per-pose face lists belong to a later forward: redo this one
            verts, cam = ctx.saved_tensors
            eng.silhouette_forward(verts, cam)
            ctx.gen = eng.generation
        dverts, dcam = eng.silhouette_backward(galpha.contiguous())
        return dverts, dcam, None


class Mesh_Renderer(nn.Module):
    """Mesh_Renderer(image_size)(batch, smpl_verts) -> (B, 4, H, W); channel 3 is the soft silhouette.

    As in the reference (scripts/mesh_renderer.py:48-79), `smpl_verts` are the vertices AFTER render_mesh's
    x/y flip and x2 scale (scripts/optimize.py:80-82): a call site ported from the reference keeps its own flip and
    scale.  The HIP projection kernel applies that transform itself, so forward() maps the argument back to SMPL
    space first (an exact sign flip and halving).  Channels 0-2 are 1 as pytorch3d's SoftSilhouetteShader returns
    them (sigmoid_alpha_blend of all-ones colours), independent of the textures."""

    def __init__(self, image_size: int = 256, smpl=None):
        # the reference constructor's signature and default (scripts/mesh_renderer.py:25); its loop instantiates 224
        # (scripts/optimize.py:110).  The stand-alone HIP rasteriser takes every multiple of 32 up to 256 (include/jrr.h,
        # JRR_FLAG_SIL_SIZE; the image size is a template parameter of its strip arithmetic); the camera's focal length is
        # 5000 / image_size as in the reference (mesh_renderer.py:52-53).
        super().__init__()
        if image_size % 32 or not 32 <= image_size <= 256:
            raise NotImplementedError(f'Mesh_Renderer(image_size={image_size}): the HIP rasteriser takes the multiples of 32 up to 256 '
                                      '(include/jrr.h, JRR_FLAG_SIL_SIZE)')
        self.image_size = image_size
        self.smpl = smpl
        self._engines = {}

    def _engine(self, batch):
        if batch not in self._engines:
            flags = _engine.FLAG_SILHOUETTE | (0 if self.image_size == 224 else _engine.FLAG_SIL_SIZE(self.image_size))
            self._engines[batch] = _engine.RefineEngine(self.smpl.device_model, batch, flags=flags)
        return self._engines[batch]

    def forward(self, batch, smpl_verts):
        world = smpl_verts * smpl_verts.new_tensor([-0.5, -0.5, 0.5])
        alpha = _SilhouetteFn.apply(world, batch['cam'], self._engine(smpl_verts.shape[0]))
        ones = torch.ones_like(alpha)
        return torch.stack([ones, ones, ones, alpha], dim=1)


def render_mesh(smpl, silhouette_renderer, betas, orient, pose, batch):
    """scripts/optimize.py:77-85: SMPL vertices, x/y flip, x2 scale, alpha channel of the renderer."""
    pred_vertices = smpl(global_orient=orient, body_pose=pose, betas=betas, pose2rot=False).vertices
    pred_vertices = pred_vertices * pred_vertices.new_tensor([-2.0, -2.0, 2.0])
    return silhouette_renderer(batch, pred_vertices)[:, 3].unsqueeze(1)
