"""Thin Python wrappers over the C ABI: device model, per-batch engine, free operator functions.

All tensors are torch CUDA (ROCm) float32 tensors; PyTorch owns every buffer including the
engine workspace, and kernels are enqueued on torch's current stream.
"""
from __future__ import annotations

import ctypes
from ctypes import byref, c_int32, c_void_p
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

FLAG_POSE_DISC = 1
FLAG_SHAPE_DISC = 2
FLAG_KEEP_VERTS = 4
FLAG_FOLDED = 8
FLAG_SILHOUETTE = 16
FLAG_NO_MODEL = 32
FLAG_SIL_256 = 64      # with FLAG_SILHOUETTE: 256 x 256 silhouettes (the reference constructor's default) instead of 224 x 224
def FLAG_SIL_SIZE(size: int) -> int:
    """with FLAG_SILHOUETTE: an explicit silhouette size, any multiple of 32 up to 256 (include/jrr.h JRR_FLAG_SIL_SIZE)"""
    if size % 32 or not 32 <= size <= 256:
        raise ValueError(f'silhouette size {size}: a multiple of 32 up to 256')
    return (size // 32) << 16


FLAG_SUPPORT_TILES = 128      # joint-loss iterations on the tiles of the regressor's support only (see include/jrr.h)
FLAG_BLEND_BF16X3 = 256       # SIDE MODE, not the reference's arithmetic: split-bf16 blend adjoint (see include/jrr.h); never the default
SIL = 224

NUM_VERTS, NUM_JOINTS, NUM_H36M, NUM_BETAS = 6890, 24, 17, 10
DISC_PARAMS = 1840153
SHAPE_DISC_PARAMS = 171

DISC_KEYS = (['conv_operations.0.weight', 'conv_operations.0.bias', 'conv_operations.2.weight', 'conv_operations.2.bias']
             + [k for i in range(24) for k in (f'linears.{i}.weight', f'linears.{i}.bias')]
             + ['linear_operations.0.weight', 'linear_operations.0.bias', 'linear_operations.2.weight',
                'linear_operations.2.bias', 'linear_operations.4.weight', 'linear_operations.4.bias'])
SHAPE_DISC_KEYS = ['shape_operations.0.weight', 'shape_operations.0.bias', 'shape_operations.2.weight',
                   'shape_operations.2.bias', 'shape_operations.4.weight', 'shape_operations.4.bias']


def flatten_state_dict(sd: Dict[str, torch.Tensor], keys) -> torch.Tensor:
    """state_dict -> the flat parameter vector the C ABI expects (state_dict order)."""
    return torch.cat([sd[k].detach().reshape(-1).float() for k in keys])


def unflatten_state_dict(flat: torch.Tensor, like: Dict[str, torch.Tensor], keys) -> Dict[str, torch.Tensor]:
    out, off = {}, 0
    for k in keys:
        n = like[k].numel()
        out[k] = flat[off:off + n].view_as(like[k])
        off += n
    return out


def _f32c(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


class DeviceModel:
    """SMPL constants uploaded in the kernels' tile-major layouts (jrr_model_create)."""

    def __init__(self, model: Dict[str, np.ndarray], device='cuda:0', hint_vertices=None):
        """hint_vertices (optional): file indices of the vertices the H36M regressor reads (its positive columns); the library stores
        them first internally, so that the iterations of FLAG_SUPPORT_TILES run ceil(n / 32) tiles (jrr_model_create_hinted)"""
        self.lib = _lib.load()
        self.device = torch.device(device)
        vt, sd, pd = _f32c(model['v_template']), _f32c(model['shapedirs']), _f32c(model['posedirs'])
        jr, w = _f32c(model['J_regressor']), _f32c(model['lbs_weights'])
        par = np.ascontiguousarray(np.asarray(model['parents'], dtype=np.int32))
        assert vt.shape == (NUM_VERTS, 3) and sd.shape == (NUM_VERTS, 3, NUM_BETAS) and pd.shape == (207, NUM_VERTS * 3)
        assert jr.shape == (NUM_JOINTS, NUM_VERTS) and w.shape == (NUM_VERTS, NUM_JOINTS) and par.shape == (NUM_JOINTS,)
        self.faces = model.get('faces')
        h = c_void_p()
        # the model's device memory is a torch buffer like every other buffer of the path (jrr_model_create_in)
        nbytes = int(self.lib.jrr_model_bytes())
        self.buffer = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.device)
        base = (self.buffer.data_ptr() + 255) // 256 * 256
        with torch.cuda.device(self.device):
            if hint_vertices is not None and len(hint_vertices):
                hv = np.ascontiguousarray(np.asarray(hint_vertices, dtype=np.int32).ravel())
                check(self.lib.jrr_model_create_hinted(vt.ctypes.data, sd.ctypes.data, pd.ctypes.data, jr.ctypes.data, w.ctypes.data,
                                                       par.ctypes.data, hv.ctypes.data, int(hv.size), c_void_p(base), nbytes, byref(h)),
                      'jrr_model_create_hinted')
            else:
                check(self.lib.jrr_model_create_in(vt.ctypes.data, sd.ctypes.data, pd.ctypes.data, jr.ctypes.data,
                                                   w.ctypes.data, par.ctypes.data, c_void_p(base), nbytes, byref(h)), 'jrr_model_create_in')
        self.handle = h
        info = (c_int32 * 30)()
        check(self.lib.jrr_model_info(self.handle, info, 30), 'jrr_model_info')
        # what the LBS kernels run for this body: joint slots per tile and pass (0: dense kernels), tiles that need a second
        # pass, the most joints of any tile, whether the library re-ordered the vertices internally, tiles by joint count
        self.info = {'joint_slots': int(info[0]), 'wide_tiles': int(info[1]), 'most_joints_per_tile': int(info[2]),
                     'internal_vertex_order': bool(info[3]), 'tile_joint_histogram': [int(x) for x in info[4:29]],
                     'hinted_vertices_stored_first': int(info[29])}
        if self.faces is not None:
            f = np.ascontiguousarray(np.asarray(self.faces, dtype=np.int32))
            check(self.lib.jrr_model_set_faces(self.handle, f.ctypes.data, int(f.shape[0])), 'jrr_model_set_faces')

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.jrr_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class RefineEngine:
    """Per-batch plan over a caller-owned (torch) workspace: jrr_engine_*."""

    def __init__(self, model: Optional[DeviceModel], batch: int, batch_norm: Optional[int] = None, flags: int = 0,
                 device=None):
        """`model` may be None for an engine that serves the discriminators only (flags within POSE_DISC | SHAPE_DISC)"""
        self.lib = model.lib if model is not None else _lib.load()
        self.model = model
        self.device = model.device if model is not None else torch.device(device if device is not None else 'cuda:0')
        self.batch = int(batch)
        self.batch_norm = int(batch_norm or batch)
        self.flags = int(flags) | (FLAG_NO_MODEL if model is None else 0)   # no SMPL workspace for a discriminator-only engine
        self.sil = 32 * ((self.flags >> 16) & 15) or (256 if (self.flags & FLAG_SIL_256) else SIL)      # silhouette image size of this engine
        # Forward-generation counter: the adjoint entry points read the engine's internal state of the MOST RECENT
        # forward (include/jrr.h: "must follow it"), while autograd defers backward.  Every call that overwrites that
        # state bumps the counter; the autograd wrappers compare it with the value saved at forward time and re-run
        # the forward from their saved inputs when it has moved (utils._FindJointsFn, smpl._SMPLVerticesFn,
        # discriminator._PoseDiscFn).
        self.generation = 0
        nbytes = self.lib.jrr_engine_workspace_bytes(self.batch, self.flags)
        # zero-filled: padded rows/columns of several sections are read by the GEMM kernels
        self.workspace = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.device)
        base = (self.workspace.data_ptr() + 255) // 256 * 256
        h = c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.jrr_engine_create(model.handle if model is not None else None, self.batch, int(batch_norm or batch), c_void_p(base), nbytes,
                                             self.flags, byref(h)), 'jrr_engine_create')
        self.handle = h
        info = (c_int32 * 9)()
        self.lib.jrr_engine_info(self.handle, info, 9)
        self.info = dict(zip(['B', 'BP', 'batch_norm', 'nvc', 'nvcb', 'nsplit', 'nsplitJ', 'flags', 'joint_sparse'], list(info)))

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.jrr_engine_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------------------
    def _s(self):
        return stream_ptr(self.device)

    def _chk(self, t, shape, name):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == tuple(shape), \
            f'{name}: expected contiguous float32 cuda tensor of shape {tuple(shape)}, got {tuple(t.shape)} {t.dtype}'
        return t

    # -- configuration ---------------------------------------------------------------------
    def set_batch_norm(self, n: int):
        check(self.lib.jrr_engine_set_batch_norm(self.handle, int(n)), 'set_batch_norm')
        self.batch_norm = int(n)
        self.info['batch_norm'] = int(n)

    def set_folded(self, enabled: bool):
        """refine_run through the folded regressor tables (engine must have FLAG_FOLDED)"""
        check(self.lib.jrr_engine_set_folded(self.handle, int(bool(enabled)), self._s()), 'set_folded')

    def set_j_regressor(self, J: torch.Tensor, mask: Optional[torch.Tensor] = None):
        self.generation += 1
        J = J.detach().to(self.device, torch.float32).contiguous()      # any stride / device tag
        self._chk(J, (NUM_H36M, NUM_VERTS), 'J_regressor')
        if mask is not None:
            mask = self._chk(mask.detach().to(self.device, torch.float32).contiguous(), (NUM_H36M, NUM_VERTS), 'mask')
        check(self.lib.jrr_engine_set_j_regressor(self.handle, ptr(J), ptr(mask), self._s()), 'set_j_regressor')

    def set_pose_disc(self, flat: torch.Tensor):
        self.generation += 1
        flat = self._chk(flat.detach().to(self.device, torch.float32).contiguous(), (DISC_PARAMS,), 'disc params')
        check(self.lib.jrr_engine_set_pose_disc(self.handle, ptr(flat), self._s()), 'set_pose_disc')

    def set_shape_disc(self, flat: torch.Tensor):
        flat = self._chk(flat.detach().to(self.device, torch.float32).contiguous(), (SHAPE_DISC_PARAMS,), 'shape disc params')
        check(self.lib.jrr_engine_set_shape_disc(self.handle, ptr(flat), self._s()), 'set_shape_disc')

    # -- operators --------------------------------------------------------------------------
    def find_joints_forward(self, betas, x6d=None, R=None, return_verts=False):
        self.generation += 1
        B = self.batch
        self._chk(betas, (B, NUM_BETAS), 'betas')
        if x6d is not None:
            self._chk(x6d, (B, NUM_JOINTS, 6), 'x6d')
        if R is not None:
            self._chk(R, (B, NUM_JOINTS, 3, 3), 'R')
        joints = torch.empty(B, NUM_H36M, 3, device=self.device)
        verts = torch.empty(B, NUM_VERTS, 3, device=self.device) if return_verts else None
        check(self.lib.jrr_find_joints_forward(self.handle, ptr(x6d), ptr(R), ptr(betas), ptr(joints), ptr(verts),
                                               self._s()), 'find_joints_forward')
        return (joints, verts) if return_verts else joints

    def find_joints_backward(self, betas, djoints, x6d=None, R=None, want_dJ=False):
        B = self.batch
        self._chk(djoints, (B, NUM_H36M, 3), 'djoints')
        dx = torch.empty(B, NUM_JOINTS, 6, device=self.device) if x6d is not None else None
        dR = torch.empty(B, NUM_JOINTS, 3, 3, device=self.device) if R is not None else None
        db = torch.empty(B, NUM_BETAS, device=self.device)
        dJ = torch.empty(NUM_H36M, NUM_VERTS, device=self.device) if want_dJ else None
        check(self.lib.jrr_find_joints_backward(self.handle, ptr(x6d), ptr(R), ptr(betas), ptr(djoints), ptr(dx), ptr(dR),
                                                ptr(db), ptr(dJ), self._s()), 'find_joints_backward')
        return (dx if x6d is not None else dR), db, dJ

    def pose_disc_forward(self, x6d):
        self.generation += 1
        self._chk(x6d, (self.batch, NUM_JOINTS, 6), 'x6d')
        out = torch.empty(self.batch, 25, device=self.device)
        check(self.lib.jrr_pose_disc_forward(self.handle, ptr(x6d), ptr(out), self._s()), 'pose_disc_forward')
        return out

    def pose_disc_backward_input(self, x6d, weight: float, target: float):
        dx = torch.empty(self.batch, NUM_JOINTS, 6, device=self.device)
        check(self.lib.jrr_pose_disc_backward_input(self.handle, ptr(x6d), float(weight), float(target), ptr(dx),
                                                    self._s()), 'pose_disc_backward_input')
        return dx

    def pose_disc_backward_params(self, x6d, target: float, dparams: torch.Tensor):
        """dparams += d mean((D(x)-target)^2)/d weights; returns per-pose sum_k (D-target)^2"""
        self.generation += 1
        self._chk(x6d, (self.batch, NUM_JOINTS, 6), 'x6d')
        self._chk(dparams, (DISC_PARAMS,), 'dparams')
        sq = torch.empty(self.batch, device=self.device)
        check(self.lib.jrr_pose_disc_backward_params(self.handle, ptr(x6d), float(target), ptr(dparams), ptr(sq), self._s()),
              'pose_disc_backward_params')
        return sq

    def shape_disc_backward_params(self, betas, target: float, dparams: torch.Tensor):
        self._chk(betas, (self.batch, NUM_BETAS), 'betas')
        self._chk(dparams, (SHAPE_DISC_PARAMS,), 'dparams')
        sq = torch.empty(self.batch, device=self.device)
        check(self.lib.jrr_shape_disc_backward_params(self.handle, ptr(betas), float(target), ptr(dparams), ptr(sq), self._s()),
              'shape_disc_backward_params')
        return sq

    def pose_disc_vjp_params(self, x6d, gout, dparams: torch.Tensor):
        """dparams += (dD/dweights)^T gout for an upstream gradient gout (B,25); runs its own forward"""
        self.generation += 1
        self._chk(x6d, (self.batch, NUM_JOINTS, 6), 'x6d')
        self._chk(gout, (self.batch, 25), 'gout')
        self._chk(dparams, (DISC_PARAMS,), 'dparams')
        check(self.lib.jrr_pose_disc_vjp_params(self.handle, ptr(x6d), ptr(gout), ptr(dparams), self._s()), 'pose_disc_vjp_params')

    def shape_disc_vjp_params(self, betas, gout, dparams: torch.Tensor):
        self._chk(betas, (self.batch, NUM_BETAS), 'betas')
        self._chk(gout, (self.batch,), 'gout')
        self._chk(dparams, (SHAPE_DISC_PARAMS,), 'dparams')
        check(self.lib.jrr_shape_disc_vjp_params(self.handle, ptr(betas), ptr(gout), ptr(dparams), self._s()), 'shape_disc_vjp_params')

    def shape_disc_forward(self, betas):
        """Shape_Discriminator.forward: betas (B,10) -> (B) sigmoid scores"""
        self._chk(betas, (self.batch, NUM_BETAS), 'betas')
        out = torch.empty(self.batch, device=self.device)
        check(self.lib.jrr_shape_disc_forward(self.handle, ptr(betas), ptr(out), self._s()), 'shape_disc_forward')
        return out

    def shape_disc_vjp_input(self, betas, gout):
        self._chk(betas, (self.batch, NUM_BETAS), 'betas')
        self._chk(gout, (self.batch,), 'gout')
        db = torch.empty(self.batch, NUM_BETAS, device=self.device)
        check(self.lib.jrr_shape_disc_vjp_input(self.handle, ptr(betas), ptr(gout), ptr(db), self._s()), 'shape_disc_vjp_input')
        return db

    def refine_aux_losses(self, pose_disc=True, shape_disc=False):
        """per-pose squared adversarial errors of the last refine_run iteration (pose: summed over the 25 outputs)"""
        pd = torch.empty(self.batch, device=self.device) if pose_disc else None
        sd = torch.empty(self.batch, device=self.device) if shape_disc else None
        check(self.lib.jrr_refine_aux_losses(self.handle, ptr(pd), ptr(sd), self._s()), 'refine_aux_losses')
        return pd, sd

    def pose_disc_vjp_input(self, x6d, gout):
        """input gradient for an arbitrary upstream gradient gout (B,25); follows pose_disc_forward"""
        self._chk(gout, (self.batch, 25), 'gout')
        dx = torch.empty(self.batch, NUM_JOINTS, 6, device=self.device)
        check(self.lib.jrr_pose_disc_vjp_input(self.handle, ptr(x6d), ptr(gout), ptr(dx), self._s()), 'pose_disc_vjp_input')
        return dx

    def find_joints_after_j_step(self, betas, x6d):
        """joints (B,17,3) of the poses of the J step that preceded (j_regressor_grad* + j_step_apply* on these very buffers) with the
        stepped regressor, from that step's stored vertices -- no second SMPL forward (jrr_find_joints_after_j_step)"""
        joints = torch.empty(self.batch, NUM_H36M, 3, device=self.device)
        check(self.lib.jrr_find_joints_after_j_step(self.handle, ptr(x6d), ptr(betas), ptr(joints), self._s()), 'find_joints_after_j_step')
        return joints

    def posed_joints(self, betas):
        """(B,24,3) posed SMPL joints (smplx J_transformed) of the most recent forward on this engine with these betas"""
        self._chk(betas, (self.batch, NUM_BETAS), 'betas')
        out = torch.empty(self.batch, NUM_JOINTS, 3, device=self.device)
        check(self.lib.jrr_smpl_posed_joints(self.handle, ptr(betas), ptr(out), self._s()), 'smpl_posed_joints')
        return out

    def posed_joints_backward(self, betas, djoints24, x6d=None, R=None):
        """adjoint of posed_joints: (B,24,3) -> (dx6d or dR, dbetas) through the kinematic chain of the most recent forward"""
        B = self.batch
        self._chk(djoints24, (B, NUM_JOINTS, 3), 'djoints24')
        dx = torch.empty(B, NUM_JOINTS, 6, device=self.device) if x6d is not None else None
        dR = torch.empty(B, NUM_JOINTS, 3, 3, device=self.device) if R is not None else None
        db = torch.empty(B, NUM_BETAS, device=self.device)
        check(self.lib.jrr_smpl_posed_joints_backward(self.handle, ptr(x6d), ptr(R), ptr(betas), ptr(djoints24), ptr(dx), ptr(dR),
                                                      ptr(db), self._s()), 'smpl_posed_joints_backward')
        return (dx if x6d is not None else dR), db

    def smpl_vertices_backward(self, betas, dverts, x6d=None, R=None):
        B = self.batch
        self._chk(dverts, (B, NUM_VERTS, 3), 'dverts')
        dx = torch.empty(B, NUM_JOINTS, 6, device=self.device) if x6d is not None else None
        dR = torch.empty(B, NUM_JOINTS, 3, 3, device=self.device) if R is not None else None
        db = torch.empty(B, NUM_BETAS, device=self.device)
        check(self.lib.jrr_smpl_vertices_backward(self.handle, ptr(x6d), ptr(R), ptr(betas), ptr(dverts), ptr(dx), ptr(dR),
                                                  ptr(db), self._s()), 'smpl_vertices_backward')
        return (dx if x6d is not None else dR), db

    def set_reprojection(self, gt_j2d=None, cam=None, cam_m=None, cam_v=None):
        """enable (tensors) / disable (None) the 2-D term of refine_run; the tensors must outlive the runs"""
        if gt_j2d is not None:
            self._chk(gt_j2d, (self.batch, NUM_H36M, 2), 'gt_j2d')
            for t, n in ((cam, 'cam'), (cam_m, 'cam_m'), (cam_v, 'cam_v')):
                self._chk(t, (self.batch, 3), n)
        self._reproj_refs = (gt_j2d, cam, cam_m, cam_v)
        check(self.lib.jrr_engine_set_reprojection(self.handle, ptr(gt_j2d), ptr(cam), ptr(cam_m), ptr(cam_v)), 'set_reprojection')

    def camera_prefit(self, x6d, betas, gt_j2d, cam, n_steps: int = 1000, lr: float = 1e-2):
        self.generation += 1
        sq = torch.empty(self.batch, device=self.device)
        check(self.lib.jrr_camera_prefit(self.handle, ptr(x6d), ptr(betas), ptr(gt_j2d), ptr(cam), int(n_steps), float(lr),
                                         ptr(sq), self._s()), 'camera_prefit')
        return sq

    def silhouette_forward(self, verts, cam):
        """render_mesh alpha channel: verts (B,6890,3) in SMPL space, cam (B,3) -> (B,S,S), S = 224 (256 with FLAG_SIL_256)"""
        self.generation += 1
        self._chk(verts, (self.batch, NUM_VERTS, 3), 'verts')
        self._chk(cam, (self.batch, 3), 'cam')
        alpha = torch.empty(self.batch, self.sil, self.sil, device=self.device)
        check(self.lib.jrr_silhouette_forward(self.handle, ptr(verts), ptr(cam), ptr(alpha), self._s()), 'silhouette_forward')
        return alpha

    def silhouette_backward(self, galpha):
        self._chk(galpha, (self.batch, self.sil, self.sil), 'galpha')
        dverts = torch.empty(self.batch, NUM_VERTS, 3, device=self.device)
        dcam = torch.empty(self.batch, 3, device=self.device)
        check(self.lib.jrr_silhouette_backward(self.handle, ptr(galpha), ptr(dverts), ptr(dcam), self._s()), 'silhouette_backward')
        return dverts, dcam

    def silhouette_pix_to_face(self) -> torch.Tensor:
        """(B,224,224) int32 nearest-face index per pixel (-1 = background) of the most recent rasterisation"""
        p2f = torch.empty(self.batch, self.sil, self.sil, dtype=torch.int32, device=self.device)
        check(self.lib.jrr_silhouette_pix_to_face(self.handle, ptr(p2f), self._s()), 'silhouette_pix_to_face')
        return p2f

    def silhouette_loss_grad(self, x6d, betas, cam, mask, want_dverts=True):
        """the silhouette term as the fused loop evaluates it: per-pose sum (silhouette - mask)^2, d/dverts (B,6890,3) and
        d/dcam (B,3) of 100 * mean((silhouette - mask)^2)  (jrr_silhouette_loss_grad)"""
        self.generation += 1
        B = self.batch
        self._chk(x6d, (B, NUM_JOINTS, 6), 'x6d'); self._chk(betas, (B, NUM_BETAS), 'betas')
        self._chk(cam, (B, 3), 'cam'); self._chk(mask, (B, self.sil, self.sil), 'mask')
        sq = torch.empty(B, device=self.device)
        dv = torch.empty(B, NUM_VERTS, 3, device=self.device) if want_dverts else None
        dc = torch.empty(B, 3, device=self.device)
        check(self.lib.jrr_silhouette_loss_grad(self.handle, ptr(x6d), ptr(betas), ptr(cam), ptr(mask), ptr(sq), ptr(dv), ptr(dc),
                                                self._s()), 'silhouette_loss_grad')
        return sq, dv, dc

    def set_silhouette(self, mask=None, cam=None, cam_m=None, cam_v=None):
        """enable (tensors) / disable (None) the silhouette term of refine_run"""
        if mask is not None:
            self._chk(mask, (self.batch, self.sil, self.sil), 'mask')
            for t, n in ((cam, 'cam'), (cam_m, 'cam_m'), (cam_v, 'cam_v')):
                self._chk(t, (self.batch, 3), n)
        self._sil_refs = (mask, cam, cam_m, cam_v)
        check(self.lib.jrr_engine_set_silhouette(self.handle, ptr(mask), ptr(cam), ptr(cam_m), ptr(cam_v)), 'set_silhouette')

    def _refine_args(self, x6d, betas, gt_centred_mm, adam_m, adam_v, step):
        B = self.batch
        self._chk(x6d, (B, NUM_JOINTS, 6), 'x6d')
        self._chk(betas, (B, NUM_BETAS), 'betas')
        self._chk(gt_centred_mm, (B, NUM_H36M, 3), 'gt')
        self._chk(adam_m, (B, 154), 'adam_m')
        self._chk(adam_v, (B, 154), 'adam_v')
        assert step.dtype == torch.int32 and step.is_cuda

    def refine_run(self, x6d, betas, gt_centred_mm, adam_m, adam_v, step, lr: float, n_iters: int, sqerr=None,
                   after_j_step: bool = False):
        """n_iters inner iterations in ONE C call.  after_j_step=True (jrr_refine_run_after_j_step): the caller states that
        the previous call on this engine was j_regressor_grad (+ j_step_apply) on these very buffers; the first iteration
        then reuses that forward.  A mismatch the engine can detect raises instead of reusing a stale forward."""
        self.generation += 1
        self._refine_args(x6d, betas, gt_centred_mm, adam_m, adam_v, step)
        fn = self.lib.jrr_refine_run_after_j_step if after_j_step else self.lib.jrr_refine_run
        check(fn(self.handle, ptr(x6d), ptr(betas), ptr(gt_centred_mm), ptr(adam_m), ptr(adam_v),
                 ptr(step), float(lr), int(n_iters), ptr(sqerr), self._s()), 'refine_run')

    def refine_run_j_steps(self, x6d, betas, gt_centred_mm, adam_m, adam_v, step, lr: float, n_iters: int, j_every: int,
                           J, J_m, J_v, J_step, j_lr: float, mask=None, sqerr=None, j_sqerr=None, after_j_step: bool = False,
                           reuse_forward: bool = True):
        """the inner loop with a J step after every j_every-th iteration, all inside ONE C call (single process only:
        there is no collective between the two halves of a J step); J / J_m / J_v / J_step are updated in place.
        reuse_forward=False: the iteration after a J step repeats its SMPL forward (include/jrr.h, after_j_step bit 1)"""
        self.generation += 1
        self._refine_args(x6d, betas, gt_centred_mm, adam_m, adam_v, step)
        for t, n in ((J, 'J'), (J_m, 'J_m'), (J_v, 'J_v')):
            self._chk(t, (NUM_H36M, NUM_VERTS), n)
        assert J_step.dtype == torch.int32 and J_step.is_cuda
        check(self.lib.jrr_refine_run_j_steps(self.handle, ptr(x6d), ptr(betas), ptr(gt_centred_mm), ptr(adam_m), ptr(adam_v),
                                              ptr(step), float(lr), int(n_iters), ptr(sqerr), int(j_every), ptr(J), ptr(J_m),
                                              ptr(J_v), ptr(J_step), float(j_lr), ptr(mask), ptr(j_sqerr),
                                              int(bool(after_j_step)) | (0 if reuse_forward else 2),
                                              self._s()), 'refine_run_j_steps')

    def j_step_apply(self, J, dJ, J_m, J_v, J_step, lr: float, mask=None):
        """second half of the J step: Adam on the raw regressor with the (all-reduced) gradient + re-normalisation, one call"""
        self.generation += 1
        for t, n in ((J, 'J'), (dJ, 'dJ'), (J_m, 'J_m'), (J_v, 'J_v')):
            self._chk(t, (NUM_H36M, NUM_VERTS), n)
        assert J_step.dtype == torch.int32 and J_step.is_cuda
        check(self.lib.jrr_j_step_apply(self.handle, ptr(J), ptr(dJ), ptr(J_m), ptr(J_v), ptr(J_step), float(lr), ptr(mask),
                                        self._s()), 'j_step_apply')

    def set_loss_history(self, records: int = 0, every: int = 10):
        """scripts/optimize.py:255-261: keep the five weighted loss terms of every `every`-th iteration (records = 0: off)"""
        self._hist = torch.zeros(records, 5, device=self.device) if records > 0 else None
        check(self.lib.jrr_engine_set_loss_history(self.handle, ptr(self._hist), int(records), int(every)), 'set_loss_history')

    def loss_history(self) -> Optional[torch.Tensor]:
        """(n,5) device tensor of the records written so far: [j2d/100, silhouette*100, joint*10000, poseD*10, shapeD*10]"""
        if getattr(self, '_hist', None) is None:
            return None
        return self._hist[:self.lib.jrr_engine_loss_history_count(self.handle)]

    PROF_CLASSES = ['k_prep_fwd', 'k_lbs_fwd', 'k_joints_loss', 'k_lbs_bwd', 'k_gemm_tn_blend_adjoint', 'pose_disc_gemms',
                    'k_shape_disc', 'k_prep_bwd', 'silhouette_fwd_bwd']

    def set_profiling(self, on: bool):
        check(self.lib.jrr_engine_set_profiling(self.handle, int(bool(on))), 'set_profiling')

    def profile_read(self):
        """mean ms per launch (HIP events on the launch stream) and sample counts per kernel class"""
        ms = (ctypes.c_float * 9)()
        n = (c_int32 * 9)()
        check(self.lib.jrr_engine_profile_read(self.handle, ms, n), 'profile_read')
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(self.PROF_CLASSES)}

    def probe_read(self):
        """shader-clock probe of the last profiled k_lbs_fwd launch (include/jrr.h jrr_engine_probe_read):
        (resident shader clocks of workgroup 0 / wave 0, MFMA instructions it issued, waves per SIMD, clocks per MFMA,
        the interval in ns)"""
        out = (ctypes.c_int64 * 5)()
        check(self.lib.jrr_engine_probe_read(self.handle, out), 'probe_read')
        return tuple(int(x) for x in out)

    # -- J step over the regressor's support (data parallelism: 8.7 KB instead of 468 KB per all-reduce) --------------
    J_SUPPORT_CAP = 128

    def j_support_info(self):
        """(counts per row [17], fits): positive entries of the normalised regressor per row and whether every row has at most
        128 of them.  SYNCHRONOUS; call once after set_j_regressor (J steps only shrink the support)."""
        counts = (c_int32 * NUM_H36M)()
        fits = c_int32(0)
        check(self.lib.jrr_j_support_info(self.handle, counts, byref(fits), self._s()), 'j_support_info')
        return [int(c) for c in counts], bool(fits.value)

    def support_tiles(self):
        """(active, n_tiles): whether the next joint-loss iteration runs on the regressor's support tiles only (FLAG_SUPPORT_TILES
        after j_support_info reported fits) and on how many of the 216 tiles"""
        n = c_int32(0)
        rc = self.lib.jrr_engine_support_tiles(self.handle, byref(n))
        if rc < 0:
            check(rc, 'support_tiles')
        return bool(rc), int(n.value)

    def support_vertices(self):
        """(active, n_vertices): whether those iterations run per support VERTEX, one launch per iteration and 32-pose group
        (include/jrr.h jrr_engine_support_vertices: the support has at most 64 vertices), and on how many vertices"""
        n = c_int32(0)
        rc = self.lib.jrr_engine_support_vertices(self.handle, byref(n))
        if rc < 0:
            check(rc, 'support_vertices')
        return bool(rc), int(n.value)

    def j_regressor_grad_support(self, x6d, betas, gt_centred_mm, out, sqerr=None, joints=None):
        """j_regressor_grad with the gradient delivered on the support only: out (17,128)"""
        self.generation += 1
        self._chk(out, (NUM_H36M, self.J_SUPPORT_CAP), 'dJs')
        check(self.lib.jrr_j_regressor_grad_support(self.handle, ptr(x6d), ptr(betas), ptr(gt_centred_mm), ptr(out), ptr(sqerr),
                                                    ptr(joints), self._s()), 'j_regressor_grad_support')
        return out

    def j_step_apply_support(self, J, dJs, J_m, J_v, J_step, lr: float, mask=None):
        self.generation += 1
        for t, n in ((J, 'J'), (J_m, 'J_m'), (J_v, 'J_v')):
            self._chk(t, (NUM_H36M, NUM_VERTS), n)
        self._chk(dJs, (NUM_H36M, self.J_SUPPORT_CAP), 'dJs')
        assert J_step.dtype == torch.int32 and J_step.is_cuda
        check(self.lib.jrr_j_step_apply_support(self.handle, ptr(J), ptr(dJs), ptr(J_m), ptr(J_v), ptr(J_step), float(lr), ptr(mask),
                                                self._s()), 'j_step_apply_support')

    def j_regressor_grad(self, x6d, betas, gt_centred_mm, sqerr=None, out=None, joints=None):
        """local dJ (first half of the J step); `out` (17,6890): write into a caller-owned buffer (e.g. a slice of the flat
        all-reduce bucket) instead of allocating; `joints` (B,17,3): receives the joints of this forward (old regressor)"""
        self.generation += 1
        dJ = out if out is not None else torch.empty(NUM_H36M, NUM_VERTS, device=self.device)
        self._chk(dJ, (NUM_H36M, NUM_VERTS), 'dJ')
        if joints is not None:
            self._chk(joints, (self.batch, NUM_H36M, 3), 'joints')
        check(self.lib.jrr_j_regressor_grad(self.handle, ptr(x6d), ptr(betas), ptr(gt_centred_mm), ptr(dJ), ptr(sqerr),
                                            ptr(joints), self._s()), 'j_regressor_grad')
        return dJ


# ---- free operator functions -----------------------------------------------------------------
def rot6d_forward(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    x = x.contiguous().view(-1, 6)
    R = torch.empty(x.shape[0], 3, 3, device=x.device)
    check(lib.jrr_rot6d_forward(ptr(x), ptr(R), x.shape[0], stream_ptr(x.device)), 'rot6d_forward')
    return R


def rot6d_backward(x: torch.Tensor, dR: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    x = x.contiguous().view(-1, 6)
    dR = dR.contiguous()
    dx = torch.empty_like(x)
    check(lib.jrr_rot6d_backward(ptr(x), ptr(dR), ptr(dx), x.shape[0], stream_ptr(x.device)), 'rot6d_backward')
    return dx


def rodrigues_forward(aa: torch.Tensor) -> torch.Tensor:
    """smplx batch_rodrigues: axis-angle (N,3) -> (N,3,3)"""
    lib = _lib.load()
    aa = aa.contiguous().view(-1, 3).float()
    R = torch.empty(aa.shape[0], 3, 3, device=aa.device)
    check(lib.jrr_rodrigues_forward(ptr(aa), ptr(R), aa.shape[0], stream_ptr(aa.device)), 'rodrigues_forward')
    return R


def rodrigues_backward(aa: torch.Tensor, dR: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    aa = aa.contiguous().view(-1, 3).float()
    dR = dR.contiguous().float()
    daa = torch.empty_like(aa)
    check(lib.jrr_rodrigues_backward(ptr(aa), ptr(dR), ptr(daa), aa.shape[0], stream_ptr(aa.device)), 'rodrigues_backward')
    return daa


def project_joints(joints: torch.Tensor, cam: torch.Tensor) -> torch.Tensor:
    """return_2d_joints core (scripts/renderer.py:35-49): (B,17,3), (B,3) -> screen xy (B,17,2)"""
    lib = _lib.load()
    B = joints.shape[0]
    out = torch.empty(B, NUM_H36M, 2, device=joints.device)
    check(lib.jrr_project_joints(ptr(joints.contiguous()), ptr(cam.contiguous()), ptr(out), B, stream_ptr(joints.device)),
          'project_joints')
    return out


def evaluate(pred_j3d: torch.Tensor, target_j3d_mm: torch.Tensor):
    """per-pose joint error and Procrustes-aligned joint error (metres), on device"""
    lib = _lib.load()
    B = pred_j3d.shape[0]
    err = torch.empty(B, device=pred_j3d.device)
    err_pa = torch.empty(B, device=pred_j3d.device)
    check(lib.jrr_evaluate(ptr(pred_j3d.contiguous()), ptr(target_j3d_mm.contiguous()), ptr(err), ptr(err_pa), B,
                           stream_ptr(pred_j3d.device)), 'evaluate')
    return err, err_pa


def joint_loss(joints, gt_centred_mm, weight: float, batch_norm: Optional[int] = None, want_grad=True):
    lib = _lib.load()
    B = joints.shape[0]
    sq = torch.empty(B, device=joints.device)
    dj = torch.empty_like(joints) if want_grad else None
    check(lib.jrr_joint_loss(ptr(joints.contiguous()), ptr(gt_centred_mm.contiguous()), float(weight), B,
                             int(batch_norm or B), ptr(sq), ptr(dj), stream_ptr(joints.device)), 'joint_loss')
    return sq, dj


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    lib = _lib.load()
    assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
    check(lib.jrr_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(step), float(lr), float(beta1), float(beta2),
                            float(eps), stream_ptr(p.device)), 'adam_step')
