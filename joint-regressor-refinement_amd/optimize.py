"""The optimisation driver: `optimize_pose_refiner()` restated from the reference
(/root/reference/scripts/optimize.py:88-337) on the HIP path.

Per outer batch (reference line numbers):
  :158-162  batch to device, ground-truth joints pelvis-centred
  :164-185  SPIN initial pose (B,24,6) / betas / camera  -> here: seeded synthetic "SPIN-init" batches
            (the SPIN network, its checkpoint and Human3.6M are absent)
  :187-199  camera pre-fit on the 2-D loss                 -> "next" row f1 (not in BASELINE configs 1-4)
  :201-202  fresh Adam over [pose, orient, betas, cam], lr 1e-2
  :220-265  100 inner iterations                            -> ONE C-ABI call, jrr_refine_run
  :276-284  pose-discriminator update, Adam(lr=args.opt_disc_learning_rate)
  :286-293  shape-discriminator update
  :300-312  J_regressor step, Adam(lr=args.j_reg_lr)        -> + one RCCL all-reduce under data parallelism
  :314-337  MPJPE / PA-MPJPE before and after the J step, logging

Data parallelism (new): one process per GPU, the batch is sharded contiguously, per-pose state is
rank-local, the MSE means are normalised by the GLOBAL batch; the only collectives are one
sum-all-reduce per shared-parameter step (J gradient; discriminator gradients).
"""
from __future__ import annotations

import time
from typing import Dict, Optional

import numpy as np
import torch

from . import checkpoint, dist as jdist, engine as _engine, smpl_model, utils
from .args import args
from .discriminator import Discriminator, Shape_Discriminator
from .smpl import SMPL


class AdamState:
    """torch.optim.Adam state for one flat parameter tensor, stepped by jrr_adam_step."""

    def __init__(self, param: torch.Tensor, lr: float):
        self.m = torch.zeros_like(param)
        self.v = torch.zeros_like(param)
        self.step = torch.zeros(1, dtype=torch.int32, device=param.device)
        self.lr = lr

    def apply(self, param: torch.Tensor, grad: torch.Tensor):
        self.step += 1
        _engine.adam_step(param, grad, self.m, self.v, self.step, self.lr)


def optimize_pose_refiner(log=print) -> Dict:
    dist = jdist.init()
    rank, local_rank, world = jdist.env_rank_world()
    device = torch.device(args.device if world == 1 else f'cuda:{local_rank}')
    torch.cuda.set_device(device)
    utils.set_seed(args.seed)

    smpl = SMPL(args.smpl_dir, batch_size=1).to(device)                                   # :96-99
    J_regressor = torch.from_numpy(smpl_model.default_h36m_regressor(args.j_regressor_init)).float().to(device)   # :105-107
    j_reg_mask = utils.find_j_reg_mask(J_regressor)                                        # :130

    use_pd, use_sd = not args.no_pose_disc, bool(args.shape_disc)
    pose_discriminator = Discriminator()                                                   # :112-113 (torch default init)
    shape_discriminator = Shape_Discriminator()                                            # :119-120
    disc_flat = pose_discriminator.flat_parameters().to(device).contiguous()
    sdisc_flat = shape_discriminator.flat_parameters().to(device).contiguous()
    disc_opt = AdamState(disc_flat, args.opt_disc_learning_rate)                          # :116-117
    sdisc_opt = AdamState(sdisc_flat, args.opt_disc_learning_rate)                        # :122-123
    J_opt = AdamState(J_regressor, args.j_reg_lr)                                          # :125-126

    B_global = args.batch_size
    lo, hi = jdist.shard_bounds(B_global, rank, world)
    B = hi - lo
    flags = _engine.FLAG_KEEP_VERTS | (_engine.FLAG_POSE_DISC if use_pd else 0) | (_engine.FLAG_SHAPE_DISC if use_sd else 0) \
        | (_engine.FLAG_SILHOUETTE if args.silhouette else 0)
    eng = _engine.RefineEngine(smpl.device_model, B, batch_norm=B_global, flags=flags)
    eng.set_j_regressor(J_regressor, j_reg_mask)
    if use_pd:
        eng.set_pose_disc(disc_flat)
    if use_sd:
        eng.set_shape_disc(sdisc_flat)

    history = []
    for it in range(args.synthetic_batches):                                               # :144-148
        full = smpl_model.synthetic_batch(smpl.model_np, J_regressor.detach().cpu().numpy(), B_global, seed=args.seed * 1000 + it)
        spin_pose = torch.from_numpy(full['pose6d'][lo:hi]).to(device).contiguous()        # :166-168 (synthetic SPIN-init)
        spin_betas = torch.from_numpy(full['betas'][lo:hi]).to(device).contiguous()
        gt_j3d = utils.move_pelvis(torch.from_numpy(full['gt_j3d'][lo:hi]).to(device)).contiguous()   # :162
        x6d = spin_pose.clone()                                                            # :177-179 pose + orient
        betas = spin_betas.clone()
        m = torch.zeros(B, 154, device=device)                                             # :201-202 fresh optimizer
        v = torch.zeros(B, 154, device=device)
        step = torch.zeros(1, dtype=torch.int32, device=device)
        sq = torch.zeros(B, device=device)

        # ---- camera pre-fit + 2-D term (:170-173,187-199,231-233; row f1) on synthetic 2-D targets ----
        cam = torch.from_numpy(full['cam'][lo:hi]).to(device).contiguous()               # :170-172 pred_cam_t
        cam_m, cam_v = torch.zeros_like(cam), torch.zeros_like(cam)
        if args.reprojection:
            gt_j2d = _synthetic_gt_j2d(eng, x6d, betas, cam, args.seed * 1000 + it)
            eng.camera_prefit(x6d, betas, gt_j2d, cam, n_steps=args.camera_iters, lr=1e-2)   # :187-199
            eng.set_reprojection(gt_j2d, cam, cam_m, cam_v)
        if args.silhouette:                                                                # :234-237 (row f2)
            sil_mask = _synthetic_mask(eng, x6d, betas, cam, args.seed * 1000 + it)
            eng.set_silhouette(sil_mask, cam, cam_m, cam_v)

        t0 = time.perf_counter()
        done = 0
        while done < args.inner_iters:                                                     # :220-265
            seg = min(args.j_step_every - done % args.j_step_every, args.inner_iters - done)
            eng.refine_run(x6d, betas, gt_j3d, m, v, step, 1e-2, seg, sqerr=sq)
            done += seg
            if done % args.j_step_every == 0 and done < args.inner_iters:
                _j_step(eng, J_regressor, J_opt, x6d, betas, gt_j3d, j_reg_mask)
        joint_loss = _global_mean(sq, B_global * 51)
        if args.reprojection:
            eng.set_reprojection(None)
        if args.silhouette:
            eng.set_silhouette(None)

        # ---- pose-discriminator update (:276-284) ----
        pose_d_loss = None
        if use_pd:
            g = torch.zeros_like(disc_flat)
            l0 = eng.pose_disc_backward_params(x6d, 0.0, g)                                # MSE(D(opt.detach()), 0)
            l1 = eng.pose_disc_backward_params(spin_pose, 1.0, g)                          # MSE(D(spin), 1)
            jdist.all_reduce_sum_(g)
            disc_opt.apply(disc_flat, g)
            eng.set_pose_disc(disc_flat)
            pose_d_loss = _global_mean(l0 + l1, B_global * 25)
        # ---- shape-discriminator update (:286-293) ----
        shape_d_loss = None
        if use_sd:
            g = torch.zeros_like(sdisc_flat)
            l0 = eng.shape_disc_backward_params(betas, 0.0, g)
            l1 = eng.shape_disc_backward_params(spin_betas, 1.0, g)
            jdist.all_reduce_sum_(g)
            sdisc_opt.apply(sdisc_flat, g)
            eng.set_shape_disc(sdisc_flat)
            shape_d_loss = _global_mean(l0 + l1, B_global)

        # ---- J_regressor step (:300-312) and before/after metrics (:314-321) ----
        joints_before = eng.find_joints_forward(betas, x6d=x6d)
        j_err = _j_step(eng, J_regressor, J_opt, x6d, betas, gt_j3d, j_reg_mask)
        joints_after = eng.find_joints_forward(betas, x6d=x6d)
        gt_mm = torch.from_numpy(full['gt_j3d'][lo:hi]).to(device)
        mpjpe_b, pampjpe_b = utils.evaluate(joints_before, gt_mm)
        mpjpe_a, pampjpe_a = utils.evaluate(joints_after, gt_mm)
        torch.cuda.synchronize()
        rec = {'batch': it, 'joint_loss': joint_loss, 'pose_discriminator_loss': pose_d_loss,
               'shape_discriminator_loss': shape_d_loss, 'j_regressor_error': j_err, 'mpjpe': float(mpjpe_a),
               'pampjpe': float(pampjpe_a), 'mpjpe difference': float(mpjpe_b - mpjpe_a),
               'pampjpe difference': float(pampjpe_b - pampjpe_a), 'seconds': time.perf_counter() - t0}
        history.append(rec)
        if rank == 0:
            log(rec)                                                                        # :323-337 (wandb.log analogue)
            if args.wandb_log:
                try:
                    import wandb
                    wandb.log(rec)
                except ImportError:
                    pass

    if args.save_j_regressor and rank == 0:
        checkpoint.save_j_regressor(J_regressor, args.save_j_regressor)
    return {'history': history, 'J_regressor': J_regressor, 'disc_flat': disc_flat}


def _synthetic_gt_j2d(eng, x6d, betas, cam, seed):
    """2-D targets in the 224-crop pixel frame (scripts/data.py:134-138): the current joints seen through a
    perturbed camera, plus pixel noise (stand-in for the H36M annotations)."""
    from . import renderer
    g = torch.Generator().manual_seed(seed)
    joints = eng.find_joints_forward(betas, x6d=x6d)
    B = joints.shape[0]
    cam_true = cam + (torch.randn(B, 3, generator=g) * torch.tensor([0.3, 0.3, 3.0])).to(cam.device)
    p = renderer.project_points(joints, cam_true)[..., :2]
    return (p + (torch.randn(B, 17, 2, generator=g) * 2.0).to(cam.device)).contiguous()


def _synthetic_mask(eng, x6d, betas, cam, seed):
    """Target silhouettes (stand-in for batch['mask_rcnn'], scripts/data.py): the current mesh seen through a
    perturbed camera, binarised."""
    g = torch.Generator().manual_seed(seed + 7)
    _, verts = eng.find_joints_forward(betas, x6d=x6d, return_verts=True)
    cam_true = (cam + (torch.randn(cam.shape[0], 3, generator=g) * torch.tensor([0.2, 0.2, 2.0])).to(cam.device)).contiguous()
    return (eng.silhouette_forward(verts, cam_true) > 0).float().contiguous()


def _global_mean(local_sum_tensor: torch.Tensor, denom: int) -> float:
    t = local_sum_tensor.sum().reshape(1).clone()
    jdist.all_reduce_sum_(t)
    return float(t.item()) / denom


def _j_step(eng, J_regressor, J_opt, x6d, betas, gt_j3d, mask) -> float:
    """scripts/optimize.py:300-312: joint MSE of the detached poses w.r.t. the raw regressor, one
    all-reduce of the (17,6890) gradient, replicated Adam, re-normalisation."""
    sq = torch.zeros(x6d.shape[0], device=x6d.device)
    dJ = eng.j_regressor_grad(x6d, betas, gt_j3d, sqerr=sq)
    jdist.all_reduce_sum_(dJ)
    J_opt.apply(J_regressor, dJ)
    eng.set_j_regressor(J_regressor, mask)
    return _global_mean(sq, eng.info['batch_norm'] * 51)
