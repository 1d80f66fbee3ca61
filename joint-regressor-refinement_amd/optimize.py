"""The optimisation driver: `optimize_pose_refiner()` restated from the reference
(/root/reference/scripts/optimize.py:88-337) on the HIP path.

Per outer batch (reference line numbers):
  :132-139  DataLoader over data_set(...)                   -> `--data_root` (precomputed tensors in the reference
            layout, data.py) or seeded synthetic "SPIN-init" batches
  :158-162  batch to device, ground-truth joints pelvis-centred
  :164-185  SPIN initial pose (B,24,6) / betas / camera     -> the dataset's pose / orient / betas tensors stand in for
            the SPIN network's prediction (the network and its checkpoint are absent)
  :187-199  camera pre-fit on the 2-D loss                 -> row f1 (`--reprojection`)
  :201-202  fresh Adam over [pose, orient, betas, cam], lr 1e-2
  :220-265  100 inner iterations                            -> ONE C-ABI call, jrr_refine_run
  :276-284  pose-discriminator update, Adam(lr=args.opt_disc_learning_rate)
  :286-293  shape-discriminator update
  :300-312  J_regressor step, Adam(lr=args.j_reg_lr)        -> + one RCCL all-reduce under data parallelism
  :314-337  MPJPE / PA-MPJPE before and after the J step, logging (all ten scalars of the reference's record)

Data parallelism (new): one process per GPU, the batch is sharded contiguously, per-pose state is
rank-local, the MSE means are normalised by the GLOBAL batch.  The inner loop has no collective.  The
outer step issues ONE sum-all-reduce over a flat bucket

    [ dJ (17 x 6890) | dD (1 840 153) | dShapeD (171) | log-record sums | loss-history records ]  ~ 7.83 MB

followed by the replicated Adam steps; the bucket's scalar tail is read back ONCE per outer batch (no
`.item()` inside the loop, none per shared-parameter step).  The MPJPE of the regressor AFTER its step
needs the stepped regressor, so those two sums ride in the NEXT batch's bucket (a last 2-float
all-reduce flushes them after the final batch): the record of batch k is logged when batch k+1's bucket
has been read.  A J step inside the inner loop (`--j_step_every` < `--inner_iters`) is
jrr_j_regressor_grad[_support] -> one all-reduce (the regressor's support: 8 704 bytes; dist.JStepExchange) ->
jrr_j_step_apply[_support], with no host synchronisation; in a single process the whole loop including those J
steps is ONE C call (jrr_refine_run_j_steps).
Every rank draws the same global batch (same seed) and keeps rows [lo, hi), so a sharded run
reproduces the single-process run on the same global batch.
"""
from __future__ import annotations

import time
from typing import Dict, Iterator

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


# ---- batch sources -----------------------------------------------------------------------------
def pose_to_rot6d(orient: torch.Tensor, pose: torch.Tensor) -> torch.Tensor:
    """Dataset pose tensors -> (B,24,6) 6-D rotations (x[2i+k] = R[i,k], scripts/utils.py:198-204).
    Accepted: 6-D ((B,1,6)/(B,6) + (B,23,6)/(B,138)), rotation matrices ((B,1,3,3) + (B,23,3,3)) or
    axis-angle ((B,3)/(B,1,3) + (B,69)/(B,23,3); converted by the HIP Rodrigues kernel)."""
    B = pose.shape[0]
    if pose.shape[-1] == 6 or pose.numel() == B * 138:
        return torch.cat([orient.reshape(B, 1, 6), pose.reshape(B, 23, 6)], 1).float().contiguous()
    if pose.numel() == B * 23 * 9:
        R = torch.cat([orient.reshape(B, 1, 3, 3), pose.reshape(B, 23, 3, 3)], 1).float()
    elif pose.numel() == B * 69:
        aa = torch.cat([orient.reshape(B, 1, 3), pose.reshape(B, 23, 3)], 1).float().contiguous()
        R = _engine.rodrigues_forward(aa.reshape(-1, 3)).view(B, 24, 3, 3)
    else:
        raise ValueError(f'unrecognised pose tensor shape {tuple(pose.shape)}')
    return R[..., :, :2].reshape(B, 24, 6).contiguous()


def _synthetic_batches(model_np, J_np, B_global: int, n: int, seed: int) -> Iterator[Dict[str, torch.Tensor]]:
    for it in range(n):
        full = smpl_model.synthetic_batch(model_np, J_np, B_global, seed=seed * 1000 + it)
        yield {'pose6d': torch.from_numpy(full['pose6d']), 'betas': torch.from_numpy(full['betas']),
               'gt_j3d': torch.from_numpy(full['gt_j3d']), 'cam': torch.from_numpy(full['cam']), 'seed': seed * 1000 + it}


def _dataset_batches(root: str, B_global: int, seed: int, device, drop_last: bool = False) -> Iterator[Dict[str, torch.Tensor]]:
    """scripts/optimize.py:132-137: DataLoader(data_set("validation"), batch_size, shuffle=True, drop_last=False)
    (scripts/test.py:59-63 uses drop_last=True)."""
    from . import data as jdata
    ds = jdata.data_set('validation', root=root)
    g = torch.Generator().manual_seed(seed)          # every rank shuffles identically
    loader = torch.utils.data.DataLoader(ds, batch_size=B_global, num_workers=0, shuffle=True, drop_last=drop_last, generator=g)
    iterator = iter(loader)
    for it in range(len(loader)):
        failed = 0
        try:                                           # scripts/optimize.py:150-156: a batch that fails to load is reported and skipped
            batch = next(iterator)
        except StopIteration:
            failed = 2
        except Exception as exc:                       # noqa: BLE001  (the reference catches everything here)
            print(f'problem loading batch {it}: {type(exc).__name__}: {exc}')
            failed = 1
        # under data parallelism every rank loads on its own: the ranks agree on the outcome before going on, or a transient
        # error on one rank would pair different batches in the all-reduces that follow (one MAX over a single int per batch)
        failed = jdist.agree_max(failed, device)
        if failed == 2:
            return
        if failed:
            continue
        x6 = pose_to_rot6d(batch['orient'].to(device), batch['pose'].to(device)).cpu()
        yield {'pose6d': x6, 'betas': batch['betas'].float(), 'gt_j3d': batch['gt_j3d'].float(), 'cam': batch['cam'].float(),
               'gt_j2d': batch['gt_j2d'].float(), 'seed': seed * 1000 + it}


N_J = _engine.NUM_H36M * _engine.NUM_VERTS
N_SCALARS = 10     # joint, pose-discriminated, shape-discriminated, pose-D, shape-D, J error, MPJPE / PA-MPJPE before the
                   # J step (sums over the local poses) and the previous batch's MPJPE / PA-MPJPE after its J step


class SharedBucket:
    """The flat gradient + log bucket of one outer step: one buffer, one all-reduce, one read-back."""

    def __init__(self, device, use_pd: bool, use_sd: bool, hist_records: int):
        def pad(n):                # every section starts 256-byte aligned (the C ABI wants 16-byte aligned pointers)
            return (n + 63) // 64 * 64
        nD = _engine.DISC_PARAMS if use_pd else 0
        nS = _engine.SHAPE_DISC_PARAMS if use_sd else 0
        oD, oS = pad(N_J), pad(N_J) + pad(nD)
        oT = oS + pad(nS)
        self.flat = torch.zeros(oT + N_SCALARS + 5 * hist_records, device=device)
        self.dJ = self.flat[:N_J].view(_engine.NUM_H36M, _engine.NUM_VERTS)
        self.dD = self.flat[oD:oD + nD]
        self.dS = self.flat[oS:oS + nS]
        self.tail = self.flat[oT:]                                  # what is read back: scalars, then the history
        self.scalars = self.tail[:N_SCALARS]
        self.hist = self.tail[N_SCALARS:].view(-1, 5)
        self.nbytes = self.flat.numel() * 4

    def put(self, k: int, per_pose: torch.Tensor):
        self.scalars[k:k + 1].copy_(per_pose.sum().reshape(1))


def optimize_pose_refiner(log=print) -> Dict:
    dist = jdist.init(args.dist_backend)
    rank, local_rank, world = jdist.env_rank_world()
    device = torch.device(args.device if (world == 1 or args.single_device) else f'cuda:{local_rank}')
    torch.cuda.set_device(device)
    utils.set_seed(args.seed)

    smpl = SMPL(args.smpl_dir, batch_size=1, allow_synthetic=args.synthetic or args.smpl_dir == 'SPIN/data/smpl')              # :96-99
    J_np = smpl_model.default_h36m_regressor(args.j_regressor_init,
                                             allow_default=args.synthetic or args.j_regressor_init == 'SPIN/data/J_regressor_h36m.npy')
    # the body goes to the device with the regressor's positive columns as a hint for the library's internal vertex order: the
    # support is stored first, so the joint-loss iterations (FLAG_SUPPORT_TILES below) run a handful of tiles
    tiles_mode = not args.all_vertex_tiles and not args.silhouette
    smpl = smpl.to(device, hint_vertices=np.nonzero((J_np > 0).any(0))[0] if tiles_mode else None)
    J_regressor = torch.from_numpy(J_np).float().to(device).contiguous()                   # :105-107
    j_reg_mask = utils.find_j_reg_mask(J_regressor).contiguous()                           # :130

    use_pd, use_sd = not args.no_pose_disc, bool(args.shape_disc)
    pose_discriminator = Discriminator()                                                   # :112-113 (torch default init)
    shape_discriminator = Shape_Discriminator()                                            # :119-120
    disc_flat = pose_discriminator.flat_parameters().to(device).contiguous()
    sdisc_flat = shape_discriminator.flat_parameters().to(device).contiguous()
    disc_opt = AdamState(disc_flat, args.opt_disc_learning_rate)                          # :116-117
    sdisc_opt = AdamState(sdisc_flat, args.opt_disc_learning_rate)                        # :122-123
    J_opt = AdamState(J_regressor, args.j_reg_lr)                                          # :125-126

    flags = _engine.FLAG_KEEP_VERTS | (_engine.FLAG_POSE_DISC if use_pd else 0) | (_engine.FLAG_SHAPE_DISC if use_sd else 0) \
        | (_engine.FLAG_SILHOUETTE if args.silhouette else 0) | (0 if args.all_vertex_tiles else _engine.FLAG_SUPPORT_TILES)
    # FLAG_SUPPORT_TILES: iterations whose loss reads the joints only run their skinning kernels on the 32-vertex tiles that hold an
    # entry of the regressor's support (51 of 216 for the shipped checkpoint's structure); the other tiles meet a zero block of the
    # regressor (scripts/utils.py:87-92 multiplies them anyway) and a zero vertex adjoint.  Engaged by eng.j_support_info() below
    # when the support fits the device lists; with the silhouette term every vertex is needed and every tile runs.
    engines: Dict = {}
    n_hist = (args.inner_iters + 9) // 10                                                  # :255 `if i % 10 == 0`
    bucket = SharedBucket(device, use_pd, use_sd, n_hist)
    after_sums = torch.zeros(2, device=device)             # MPJPE / PA-MPJPE sums after the J step: next batch's bucket

    def engine_for(B_local: int, B_global: int):
        """one engine per shard size (the last batch of a dataset may be ragged: drop_last=False)"""
        if B_local not in engines:
            engines[B_local] = _engine.RefineEngine(smpl.device_model, B_local, batch_norm=B_global, flags=flags)
        eng = engines[B_local]
        eng.set_batch_norm(B_global)
        eng.set_j_regressor(J_regressor, j_reg_mask)
        eng.j_support_info()      # one small read-back per outer batch: an engine that knows its regressor's support fits the device
        #                           lists enqueues the support-restricted J-step products only (include/jrr.h, jrr_j_support_info)
        if use_pd:
            eng.set_pose_disc(disc_flat)
        if use_sd:
            eng.set_shape_disc(sdisc_flat)
        return eng

    if args.data_root:
        source = _dataset_batches(args.data_root, args.batch_size, args.seed, device)
    else:
        source = _synthetic_batches(smpl.model_np, J_np, args.batch_size, args.synthetic_batches, args.seed)

    history = []
    pending = None             # the previous batch's record, waiting for its after-the-J-step metrics
    x6d = betas = cam = None
    lo = hi = 0

    def finish(rec, B_global, after):
        """complete a record with the all-reduced MPJPE / PA-MPJPE of the stepped regressor and log it (:323-337)"""
        mpjpe_a, pampjpe_a = float(after[0]) * 1000 / B_global, float(after[1]) * 1000 / B_global
        rec['mpjpe'], rec['pampjpe'] = mpjpe_a, pampjpe_a
        rec['mpjpe difference'] = rec.pop('_mpjpe_before') - mpjpe_a
        rec['pampjpe difference'] = rec.pop('_pampjpe_before') - pampjpe_a
        history.append(rec)
        if rank == 0:
            log(rec)                                                                        # wandb.log analogue
            if args.wandb_log:
                try:
                    import wandb
                    wandb.log(rec)
                except ImportError:
                    pass

    for it, full in enumerate(source):                                                     # :144-148
        t_batch = time.perf_counter()
        B_global = int(full['pose6d'].shape[0])
        lo, hi = jdist.shard_bounds(B_global, rank, world)
        B = hi - lo
        if B == 0:
            raise RuntimeError(f'batch of {B_global} poses cannot be sharded over {world} ranks')
        eng = engine_for(B, B_global)
        spin_pose = full['pose6d'][lo:hi].to(device).float().contiguous()                   # :166-168,177-178
        spin_betas = full['betas'][lo:hi].to(device).float().contiguous()
        gt_mm = full['gt_j3d'][lo:hi].to(device).float().contiguous()
        gt_j3d = utils.move_pelvis(gt_mm).contiguous()                                     # :162
        x6d = spin_pose.clone()                                                            # :177-179 pose + orient
        betas = spin_betas.clone()
        m = torch.zeros(B, 154, device=device)                                             # :201-202 fresh optimizer
        v = torch.zeros(B, 154, device=device)
        step = torch.zeros(1, dtype=torch.int32, device=device)
        sq = torch.zeros(B, device=device)

        # ---- camera pre-fit + 2-D term (:170-173,187-199,231-233; row f1) ----
        cam = full['cam'][lo:hi].to(device).float().contiguous()                            # :170-172 pred_cam_t
        cam_m, cam_v = torch.zeros_like(cam), torch.zeros_like(cam)
        if args.reprojection:
            if 'gt_j2d' in full:
                gt_j2d = full['gt_j2d'][lo:hi].to(device).float().contiguous()
            else:
                gt_j2d = _synthetic_gt_j2d(eng, x6d, betas, cam, full['seed'], lo, hi, B_global)
            eng.camera_prefit(x6d, betas, gt_j2d, cam, n_steps=args.camera_iters, lr=1e-2)   # :187-199
            eng.set_reprojection(gt_j2d, cam, cam_m, cam_v)
        if args.silhouette:                                                                # :234-237 (row f2)
            sil_mask = _synthetic_mask(eng, x6d, betas, cam, full['seed'], lo, hi, B_global)
            eng.set_silhouette(sil_mask, cam, cam_m, cam_v)

        t0 = time.perf_counter()
        eng.set_loss_history(n_hist, 10)                                                   # :255-261 (read back with the bucket)
        # ---- the 100 inner iterations (:220-265); J steps inside the loop only when --j_step_every < --inner_iters ----
        n_inloop = (args.inner_iters - 1) // args.j_step_every * args.j_step_every if args.inner_iters > 0 else 0
        if n_inloop and dist is None:       # one C call for the iterations AND their J steps (no collective needed)
            eng.refine_run_j_steps(x6d, betas, gt_j3d, m, v, step, 1e-2, n_inloop, args.j_step_every, J_regressor, J_opt.m,
                                   J_opt.v, J_opt.step, J_opt.lr, mask=j_reg_mask, sqerr=sq)
        elif n_inloop:
            # the J step's all-reduce carries the regressor's support only (8.7 KB) when it fits the engine's lists, the dense
            # (17,6890) gradient otherwise or with --j_allreduce dense (asked once per batch: one small read-back)
            xch = jdist.JStepExchange(eng, bucket.dJ, compact=args.j_allreduce == 'support')
            for done in range(0, n_inloop, args.j_step_every):
                eng.refine_run(x6d, betas, gt_j3d, m, v, step, 1e-2, args.j_step_every, sqerr=sq, after_j_step=done > 0)
                xch.step(J_regressor, J_opt.m, J_opt.v, J_opt.step, J_opt.lr, x6d, betas, gt_j3d, mask=j_reg_mask)
        eng.refine_run(x6d, betas, gt_j3d, m, v, step, 1e-2, args.inner_iters - n_inloop, sqerr=sq, after_j_step=n_inloop > 0)
        tiles_run = eng.support_tiles()[1]          # vertex tiles the inner iterations ran (asked while the silhouette term is still set)
        sv_on, sv_n = eng.support_vertices()          # ... or, per support VERTEX, one launch per iteration (include/jrr.h)
        bucket.flat.zero_()
        bucket.put(0, sq)                                                                   # joint_loss (:238-239)
        pose_disc_sq, shape_disc_sq = eng.refine_aux_losses(use_pd, use_sd) if args.inner_iters > 0 else (None, None)
        if pose_disc_sq is not None:
            bucket.put(1, pose_disc_sq)                                                     # :246-247
        if shape_disc_sq is not None:
            bucket.put(2, shape_disc_sq)                                                    # :249-250
        hist = eng.loss_history()
        if hist is not None and hist.shape[0]:
            bucket.hist[:hist.shape[0]].copy_(hist)
        eng.set_loss_history(0)
        if args.reprojection:
            eng.set_reprojection(None)
        if args.silhouette:
            eng.set_silhouette(None)

        # ---- LOCAL gradients of the three shared-parameter steps, written straight into the bucket ----
        if use_pd:                                                                          # :276-284
            l0 = eng.pose_disc_backward_params(x6d, 0.0, bucket.dD)                        # MSE(D(opt.detach()), 0)
            l1 = eng.pose_disc_backward_params(spin_pose, 1.0, bucket.dD)                  # MSE(D(spin), 1)
            bucket.put(3, l0 + l1)
        if use_sd:                                                                          # :286-293
            l0 = eng.shape_disc_backward_params(betas, 0.0, bucket.dS)
            l1 = eng.shape_disc_backward_params(spin_betas, 1.0, bucket.dS)
            bucket.put(4, l0 + l1)
        jsq = torch.zeros(B, device=device)                                                 # :300-312
        joints_before = torch.empty(B, 17, 3, device=device)
        eng.j_regressor_grad(x6d, betas, gt_j3d, sqerr=jsq, out=bucket.dJ, joints=joints_before)
        bucket.put(5, jsq)
        e_b, epa_b = utils.evaluate_sums(joints_before, gt_mm)                             # :314-315 joints of the old regressor
        bucket.scalars[6:7].copy_(e_b.reshape(1)); bucket.scalars[7:8].copy_(epa_b.reshape(1))
        bucket.scalars[8:10].copy_(after_sums)                                              # previous batch, after its J step

        jdist.all_reduce_sum_(bucket.flat)                                                  # THE collective of the outer step

        # ---- replicated Adam steps (identical on every rank; the three parameter sets are independent: the J step goes first so that
        #      the joints after it come from the vertices its forward stored -- uploading discriminator weights drops that state) ----
        eng.j_step_apply(J_regressor, bucket.dJ, J_opt.m, J_opt.v, J_opt.step, J_opt.lr, mask=j_reg_mask)
        joints_after = eng.find_joints_after_j_step(betas, x6d)                            # :317-321 with the stepped regressor (re-regressed from the J step's vertices)
        e_a, epa_a = utils.evaluate_sums(joints_after, gt_mm)
        after_sums = torch.stack([e_a, epa_a])
        if use_pd:
            disc_opt.apply(disc_flat, bucket.dD)
            eng.set_pose_disc(disc_flat)
        if use_sd:
            sdisc_opt.apply(sdisc_flat, bucket.dS)
            eng.set_shape_disc(sdisc_flat)

        tail = bucket.tail.cpu().double().numpy()                                           # the ONE read-back of this batch
        sc, hist_np = tail[:N_SCALARS], tail[N_SCALARS:].reshape(-1, 5)
        if pending is not None:
            finish(pending[0], pending[1], sc[8:10])
        rec = {'batch': it, 'joint_loss': sc[0] / (B_global * 51),
               'pose_discriminated_loss': sc[1] / (B_global * 25) if pose_disc_sq is not None else None,
               'shape_discriminated_loss': sc[2] / B_global if shape_disc_sq is not None else None,
               'pose_discriminator_loss': sc[3] / (B_global * 25) if use_pd else None,
               'shape_discriminator_loss': sc[4] / B_global if use_sd else None,
               'j_regressor_error': sc[5] / (B_global * 51),
               '_mpjpe_before': sc[6] * 1000 / B_global, '_pampjpe_before': sc[7] * 1000 / B_global,
               'loss_history': [[float(x) for x in row] for row in hist_np],                # :255-261, every 10th iteration
               'seconds': time.perf_counter() - t0,                                         # inner loop + outer step, as the reference times nothing finer
               'seconds_batch': time.perf_counter() - t_batch,                              # + H->D copies, camera pre-fit, target set-up
               'vertex_tiles_run': tiles_run,     # 216, or the tiles of the regressor's support (FLAG_SUPPORT_TILES engaged)
               'support_vertices_run': sv_n if sv_on else None,      # the vertices the per-vertex iteration ran on (None: tile kernels)
               'body_model': smpl.provenance, 'data': 'dataset' if args.data_root else 'synthetic'}
        rec = {k: (float(x) if isinstance(x, np.floating) else x) for k, x in rec.items()}
        pending = (rec, B_global)

    if pending is not None:                 # flush: the last batch's after-the-J-step sums
        jdist.all_reduce_sum_(after_sums)
        finish(pending[0], pending[1], after_sums.cpu().double().numpy())
    if args.save_j_regressor and rank == 0:
        checkpoint.save_j_regressor(J_regressor, args.save_j_regressor)
    return {'history': history, 'J_regressor': J_regressor, 'disc_flat': disc_flat, 'sdisc_flat': sdisc_flat,
            'x6d': x6d, 'betas': betas, 'cam': cam, 'shard': (lo, hi)}


def _synthetic_gt_j2d(eng, x6d, betas, cam, seed, lo, hi, B_global):
    """2-D targets in the 224-crop pixel frame (scripts/data.py:134-138): the current joints seen through a
    perturbed camera, plus pixel noise (stand-in for the H36M annotations).  The noise is drawn for the GLOBAL
    batch and sliced, so a sharded run sees the same targets as the single-process run."""
    from . import renderer
    g = torch.Generator().manual_seed(seed)
    joints = eng.find_joints_forward(betas, x6d=x6d)
    dcam = (torch.randn(B_global, 3, generator=g) * torch.tensor([0.3, 0.3, 3.0]))[lo:hi]
    noise = (torch.randn(B_global, 17, 2, generator=g) * 2.0)[lo:hi]
    p = renderer.project_points(joints, cam + dcam.to(cam.device))[..., :2]
    return (p + noise.to(cam.device)).contiguous()


def _synthetic_mask(eng, x6d, betas, cam, seed, lo, hi, B_global):
    """Target silhouettes (stand-in for batch['mask_rcnn'], scripts/data.py): the current mesh seen through a
    perturbed camera, binarised."""
    g = torch.Generator().manual_seed(seed + 7)
    _, verts = eng.find_joints_forward(betas, x6d=x6d, return_verts=True)
    dcam = (torch.randn(B_global, 3, generator=g) * torch.tensor([0.2, 0.2, 2.0]))[lo:hi]
    cam_true = (cam + dcam.to(cam.device)).contiguous()
    return (eng.silhouette_forward(verts, cam_true) > 0).float().contiguous()
