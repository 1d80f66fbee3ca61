"""CPU restatement (PyTorch-CPU, dtype-generic) of the reference's pose-refinement hot path.

TEST INFRASTRUCTURE: the checker for the HIP path and the timed `cpu_baseline` of bench.py.
Nothing in the product package imports this file.

Parity pinning
--------------
* rot6d_to_rotmat, find_joints, move_pelvis, find_j_reg_mask, Discriminator,
  Shape_Discriminator, evaluate / Procrustes, torch Adam: PINNED against golden vectors
  captured by importing the reference's own modules (tests/golden/make_golden.py ->
  tests/golden/*.npz, checked by tests/test_oracle_golden.py).
* SMPL linear blend skinning (`smpl_lbs`): **parity unpinned**.  The arithmetic lives in the
  third-party `smplx==0.1.26` (reference requirements.txt:12; call sites scripts/utils.py:94-95,
  scripts/optimize.py:78-79, scripts/renderer.py:32-33), which is absent from the reference
  tree and from this image, as is the licence-gated SMPL model file.  `smpl_lbs` restates
  the published smplx `lbs()` algorithm (pose2rot=False branch) and
  is checked by analytic known-answer tests (tests/test_oracle_kat.py) instead.

Every function cites the reference file:line it follows.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
NUM_VERTS = 6890
NUM_JOINTS = 24
NUM_H36M = 17
NUM_BETAS = 10


# ----------------------------------------------------------------------------------------
# scripts/utils.py
# ----------------------------------------------------------------------------------------
def rot6d_to_rotmat(x: torch.Tensor) -> torch.Tensor:
    """scripts/utils.py:190-204.  (N*6,) / (N,6) -> (N,3,3); columns of R are b1,b2,b3.

    The reference's `torch.cross(b1, b2)` has no `dim` (utils.py:203): for (N,3) inputs with
    N != 3 that is the row-wise cross product, which is what is restated here for every N
    (the N == 3 behaviour of the legacy default is a reference bug).
    """
    x = x.reshape(-1, 3, 2)
    a1 = x[:, :, 0]
    a2 = x[:, :, 1]
    b1 = F.normalize(a1)  # eps=1e-12 clamp on the norm
    b2 = F.normalize(a2 - torch.einsum('bi,bi->b', b1, a2).unsqueeze(-1) * b1)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)


def find_j_reg_mask(j_reg: torch.Tensor) -> torch.Tensor:
    """scripts/utils.py:182-187.  Reproduces the reference's behaviour: `zeros` is built with
    torch.ones (utils.py:184), so the mask is identically 1."""
    ones = torch.ones_like(j_reg)
    zeros = torch.ones_like(j_reg)
    return torch.where(j_reg == 0, zeros, ones)


def normalize_j_regressor(J_regressor: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """scripts/utils.py:87-92: J*mask -> ReLU -> divide each row by its sum."""
    if mask is not None:
        J_regressor = J_regressor * mask
    Jn = torch.relu(J_regressor)
    return Jn / torch.sum(Jn, dim=1).unsqueeze(1).expand(Jn.shape)


def find_joints(smpl, shape, orient, pose, J_regressor, mask=None, return_verts=False):
    """scripts/utils.py:85-103 with `smpl` any callable returning an object with `.vertices`."""
    Jn = normalize_j_regressor(J_regressor, mask)
    pred_vertices = smpl(global_orient=orient, body_pose=pose, betas=shape, pose2rot=False).vertices
    Jb = Jn[None, :].expand(pred_vertices.shape[0], -1, -1)
    pred_joints = torch.matmul(Jb, pred_vertices)
    if return_verts:
        return pred_joints, pred_vertices
    return pred_joints


def move_pelvis(j3ds: torch.Tensor) -> torch.Tensor:
    """scripts/utils.py:106-114."""
    return j3ds - j3ds[:, [0], :]


def batch_compute_similarity_transform_torch(S1, S2):
    """scripts/eval_utils.py:7-58 (Procrustes alignment of S1 onto S2)."""
    transposed = False
    if S1.shape[0] != 3 and S1.shape[0] != 2:
        S1 = S1.permute(0, 2, 1)
        S2 = S2.permute(0, 2, 1)
        transposed = True
    mu1 = S1.mean(dim=-1, keepdim=True)
    mu2 = S2.mean(dim=-1, keepdim=True)
    X1 = S1 - mu1
    X2 = S2 - mu2
    var1 = torch.sum(X1 ** 2, dim=1).sum(dim=1)
    K = X1.bmm(X2.permute(0, 2, 1))
    U, s, Vh = torch.linalg.svd(K)
    V = Vh.transpose(1, 2)
    Z = torch.eye(U.shape[1], dtype=S1.dtype).unsqueeze(0).repeat(U.shape[0], 1, 1)
    Z[:, -1, -1] *= torch.sign(torch.det(U.bmm(V.permute(0, 2, 1))))
    R = V.bmm(Z.bmm(U.permute(0, 2, 1)))
    scale = torch.stack([torch.trace(x) for x in R.bmm(K)]) / var1
    t = mu2 - (scale.unsqueeze(-1).unsqueeze(-1) * (R.bmm(mu1)))
    S1_hat = scale.unsqueeze(-1).unsqueeze(-1) * R.bmm(S1) + t
    if transposed:
        S1_hat = S1_hat.permute(0, 2, 1)
    return S1_hat


def evaluate(pred_j3ds: torch.Tensor, target_j3ds: torch.Tensor):
    """scripts/utils.py:117-145: MPJPE and PA-MPJPE in mm (target in mm, pred in m)."""
    with torch.no_grad():
        pred = pred_j3ds.clone().detach()
        target = target_j3ds.clone().detach() / 1000
        pred = pred - pred[:, [0], :]
        target = target - target[:, [0], :]
        errors = torch.sqrt(((pred - target) ** 2).sum(dim=-1)).mean(dim=-1).numpy()
        S1_hat = batch_compute_similarity_transform_torch(pred, target)
        errors_pa = torch.sqrt(((S1_hat - target) ** 2).sum(dim=-1)).mean(dim=-1).numpy()
        return np.mean(errors) * 1000, np.mean(errors_pa) * 1000


# ----------------------------------------------------------------------------------------
# SMPL linear blend skinning (smplx 0.1.26 lbs(), pose2rot=False)  -- parity unpinned
# ----------------------------------------------------------------------------------------
class SMPLOutput:
    def __init__(self, vertices, joints=None):
        self.vertices = vertices
        self.joints = joints


def smpl_lbs(model: Dict[str, torch.Tensor], rotmats: torch.Tensor, betas: torch.Tensor,
             return_all: bool = False):
    """SURVEY.md Appendix A steps 1-5 (smplx lbs.py `lbs`, pose2rot=False branch).

    model: v_template (V,3), shapedirs (V,3,10), posedirs (207, V*3), J_regressor (24,V),
           lbs_weights (V,24), parents (24,)
    rotmats: (B,24,3,3); betas: (B,10).  Returns vertices (B,V,3) [and posed joints (B,24,3)].
    """
    B = rotmats.shape[0]
    dt = rotmats.dtype
    v_template = model['v_template'].to(dt)
    shapedirs = model['shapedirs'].to(dt)
    posedirs = model['posedirs'].to(dt)
    Jreg = model['J_regressor'].to(dt)
    W = model['lbs_weights'].to(dt)
    parents = [int(p) for p in model['parents']]
    # 1. shape blend shapes
    v_shaped = v_template[None] + torch.einsum('bl,mkl->bmk', betas, shapedirs)
    # 2. rest joints
    J = torch.einsum('bik,ji->bjk', v_shaped, Jreg)
    # 3. pose blend shapes
    ident = torch.eye(3, dtype=dt)
    pose_feature = (rotmats[:, 1:, :, :] - ident).reshape(B, -1)
    v_posed = v_shaped + torch.matmul(pose_feature, posedirs).view(B, -1, 3)
    # 4. rigid chain (smplx batch_rigid_transform)
    rel = J.clone()
    rel[:, 1:] = J[:, 1:] - J[:, parents[1:]]
    T_local = torch.zeros(B, NUM_JOINTS, 4, 4, dtype=dt)
    T_local[:, :, :3, :3] = rotmats
    T_local[:, :, :3, 3] = rel
    T_local[:, :, 3, 3] = 1
    chain = [T_local[:, 0]]
    for i in range(1, NUM_JOINTS):
        chain.append(torch.matmul(chain[parents[i]], T_local[:, i]))
    G = torch.stack(chain, dim=1)
    posed_joints = G[:, :, :3, 3]
    J_h = torch.cat([J, torch.zeros(B, NUM_JOINTS, 1, dtype=dt)], dim=2).unsqueeze(-1)
    A = G - F.pad(torch.matmul(G, J_h), [3, 0, 0, 0, 0, 0, 0, 0])
    # 5. skinning
    T = torch.matmul(W[None].expand(B, -1, -1), A.view(B, NUM_JOINTS, 16)).view(B, -1, 4, 4)
    v_h = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=dt)], dim=2)
    verts = torch.matmul(T, v_h.unsqueeze(-1))[:, :, :3, 0]
    if return_all:
        return verts, posed_joints, dict(v_shaped=v_shaped, J=J, v_posed=v_posed, A=A, T=T)
    return verts, posed_joints


class OracleSMPL:
    """Callable with the reference's SMPL operator convention (scripts/utils.py:94-95,
    scripts/optimize.py:96-99): smpl(global_orient=(B,1,3,3), body_pose=(B,23,3,3),
    betas=(B,10), pose2rot=False).vertices -> (B,6890,3)."""

    def __init__(self, model: Dict[str, np.ndarray], dtype=torch.float32):
        self.model = {k: (torch.as_tensor(np.asarray(v)).to(dtype) if k != 'parents' else
                          torch.as_tensor(np.asarray(v)).long()) for k, v in model.items()
                      if k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights', 'parents')}
        self.dtype = dtype

    def __call__(self, global_orient, body_pose, betas, pose2rot=False):
        assert not pose2rot, "hot path always passes rotation matrices (scripts/utils.py:94-95)"
        R = torch.cat([global_orient, body_pose], dim=1)
        verts, joints = smpl_lbs(self.model, R, betas)
        return SMPLOutput(verts, joints)


def rodrigues(aa: torch.Tensor) -> torch.Tensor:
    """Axis-angle (N,3) -> (N,3,3) (smplx batch_rodrigues; not on the reference's hot path,
    SURVEY.md fact 4; used only to synthesise SPIN-like initial poses)."""
    angle = torch.norm(aa + 1e-8, dim=1, keepdim=True)
    axis = aa / angle
    c = torch.cos(angle)[:, None]
    s = torch.sin(angle)[:, None]
    rx, ry, rz = axis[:, 0], axis[:, 1], axis[:, 2]
    z = torch.zeros_like(rx)
    K = torch.stack([z, -rz, ry, rz, z, -rx, -ry, rx, z], dim=1).view(-1, 3, 3)
    I = torch.eye(3, dtype=aa.dtype)[None]
    return I + s * K + (1 - c) * torch.bmm(K, K)


# ----------------------------------------------------------------------------------------
# scripts/renderer.py:10-51 (return_2d_joints) -- pytorch3d 0.3.0 camera maths, parity unpinned
# ----------------------------------------------------------------------------------------
def project_joints(point_cloud: torch.Tensor, cam: torch.Tensor) -> torch.Tensor:
    """renderer.py:35-49 with PerspectiveCameras(T=cam, R=I, focal=5000/224, principal point 0) and
    transform_points_screen at 224x224 (pytorch3d is absent: parity unpinned).
    point_cloud (B,N,3), cam (B,3) -> screen xy (B,N,2)."""
    f, W = 5000.0 / 224.0, 224.0
    X = -2 * point_cloud[..., 0] + cam[:, None, 0]
    Y = -2 * point_cloud[..., 1] + cam[:, None, 1]
    Z = 2 * point_cloud[..., 2] + cam[:, None, 2]
    xs = (W - 1) / 2 * (1 - f * X / Z)
    ys = (W - 1) / 2 * (1 - f * Y / Z)
    return torch.stack([xs, ys], dim=-1)


def camera_prefit(joints: torch.Tensor, gt_j2d: torch.Tensor, cam: torch.Tensor, n_steps: int, lr: float = 1e-2,
                  batch_norm: Optional[int] = None) -> torch.Tensor:
    """scripts/optimize.py:187-199: Adam([cam], lr) against MSE(gt_j2d, joints_2d); joints are constant."""
    cam = cam.clone().detach().requires_grad_(True)
    nb = joints.shape[0] if batch_norm is None else batch_norm
    opt = torch.optim.Adam([cam], lr=lr)
    for _ in range(n_steps):
        loss = ((gt_j2d - project_joints(joints.detach(), cam)) ** 2).sum() / (nb * NUM_H36M * 2)
        opt.zero_grad()
        loss.backward()
        opt.step()
    return cam.detach()


# ----------------------------------------------------------------------------------------
# scripts/discriminator.py (functional restatement over a state_dict)
# ----------------------------------------------------------------------------------------
def discriminator_forward(sd: Dict[str, torch.Tensor], rot6d: torch.Tensor) -> torch.Tensor:
    """scripts/discriminator.py:32-54.  rot6d (B,24,6) -> (B,25,1), sigmoid outputs.
    Output 0 = global MLP, outputs 1..24 = per-joint heads.  Flatten index = joint*32 + channel."""
    B = rot6d.shape[0]
    x = rot6d.permute(0, 2, 1).unsqueeze(-1)                       # (B,6,24,1)
    h = F.relu(F.conv2d(x, sd['conv_operations.0.weight'], sd['conv_operations.0.bias']))
    h = F.relu(F.conv2d(h, sd['conv_operations.2.weight'], sd['conv_operations.2.bias']))
    conv = h.permute(0, 2, 1, 3)                                   # (B,24,32,1)
    g = conv.reshape(-1, 24 * 32)
    g = F.relu(F.linear(g, sd['linear_operations.0.weight'], sd['linear_operations.0.bias']))
    g = F.relu(F.linear(g, sd['linear_operations.2.weight'], sd['linear_operations.2.bias']))
    g = F.linear(g, sd['linear_operations.4.weight'], sd['linear_operations.4.bias'])
    preds = [g]
    for i in range(24):
        preds.append(F.linear(conv[:, i].reshape(-1, 32), sd[f'linears.{i}.weight'], sd[f'linears.{i}.bias']))
    return torch.sigmoid(torch.stack(preds, dim=1))


def shape_discriminator_forward(sd: Dict[str, torch.Tensor], betas: torch.Tensor) -> torch.Tensor:
    """scripts/discriminator.py:70-74."""
    h = F.relu(F.linear(betas, sd['shape_operations.0.weight'], sd['shape_operations.0.bias']))
    h = F.relu(F.linear(h, sd['shape_operations.2.weight'], sd['shape_operations.2.bias']))
    return torch.sigmoid(F.linear(h, sd['shape_operations.4.weight'], sd['shape_operations.4.bias']))


DISC_PARAM_SHAPES = (
    [('conv_operations.0.weight', (32, 6, 1, 1)), ('conv_operations.0.bias', (32,)),
     ('conv_operations.2.weight', (32, 32, 1, 1)), ('conv_operations.2.bias', (32,))]
    + [p for i in range(24) for p in ((f'linears.{i}.weight', (1, 32)), (f'linears.{i}.bias', (1,)))]
    + [('linear_operations.0.weight', (1024, 768)), ('linear_operations.0.bias', (1024,)),
       ('linear_operations.2.weight', (1024, 1024)), ('linear_operations.2.bias', (1024,)),
       ('linear_operations.4.weight', (1, 1024)), ('linear_operations.4.bias', (1,))])

SHAPE_DISC_PARAM_SHAPES = [
    ('shape_operations.0.weight', (10, 10)), ('shape_operations.0.bias', (10,)),
    ('shape_operations.2.weight', (5, 10)), ('shape_operations.2.bias', (5,)),
    ('shape_operations.4.weight', (1, 5)), ('shape_operations.4.bias', (1,))]


def formula_state_dict(shapes, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Deterministic weights reproducible without any RNG:
    w[n] = scale * sin(0.37*n + 1.3*seed + 0.11*idx_of_tensor), scale = 1/sqrt(fan_in)."""
    sd = {}
    for t, (name, shp) in enumerate(shapes):
        n = int(np.prod(shp))
        fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else int(shp[0])
        idx = np.arange(n, dtype=np.float64)
        w = np.sin(0.37 * idx + 1.3 * seed + 0.11 * t) / math.sqrt(max(fan_in, 1))
        if name.endswith('bias'):
            w = 0.1 * np.sin(0.73 * idx + 0.5 * seed + 0.07 * t)
        sd[name] = torch.as_tensor(w.reshape(shp)).to(dtype)
    return sd


# ----------------------------------------------------------------------------------------
# torch.optim.Adam single-tensor formula (torch/optim/adam.py, defaults used at
# scripts/optimize.py:116-126,187,201: betas=(0.9,0.999), eps=1e-8, no weight decay/amsgrad)
# ----------------------------------------------------------------------------------------
def adam_step(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8):
    """One in-place Adam update in torch's operation order; `step` is the 1-based step count."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    step_size = lr / bc1
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-step_size)
    return p


# ----------------------------------------------------------------------------------------
# scripts/optimize.py inner loop (:220-265) restricted to the BASELINE config's loss terms,
# and the outer-step updates (:276-312)
# ----------------------------------------------------------------------------------------
W_JOINT = 10000.0   # scripts/optimize.py:252
W_POSE_D = 10.0     # scripts/optimize.py:253
W_SHAPE_D = 10.0    # scripts/optimize.py:253


W_J2D = 0.01       # scripts/optimize.py:252  (loss_j2d/100)
W_SIL = 100.0      # scripts/optimize.py:252  (silhouette_loss*100)


def inner_losses(smpl, J_regressor, mask, orient6d, pose6d, betas, gt_j3d_mm_centred,
                 disc_sd=None, shape_disc_sd=None, batch_norm: Optional[int] = None, smpl_evals: int = 1,
                 gt_j2d=None, cam=None, sil_mask=None, faces=None):
    """One evaluation of the inner-loop objective (scripts/optimize.py:222-253) without the
    2-D and silhouette terms (BASELINE configs 2-4).  Returns (opt_loss, dict of terms, joints).

    `batch_norm`: divisor batch size for the MSE means (defaults to the local batch; the
    data-parallel build passes the global batch so a sharded run equals the single-process run).
    `smpl_evals`: 1 = deduplicated; 3 = reference-faithful redundant SMPL evaluations
    (scripts/optimize.py:228,231,234) for the CPU baseline's second variant.
    """
    B = orient6d.shape[0]
    nb = B if batch_norm is None else batch_norm
    R_orient = rot6d_to_rotmat(orient6d.reshape(-1, 6)).view(-1, 1, 3, 3)     # :222-223
    R_pose = rot6d_to_rotmat(pose6d.reshape(-1, 6)).view(-1, 23, 3, 3)        # :225-226
    pred_joints, pred_verts = find_joints(smpl, betas, R_orient, R_pose, J_regressor, mask=mask, return_verts=True)   # :228-229
    for _ in range(smpl_evals - 1):     # the reference's 2nd/3rd SMPL forward on identical inputs
        _ = find_joints(smpl, betas, R_orient, R_pose, J_regressor)
    diff = move_pelvis(pred_joints) - gt_j3d_mm_centred / 1000                # :238-239
    joint_loss = (diff ** 2).sum() / (nb * NUM_H36M * 3)
    terms = {'joint_loss': joint_loss}
    opt_loss = joint_loss * W_JOINT
    if gt_j2d is not None:                                                    # :231-233 (mask=None == mask of ones)
        loss_j2d = ((gt_j2d - project_joints(pred_joints, cam)) ** 2).sum() / (nb * NUM_H36M * 2)
        terms['loss_j2d'] = loss_j2d
        opt_loss = opt_loss + loss_j2d * W_J2D
    if sil_mask is not None:                                                  # :234-237 render_mesh + MSE, weight 100
        from . import silhouette_port
        sil_loss, _ = silhouette_port.silhouette_loss(pred_verts, faces, cam, sil_mask, batch_norm=nb)
        terms['silhouette_loss'] = sil_loss
        opt_loss = opt_loss + sil_loss * W_SIL
    if disc_sd is not None:
        pred_disc = discriminator_forward(disc_sd, torch.cat([orient6d, pose6d], dim=1))   # :241-242
        pose_d = ((pred_disc - 1) ** 2).sum() / (nb * 25)                     # :246-247
        terms['pose_discriminated_loss'] = pose_d
        opt_loss = opt_loss + pose_d * W_POSE_D
    if shape_disc_sd is not None:
        pred_s = shape_discriminator_forward(shape_disc_sd, betas)            # :244
        shape_d = ((pred_s - 1) ** 2).sum() / (nb * 1)                        # :249-250
        terms['shape_discriminated_loss'] = shape_d
        opt_loss = opt_loss + shape_d * W_SHAPE_D
    return opt_loss, terms, pred_joints


def refine_poses(smpl, J_regressor, orient6d, pose6d, betas, gt_j3d_mm_centred, n_iters: int,
                 disc_sd=None, shape_disc_sd=None, lr: float = 1e-2, mask=None,
                 batch_norm: Optional[int] = None, smpl_evals: int = 1, record=None, gt_j2d=None, cam=None,
                 sil_mask=None, faces=None):
    """scripts/optimize.py:201-202,220-265: fresh torch Adam over [pose, orient, betas] (cam has
    no gradient in configs 2-4 and is skipped by torch Adam), n_iters inner iterations.
    Inputs are cloned; returns the refined (orient6d, pose6d, betas) and the loss history."""
    orient = orient6d.clone().detach().requires_grad_(True)
    pose = pose6d.clone().detach().requires_grad_(True)
    b = betas.clone().detach().requires_grad_(True)
    if mask is None:
        mask = find_j_reg_mask(J_regressor)
    params = [pose, orient, b]
    c = None
    if gt_j2d is not None or sil_mask is not None:
        c = cam.clone().detach().requires_grad_(True)
        params.append(c)                                                      # :180-181,201-202
    opt = torch.optim.Adam(params, lr=lr)
    hist = []
    for it in range(n_iters):
        loss, terms, joints = inner_losses(smpl, J_regressor.detach(), mask, orient, pose, b, gt_j3d_mm_centred,
                                           disc_sd, shape_disc_sd, batch_norm, smpl_evals, gt_j2d, c, sil_mask, faces)
        opt.zero_grad()
        loss.backward()
        if record is not None:
            record(it, dict(loss=loss.detach().clone(), joints=joints.detach().clone(),
                            g_orient=orient.grad.clone(), g_pose=pose.grad.clone(), g_betas=b.grad.clone(),
                            **{k: v.detach().clone() for k, v in terms.items()}))
        opt.step()
        hist.append({"loss": float(loss.detach()), **{k: float(v.detach()) for k, v in terms.items()}})
    if c is not None:
        return orient.detach(), pose.detach(), b.detach(), hist, c.detach()
    return orient.detach(), pose.detach(), b.detach(), hist


def j_regressor_loss_and_grad(smpl, J_regressor, orient6d, pose6d, betas, gt_j3d_mm_centred,
                              mask=None, batch_norm: Optional[int] = None):
    """scripts/optimize.py:300-309: joint MSE (weight 1, no 10000 factor) of detached poses
    w.r.t. the raw J_regressor.  Returns (loss, dL/dJ, joints)."""
    J = J_regressor.clone().detach().requires_grad_(True)
    if mask is None:
        mask = find_j_reg_mask(J)
    B = orient6d.shape[0]
    nb = B if batch_norm is None else batch_norm
    R_orient = rot6d_to_rotmat(orient6d.detach().reshape(-1, 6)).view(-1, 1, 3, 3)
    R_pose = rot6d_to_rotmat(pose6d.detach().reshape(-1, 6)).view(-1, 23, 3, 3)
    joints = find_joints(smpl, betas.detach(), R_orient, R_pose, J, mask=mask)
    loss = ((move_pelvis(joints) - gt_j3d_mm_centred / 1000) ** 2).sum() / (nb * NUM_H36M * 3)
    loss.backward()
    return loss.detach(), J.grad.detach(), joints.detach()


def discriminator_update_loss_and_grads(disc_sd, opt6d_detached, spin6d, batch_norm: Optional[int] = None):
    """scripts/optimize.py:276-284: MSE(D(opt.detach()),0) + MSE(D(spin),1); returns (loss, grads dict)."""
    sd = {k: v.clone().detach().requires_grad_(True) for k, v in disc_sd.items()}
    nb = opt6d_detached.shape[0] if batch_norm is None else batch_norm
    pred_gt = discriminator_forward(sd, spin6d)
    pred_disc = discriminator_forward(sd, opt6d_detached.detach())
    loss = (pred_disc ** 2).sum() / (nb * 25) + ((pred_gt - 1) ** 2).sum() / (nb * 25)
    loss.backward()
    return loss.detach(), {k: v.grad.detach() for k, v in sd.items()}


__all__ = [n for n in dir() if not n.startswith('_')]
