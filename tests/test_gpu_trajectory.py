"""Trajectory parity at the reference's loop length (pytest -m gpu).

scripts/optimize.py:201-202,220-265 runs 100 Adam iterations per outer batch.  `north_star` asks for regressed 3-D joints within
1e-4 m of the reference on identical inputs: checked here AFTER the whole loop, against `oracle.refine_poses` (fresh torch Adam, the
reference's own statements), for

  * the joint loss alone (BASELINE configs[1]; the exact-fp32 engine's leg of the former test_gpu_bf16x3.py trajectory test),
  * BASELINE configs[2] (3-D joint loss + pose-discriminator term), B = 128, in both tile modes of the engine -- all 216 vertex tiles
    (what bench.py's `value` runs) and the regressor's support (what optimize.py runs by default: the per-vertex iteration of supk.h
    when the support has <= 64 vertices, else the tile lists) -- with the logged terms (optimize.py:255-261) within 5e-3,
  * the same with a J step (optimize.py:300-312) after every 10th iteration inside the loop: the regressor's 62 positive entries
    within 2e-5 of an oracle loop that keeps torch's Adam for the poses and `oracle.adam_step` for J across the J steps.
"""
import importlib

import numpy as np
import pytest
import torch

import oracle
from conftest import PKG_NAME, support_tiles_available

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = 'cuda:0'
N_ITERS = 100


@pytest.fixture(scope='module')
def em():
    return importlib.import_module(PKG_NAME + '.engine')


@pytest.fixture(scope='module')
def sm():
    return importlib.import_module(PKG_NAME + '.smpl_model')


@pytest.fixture(scope='module')
def dmodels(em, smpl_model_np, j_h36m_np):
    """the body as optimize.py uploads it (vertex-order hint = the regressor's positive columns) and without the hint"""
    hint = np.nonzero((j_h36m_np > 0).any(0))[0]
    return {'hinted': em.DeviceModel(smpl_model_np, DEV, hint_vertices=hint), 'plain': em.DeviceModel(smpl_model_np, DEV)}


def _state(B):
    return (torch.zeros(B, 154, device=DEV), torch.zeros(B, 154, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV))


def _oracle_joints(smpl, J, o, p, b):
    B = o.shape[0]
    R = oracle.rot6d_to_rotmat(torch.cat([o, p], 1).reshape(-1, 6)).view(B, 24, 3, 3)
    return oracle.find_joints(smpl, b, R[:, :1], R[:, 1:], J, mask=oracle.find_j_reg_mask(J))


def test_hundred_iterations_joint_loss_vs_oracle(em, sm, dmodels, smpl_model_np, j_h36m_np):
    """BASELINE configs[1]'s loss (3-D joints only), 100 fused iterations, all vertex tiles: joints of the refined poses within
    `north_star`'s 1e-4 m of the oracle's, the last logged joint loss within 5e-3"""
    B = 64
    batch = sm.synthetic_batch(smpl_model_np, j_h36m_np, B, seed=31)
    x6d, betas = T(batch['pose6d']), T(batch['betas'])
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    eng = em.RefineEngine(dmodels['plain'], B, flags=em.FLAG_KEEP_VERTS)
    eng.set_j_regressor(T(j_h36m_np))
    xd, bd = x6d.clone().to(DEV), betas.clone().to(DEV)
    m, v, step = _state(B)
    sq = torch.zeros(B, device=DEV)
    eng.refine_run(xd, bd, gt_c.to(DEV).contiguous(), m, v, step, 1e-2, N_ITERS, sqerr=sq)
    joints = eng.find_joints_forward(bd, x6d=xd).cpu()
    smpl = oracle.OracleSMPL(smpl_model_np)
    o, p, b, hist = oracle.refine_poses(smpl, T(j_h36m_np), x6d[:, :1], x6d[:, 1:], betas, gt_c, N_ITERS)
    dj = (joints - _oracle_joints(smpl, T(j_h36m_np), o, p, b)).abs().max().item()
    assert dj < 1e-4, dj
    np.testing.assert_allclose(float(sq.sum()) / (B * 51), hist[-1]['joint_loss'], rtol=5e-3)


# ---- BASELINE configs[2]: joint loss + pose discriminator --------------------------------------------------------------------
B2 = 128


@pytest.fixture(scope='module')
def config2_oracle(sm, smpl_model_np, j_h36m_np):
    """100 oracle iterations of configs[2] on 128 poses (run once for both tile modes)"""
    batch = sm.synthetic_batch(smpl_model_np, j_h36m_np, B2, seed=131)
    x6d, betas = T(batch['pose6d']), T(batch['betas'])
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    dsd = oracle.formula_state_dict(oracle.DISC_PARAM_SHAPES, seed=0)
    smpl = oracle.OracleSMPL(smpl_model_np)
    o, p, b, hist = oracle.refine_poses(smpl, T(j_h36m_np), x6d[:, :1], x6d[:, 1:], betas, gt_c, N_ITERS, disc_sd=dsd)
    return dict(x6d=x6d, betas=betas, gt_c=gt_c, dsd=dsd, smpl=smpl, o=o, p=p, b=b, hist=hist,
                joints=_oracle_joints(smpl, T(j_h36m_np), o, p, b))


def _engine(em, dm, B, tiles, dsd, J):
    eng = em.RefineEngine(dm, B, flags=em.FLAG_KEEP_VERTS | em.FLAG_POSE_DISC | (em.FLAG_SUPPORT_TILES if tiles else 0))
    eng.set_j_regressor(J)
    eng.set_pose_disc(em.flatten_state_dict(dsd, em.DISC_KEYS))
    counts, fits = eng.j_support_info()
    assert fits
    if tiles:
        assert eng.support_tiles()[0] == support_tiles_available()
    else:
        assert eng.support_tiles() == (False, 216)
    return eng


@pytest.mark.parametrize('tiles', [False, True], ids=['all_tiles', 'support'])
def test_hundred_iterations_joint_and_pose_disc_vs_oracle(em, dmodels, j_h36m_np, config2_oracle, tiles):
    """the benchmarked workload followed for the reference's 100 iterations: regressed joints < 1e-4 m (`north_star`), the logged joint
    and pose-discriminator terms of every 10th iteration within 5e-3 of the oracle's"""
    c = config2_oracle
    eng = _engine(em, dmodels['hinted'], B2, tiles, c['dsd'], T(j_h36m_np))
    eng.set_loss_history(N_ITERS // 10, 10)
    xd, bd = c['x6d'].clone().to(DEV), c['betas'].clone().to(DEV)
    m, v, step = _state(B2)
    eng.refine_run(xd, bd, c['gt_c'].to(DEV).contiguous(), m, v, step, 1e-2, N_ITERS)
    rec = eng.loss_history().cpu()
    assert rec.shape == (N_ITERS // 10, 5)
    for k in range(N_ITERS // 10):
        h = c['hist'][10 * k]
        np.testing.assert_allclose(rec[k, 2].item(), h['joint_loss'] * 10000, rtol=5e-3, err_msg=f'joint term, iteration {10 * k}')
        np.testing.assert_allclose(rec[k, 3].item(), h['pose_discriminated_loss'] * 10, rtol=5e-3, err_msg=f'pose-D term, iteration {10 * k}')
    eng2 = em.RefineEngine(dmodels['plain'], B2, flags=0)          # the joints of the refined poses: a fresh dense forward
    eng2.set_j_regressor(T(j_h36m_np))
    joints = eng2.find_joints_forward(bd, x6d=xd).cpu()
    dj = (joints - c['joints']).abs().max().item()
    assert dj < 1e-4, dj
    d = (xd.cpu() - torch.cat([c['o'], c['p']], 1)).abs()
    assert d.mean().item() < 2e-5, d.mean().item()


@pytest.fixture(scope='module')
def config2_oracle_j_steps(sm, smpl_model_np, j_h36m_np):
    """the same loop with the J step of scripts/optimize.py:300-312 after every 10th iteration: torch Adam for the poses kept across the
    J steps (one optimizer per outer batch, optimize.py:201-202), oracle.adam_step for the raw regressor (lr = args.j_reg_lr = 1e-2)"""
    batch = sm.synthetic_batch(smpl_model_np, j_h36m_np, B2, seed=132)
    x6d, betas = T(batch['pose6d']), T(batch['betas'])
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    dsd = oracle.formula_state_dict(oracle.DISC_PARAM_SHAPES, seed=0)
    smpl = oracle.OracleSMPL(smpl_model_np)
    J = T(j_h36m_np).clone()
    mask = oracle.find_j_reg_mask(J)
    orient = x6d[:, :1].clone().requires_grad_(True)
    pose = x6d[:, 1:].clone().requires_grad_(True)
    b = betas.clone().requires_grad_(True)
    opt = torch.optim.Adam([pose, orient, b], lr=1e-2)
    Jm, Jv = torch.zeros_like(J), torch.zeros_like(J)
    n_j = 0
    for it in range(N_ITERS):
        loss, _, _ = oracle.inner_losses(smpl, J, mask, orient, pose, b, gt_c, dsd)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if (it + 1) % 10 == 0:
            _, dJ, _ = oracle.j_regressor_loss_and_grad(smpl, J, orient.detach(), pose.detach(), b.detach(), gt_c, mask=mask)
            n_j += 1
            oracle.adam_step(J, dJ, Jm, Jv, n_j, 1e-2)          # in place
    o, p, bb = orient.detach(), pose.detach(), b.detach()
    return dict(x6d=x6d, betas=betas, gt_c=gt_c, dsd=dsd, smpl=smpl, o=o, p=p, b=bb, J=J, n_j=n_j,
                joints=_oracle_joints(smpl, J, o, p, bb))


@pytest.mark.parametrize('tiles', [False, True], ids=['all_tiles', 'support'])
def test_hundred_iterations_with_j_steps_vs_oracle(em, dmodels, j_h36m_np, config2_oracle_j_steps, tiles):
    c = config2_oracle_j_steps
    J = T(j_h36m_np).to(DEV).clone()
    eng = _engine(em, dmodels['hinted'], B2, tiles, c['dsd'], J)
    Jm, Jv, Js = torch.zeros_like(J), torch.zeros_like(J), torch.zeros(1, dtype=torch.int32, device=DEV)
    xd, bd = c['x6d'].clone().to(DEV), c['betas'].clone().to(DEV)
    m, v, step = _state(B2)
    eng.refine_run_j_steps(xd, bd, c['gt_c'].to(DEV).contiguous(), m, v, step, 1e-2, N_ITERS, 10, J, Jm, Jv, Js, 1e-2)
    assert int(Js.item()) == c['n_j'] == N_ITERS // 10
    J0 = T(j_h36m_np)
    moved = J.cpu() != J0
    assert torch.equal(moved, c['J'] != J0) and int(moved.sum()) == int((J0 > 0).sum())      # exactly the positive entries (62) moved
    dJ = (J.cpu() - c['J']).abs().max().item()
    assert dJ < 2e-5, dJ
    eng2 = em.RefineEngine(dmodels['plain'], B2, flags=0)
    eng2.set_j_regressor(J)
    joints = eng2.find_joints_forward(bd, x6d=xd).cpu()
    dj = (joints - c['joints']).abs().max().item()
    assert dj < 1e-4, dj


def test_support_iteration_with_2d_term_and_pose_disc_vs_oracle(em, sm, dmodels, smpl_model_np, j_h36m_np):
    """the per-vertex iteration with the 2-D reprojection term (scripts/optimize.py:231-233: un-centred joints through the camera, weight
    1/100; the camera translation is a parameter of the same Adam) and the pose discriminator: 10 iterations against the oracle"""
    B, n = 48, 10
    batch = sm.synthetic_batch(smpl_model_np, j_h36m_np, B, seed=47)
    x6, betas, cam0 = T(batch['pose6d']), T(batch['betas']), T(batch['cam'])
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    dsd = oracle.formula_state_dict(oracle.DISC_PARAM_SHAPES, seed=0)
    smpl = oracle.OracleSMPL(smpl_model_np)
    R = oracle.rot6d_to_rotmat(x6.reshape(-1, 6)).view(B, 24, 3, 3)
    j0 = oracle.find_joints(smpl, betas, R[:, :1], R[:, 1:], T(j_h36m_np))
    gen = torch.Generator().manual_seed(5)
    cam_true = torch.tensor([0.0, 0.0, 2 * 5000 / (224 * 0.9)]).repeat(B, 1) + torch.randn(B, 3, generator=gen) * torch.tensor([0.3, 0.3, 3.0])
    gt2d = oracle.project_joints(j0, cam_true) + torch.randn(B, 17, 2, generator=gen) * 2.0
    o, p, b, hist, c = oracle.refine_poses(smpl, T(j_h36m_np), x6[:, :1], x6[:, 1:], betas, gt_c, n, disc_sd=dsd, gt_j2d=gt2d, cam=cam0)
    eng = _engine(em, dmodels['hinted'], B, True, dsd, T(j_h36m_np))
    xd, bd, cd = x6.clone().to(DEV), betas.clone().to(DEV), cam0.clone().to(DEV)
    cm, cv = torch.zeros(B, 3, device=DEV), torch.zeros(B, 3, device=DEV)
    eng.set_reprojection(gt2d.to(DEV).contiguous(), cd, cm, cv)
    if support_tiles_available():
        import os
        assert eng.support_vertices()[0] == (os.environ.get('JRR_SUPPORT_FUSED') != '0')      # the 2-D term does not leave the per-vertex iteration
    eng.set_loss_history(n, 1)
    m, v, step = _state(B)
    eng.refine_run(xd, bd, gt_c.to(DEV).contiguous(), m, v, step, 1e-2, n)
    rec = eng.loss_history().cpu()
    eng.set_reprojection(None)
    for k in range(n):
        np.testing.assert_allclose(rec[k, 0].item(), hist[k]['loss_j2d'] * 0.01, rtol=5e-3, err_msg=f'2-D term, iteration {k}')
        np.testing.assert_allclose(rec[k, 2].item(), hist[k]['joint_loss'] * 10000, rtol=5e-3, err_msg=f'joint term, iteration {k}')
    assert (xd.cpu() - torch.cat([o, p], 1)).abs().max().item() < 6e-4
    assert (xd.cpu() - torch.cat([o, p], 1)).abs().mean().item() < 5e-6
    assert (bd.cpu() - b).abs().max().item() < 3e-4
    assert (cd.cpu() - c).abs().max().item() < 3e-4
    assert (cd.cpu() - cam0).abs().max().item() > 1e-2           # the camera really moved
