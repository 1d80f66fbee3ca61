"""The host-side mirror of the reference's operator interface (SMPL / find_joints / Discriminator /
optimize_pose_refiner) and the outer-step updates, HIP path vs the CPU oracle.  pytest -m gpu."""
import importlib
import sys

import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden, PKG_NAME

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = 'cuda:0'


def _mod(name):
    return importlib.import_module(f'{PKG_NAME}.{name}')


def relerr(a, b):
    return ((a.detach().cpu().double() - b.detach().double()).abs().max() / b.detach().abs().max().clamp_min(1e-30)).item()


@pytest.fixture(scope='module')
def smpl_hip(smpl_model_np):
    return _mod('smpl').SMPL(model=smpl_model_np).to(DEV)


def test_smpl_operator_vertices_and_adjoint(smpl_hip, smpl_model_np):
    B = 5
    gen = torch.Generator().manual_seed(2)
    R = oracle.rodrigues(torch.randn(B * 24, 3, generator=gen, dtype=torch.float64) * 0.4).view(B, 24, 3, 3)
    betas = torch.randn(B, 10, generator=gen, dtype=torch.float64)
    dv = torch.randn(B, 6890, 3, generator=gen, dtype=torch.float64)
    Rr, br = R.clone().requires_grad_(True), betas.clone().requires_grad_(True)
    ref = oracle.OracleSMPL(smpl_model_np, dtype=torch.float64)(Rr[:, :1], Rr[:, 1:], br).vertices
    (ref * dv).sum().backward()
    Rg = R.float().to(DEV).requires_grad_(True)
    bg = betas.float().to(DEV).requires_grad_(True)
    out = smpl_hip(global_orient=Rg[:, :1], body_pose=Rg[:, 1:], betas=bg, pose2rot=False)
    assert out.vertices.shape == (B, 6890, 3)
    assert (out.vertices.detach().cpu().double() - ref.detach()).abs().max().item() < 2e-5
    (out.vertices * dv.float().to(DEV)).sum().backward()
    assert relerr(Rg.grad, Rr.grad) < 3e-4
    assert relerr(bg.grad, br.grad) < 3e-4


@pytest.mark.parametrize('return_verts', [False, True])
def test_find_joints_autograd(smpl_hip, smpl_model_np, j_h36m_np, return_verts):
    utils = _mod('utils')
    B = 6
    sm = _mod('smpl_model')
    batch = sm.synthetic_batch(smpl_model_np, j_h36m_np, B, seed=9)
    x6, betas = T(batch['pose6d']), T(batch['betas'])
    gen = torch.Generator().manual_seed(4)
    dj = torch.randn(B, 17, 3, generator=gen, dtype=torch.float64)
    # oracle (fp64)
    xo = x6.double().clone().requires_grad_(True)
    bo = betas.double().clone().requires_grad_(True)
    Jo = T(j_h36m_np).double().clone().requires_grad_(True)
    Ro = oracle.rot6d_to_rotmat(xo.reshape(-1, 6)).view(B, 24, 3, 3)
    jo = oracle.find_joints(oracle.OracleSMPL(smpl_model_np, dtype=torch.float64), bo, Ro[:, :1], Ro[:, 1:], Jo,
                            mask=oracle.find_j_reg_mask(Jo.detach()))
    (jo * dj).sum().backward()
    # HIP path through the reference-shaped API
    xg = x6.to(DEV).requires_grad_(True)
    bg = betas.to(DEV).requires_grad_(True)
    Jg = T(j_h36m_np).to(DEV).requires_grad_(True)
    mask = utils.find_j_reg_mask(Jg.detach())
    assert mask.unique().tolist() == [1.0]
    Rg = utils.rot6d_to_rotmat(xg.reshape(-1, 6)).view(B, 24, 3, 3)
    res = utils.find_joints(smpl_hip, bg, Rg[:, 0:1], Rg[:, 1:], Jg, mask=mask, return_verts=return_verts)
    jg = res[0] if return_verts else res
    assert (jg.detach().cpu().double() - jo.detach()).abs().max().item() < 2e-5
    (jg * dj.float().to(DEV)).sum().backward()
    assert relerr(xg.grad, xo.grad) < 3e-4
    assert relerr(bg.grad, bo.grad) < 3e-4
    assert relerr(Jg.grad, Jo.grad) < 3e-4
    moved = utils.move_pelvis(jg.detach())
    assert moved[:, 0].abs().max().item() == 0.0


def test_discriminator_module(smpl_hip):
    disc = _mod('discriminator')
    torch.manual_seed(0)
    D = disc.Discriminator()
    D.device_model = smpl_hip.device_model
    assert [(k, tuple(v.shape)) for k, v in D.state_dict().items()] == list(oracle.DISC_PARAM_SHAPES)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(7, 24, 6, generator=gen) * 0.6
    xr = x.clone().requires_grad_(True)
    ref = oracle.discriminator_forward({k: v.detach() for k, v in D.state_dict().items()}, xr)
    w = torch.randn(7, 25, 1, generator=gen)
    (ref * w).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    out = D(xg)
    assert out.shape == (7, 25, 1)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=3e-6)
    (out * w.to(DEV)).sum().backward()
    assert relerr(xg.grad, xr.grad) < 3e-4


def test_pose_disc_weight_gradients(smpl_hip):
    eng_mod = _mod('engine')
    g = load_golden('g4_disc.npz')
    sd = oracle.formula_state_dict(oracle.DISC_PARAM_SHAPES, seed=0)
    flat = eng_mod.flatten_state_dict(sd, eng_mod.DISC_KEYS)
    eng = eng_mod.RefineEngine(smpl_hip.device_model, 4, flags=eng_mod.FLAG_POSE_DISC)
    eng.set_pose_disc(flat)
    dP = torch.zeros(eng_mod.DISC_PARAMS, device=DEV)
    sq = eng.pose_disc_backward_params(T(g['x']).to(DEV), 1.0, dP)
    np.testing.assert_allclose(float(sq.sum()) / 100, float(g['loss']), rtol=1e-5)
    grads = eng_mod.unflatten_state_dict(dP.cpu(), sd, eng_mod.DISC_KEYS)
    for i, k in enumerate(eng_mod.DISC_KEYS):       # golden: reference autograd checksums per tensor
        np.testing.assert_allclose(float(grads[k].double().pow(2).sum().sqrt()), float(g[f'gw_l2_{i}']), rtol=5e-4, atol=1e-9)
        np.testing.assert_allclose(float(grads[k].double().sum()), float(g[f'gw_sum_{i}']), rtol=5e-3, atol=2e-6)
    # ragged batch, both update terms, against the oracle's full gradient tensors
    B = 70
    gen = torch.Generator().manual_seed(3)
    xo, xs = torch.randn(B, 24, 6, generator=gen) * 0.6, torch.randn(B, 24, 6, generator=gen) * 0.6
    loss, ref = oracle.discriminator_update_loss_and_grads(sd, xo, xs)
    eng2 = eng_mod.RefineEngine(smpl_hip.device_model, B, flags=eng_mod.FLAG_POSE_DISC)
    eng2.set_pose_disc(flat)
    dP2 = torch.zeros(eng_mod.DISC_PARAMS, device=DEV)
    l0 = eng2.pose_disc_backward_params(xo.to(DEV), 0.0, dP2)
    l1 = eng2.pose_disc_backward_params(xs.to(DEV), 1.0, dP2)
    np.testing.assert_allclose(float((l0 + l1).sum()) / (B * 25), float(loss), rtol=1e-5)
    got = eng_mod.unflatten_state_dict(dP2.cpu(), sd, eng_mod.DISC_KEYS)
    for k in eng_mod.DISC_KEYS:
        assert relerr(got[k], ref[k]) < 1e-3, k


@pytest.mark.parametrize('B', [300, 1030])
def test_pose_disc_weight_gradients_pose_split(smpl_hip, B):
    """the weight-gradient GEMMs split the pose (reduction) dimension into partial slabs from BP = 256 on
    (2 splits at B = 300, 4 at B = 1030) and the conv/head gradients use one slab per wave: same gradients"""
    eng_mod = _mod('engine')
    sd = oracle.formula_state_dict(oracle.DISC_PARAM_SHAPES, seed=0)
    flat = eng_mod.flatten_state_dict(sd, eng_mod.DISC_KEYS)
    gen = torch.Generator().manual_seed(11)
    xo, xs = torch.randn(B, 24, 6, generator=gen) * 0.6, torch.randn(B, 24, 6, generator=gen) * 0.6
    loss, ref = oracle.discriminator_update_loss_and_grads(sd, xo, xs)
    eng = eng_mod.RefineEngine(smpl_hip.device_model, B, flags=eng_mod.FLAG_POSE_DISC)
    eng.set_pose_disc(flat)
    dP = torch.zeros(eng_mod.DISC_PARAMS, device=DEV)
    l0 = eng.pose_disc_backward_params(xo.to(DEV), 0.0, dP)
    l1 = eng.pose_disc_backward_params(xs.to(DEV), 1.0, dP)
    np.testing.assert_allclose(float((l0 + l1).sum()) / (B * 25), float(loss), rtol=1e-5)
    got = eng_mod.unflatten_state_dict(dP.cpu(), sd, eng_mod.DISC_KEYS)
    for k in eng_mod.DISC_KEYS:
        assert relerr(got[k], ref[k]) < 1e-3, k


def test_shape_disc_weight_gradients(smpl_hip):
    eng_mod = _mod('engine')
    sd = oracle.formula_state_dict(oracle.SHAPE_DISC_PARAM_SHAPES, seed=1)
    B = 90
    gen = torch.Generator().manual_seed(5)
    b0, b1 = torch.randn(B, 10, generator=gen), torch.randn(B, 10, generator=gen)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss = (oracle.shape_discriminator_forward(sdr, b0) ** 2).mean() + ((oracle.shape_discriminator_forward(sdr, b1) - 1) ** 2).mean()
    loss.backward()
    eng = eng_mod.RefineEngine(smpl_hip.device_model, B, flags=eng_mod.FLAG_SHAPE_DISC)
    eng.set_shape_disc(eng_mod.flatten_state_dict(sd, eng_mod.SHAPE_DISC_KEYS))
    dP = torch.zeros(eng_mod.SHAPE_DISC_PARAMS, device=DEV)
    l0 = eng.shape_disc_backward_params(b0.to(DEV), 0.0, dP)
    l1 = eng.shape_disc_backward_params(b1.to(DEV), 1.0, dP)
    np.testing.assert_allclose(float((l0 + l1).sum()) / B, float(loss), rtol=1e-5)
    got = eng_mod.unflatten_state_dict(dP.cpu(), sd, eng_mod.SHAPE_DISC_KEYS)
    for k in eng_mod.SHAPE_DISC_KEYS:
        assert relerr(got[k], sdr[k].grad) < 1e-3, k


def test_optimize_pose_refiner_outer_step_matches_oracle(smpl_model_np, j_h36m_np, tmp_path):
    """One outer batch of the driver (3 inner iterations, pose-D + shape-D updates, J step) against
    the oracle's restatement of scripts/optimize.py:220-312 with torch autograd / torch Adam."""
    B, n_inner = 48, 3
    argsmod = _mod('args')
    ckpt = str(tmp_path / 'J.pt')
    argsmod._LazyArgs._ns = argsmod.get_args(['--batch_size', str(B), '--synthetic_batches', '1', '--inner_iters', str(n_inner),
                                               '--shape_disc', '--device', DEV, '--save_j_regressor', ckpt,
                                               '--smpl_dir', '/nonexistent', '--j_regressor_init', '/nonexistent', '--synthetic'])
    opt = _mod('optimize')
    torch.manual_seed(0)
    res = opt.optimize_pose_refiner(log=lambda r: None)
    # ---- oracle replication ----
    sm = _mod('smpl_model')
    disc = _mod('discriminator')
    eng_mod = _mod('engine')
    torch.manual_seed(0)            # utils.set_seed(args.seed=0) precedes the discriminator constructors
    D, SD = disc.Discriminator(), disc.Shape_Discriminator()
    dsd = {k: v.detach().clone() for k, v in D.state_dict().items()}
    ssd = {k: v.detach().clone() for k, v in SD.state_dict().items()}
    full = sm.synthetic_batch(smpl_model_np, j_h36m_np, B, seed=0)
    x6, betas = T(full['pose6d']), T(full['betas'])
    gt_c = oracle.move_pelvis(T(full['gt_j3d']))
    smpl = oracle.OracleSMPL(smpl_model_np)
    J0 = T(j_h36m_np)
    o, p, b, hist = oracle.refine_poses(smpl, J0, x6[:, :1], x6[:, 1:], betas, gt_c, n_inner, disc_sd=dsd, shape_disc_sd=ssd)
    x_opt = torch.cat([o, p], 1)
    # pose-D update: Adam(lr 1e-3) one step
    _, gD = oracle.discriminator_update_loss_and_grads(dsd, x_opt, x6)
    flat = eng_mod.flatten_state_dict(dsd, eng_mod.DISC_KEYS).clone()
    gflat = eng_mod.flatten_state_dict(gD, eng_mod.DISC_KEYS)
    oracle.adam_step(flat, gflat, torch.zeros_like(flat), torch.zeros_like(flat), 1, 1e-3)
    # J step: Adam(lr 1e-2) one step
    _, gJ, _ = oracle.j_regressor_loss_and_grad(smpl, J0, o, p, b, gt_c)
    J1 = J0.clone()
    oracle.adam_step(J1, gJ, torch.zeros_like(J1), torch.zeros_like(J1), 1, 1e-2)
    # ---- compare ----
    np.testing.assert_allclose(res['history'][0]['joint_loss'], hist[-1]['joint_loss'], rtol=2e-3)
    assert (res['disc_flat'].cpu() - flat).abs().max().item() < 2e-5      # Adam step is +-lr for every weight
    assert (res['J_regressor'].cpu() - J1).abs().max().item() < 2e-5
    ck = _mod('checkpoint').load_j_regressor(ckpt)
    assert torch.equal(ck, res['J_regressor'].cpu())
    raw = torch.load(ckpt, weights_only=True)
    assert raw.shape == (17, 6890) and raw.dtype == torch.float32 and raw.stride() == (1, 17)


def _synthetic_2d(joints_m, seed, noise=2.0):
    """gt_j2d in the 224-crop pixel frame: projection of the (noisy) joints through a perturbed camera"""
    gen = torch.Generator().manual_seed(seed)
    B = joints_m.shape[0]
    cam_true = torch.tensor([0.0, 0.0, 2 * 5000 / (224 * 0.9)]).repeat(B, 1) + torch.randn(B, 3, generator=gen) * torch.tensor([0.3, 0.3, 3.0])
    gt2d = oracle.project_joints(joints_m, cam_true) + torch.randn(B, 17, 2, generator=gen) * noise
    return gt2d, cam_true


def test_projection_and_camera_prefit(smpl_hip, smpl_model_np, j_h36m_np):
    """row f1: return_2d_joints core and the 1000-step camera pre-fit (scripts/optimize.py:187-199)"""
    eng_mod = _mod('engine')
    B = 40
    batch = _mod('smpl_model').synthetic_batch(smpl_model_np, j_h36m_np, B, seed=13)
    x6, betas, cam0 = T(batch['pose6d']), T(batch['betas']), T(batch['cam'])
    eng = eng_mod.RefineEngine(smpl_hip.device_model, B)
    eng.set_j_regressor(T(j_h36m_np))
    joints = eng.find_joints_forward(betas.to(DEV), x6d=x6.to(DEV))
    gt2d, _ = _synthetic_2d(joints.cpu(), 1)
    p2d = eng_mod.project_joints(joints, cam0.to(DEV))
    np.testing.assert_allclose(p2d.cpu().numpy(), oracle.project_joints(joints.cpu(), cam0).numpy(), rtol=0, atol=2e-3)
    cam = cam0.clone().to(DEV)
    sq = eng.camera_prefit(x6.to(DEV), betas.to(DEV), gt2d.to(DEV).contiguous(), cam, n_steps=1000, lr=1e-2)
    ref = oracle.camera_prefit(joints.cpu(), gt2d, cam0, 1000, lr=1e-2)
    assert (cam.cpu() - ref).abs().max().item() < 2e-3          # 1000 Adam steps of size <= 1e-2
    e0 = ((gt2d - oracle.project_joints(joints.cpu(), cam0)) ** 2).sum().item()
    assert float(sq.sum()) < 0.2 * e0                            # the fit actually reduced the 2-D error


def test_refine_run_with_2d_term_matches_oracle(smpl_hip, smpl_model_np, j_h36m_np):
    eng_mod = _mod('engine')
    B, n = 24, 5
    batch = _mod('smpl_model').synthetic_batch(smpl_model_np, j_h36m_np, B, seed=17)
    x6, betas, cam0 = T(batch['pose6d']), T(batch['betas']), T(batch['cam'])
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    smpl = oracle.OracleSMPL(smpl_model_np)
    R = oracle.rot6d_to_rotmat(x6.reshape(-1, 6)).view(B, 24, 3, 3)
    j0 = oracle.find_joints(smpl, betas, R[:, :1], R[:, 1:], T(j_h36m_np))
    gt2d, _ = _synthetic_2d(j0, 2)
    o, p, b, hist, c = oracle.refine_poses(smpl, T(j_h36m_np), x6[:, :1], x6[:, 1:], betas, gt_c, n, gt_j2d=gt2d, cam=cam0)
    eng = eng_mod.RefineEngine(smpl_hip.device_model, B)
    eng.set_j_regressor(T(j_h36m_np))
    xd, bd, cd = x6.clone().to(DEV), betas.clone().to(DEV), cam0.clone().to(DEV)
    cm, cv = torch.zeros(B, 3, device=DEV), torch.zeros(B, 3, device=DEV)
    eng.set_reprojection(gt2d.to(DEV).contiguous(), cd, cm, cv)
    m, v = torch.zeros(B, 154, device=DEV), torch.zeros(B, 154, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    eng.refine_run(xd, bd, gt_c.to(DEV).contiguous(), m, v, step, 1e-2, n)
    eng.set_reprojection(None)
    assert (xd.cpu() - torch.cat([o, p], 1)).abs().max().item() < 3e-4
    assert (bd.cpu() - b).abs().max().item() < 3e-4
    assert (cd.cpu() - c).abs().max().item() < 3e-4
    assert (cd.cpu() - cam0).abs().max().item() > 1e-2           # the camera really moved


def _sil_setup(smpl_model_np, j_h36m_np, B, seed):
    from oracle import silhouette_port as sp
    batch = _mod('smpl_model').synthetic_batch(smpl_model_np, j_h36m_np, B, seed=seed)
    x6, betas, cam = T(batch['pose6d']), T(batch['betas']), T(batch['cam'])
    smpl = oracle.OracleSMPL(smpl_model_np)
    R = oracle.rot6d_to_rotmat(x6.reshape(-1, 6)).view(B, 24, 3, 3)
    verts = smpl(R[:, :1], R[:, 1:], betas).vertices
    # target mask: the silhouette of a slightly shifted camera, binarised (stand-in for Mask-RCNN output)
    img = sp.soft_silhouette(verts, smpl_model_np['faces'], cam + torch.tensor([0.15, -0.1, 1.0]))
    mask = (img[:, 0] > 0).float()
    return sp, batch, x6, betas, cam, verts, mask


def test_silhouette_forward_backward(smpl_hip, smpl_model_np, j_h36m_np):
    """row f2: rasteriser + soft silhouette vs the oracle's brute-force restatement of pytorch3d 0.3.0"""
    eng_mod = _mod('engine')
    B = 3
    sp, batch, x6, betas, cam, verts, mask = _sil_setup(smpl_model_np, j_h36m_np, B, 51)
    eng = eng_mod.RefineEngine(smpl_hip.device_model, B, flags=eng_mod.FLAG_SILHOUETTE)
    alpha = eng.silhouette_forward(verts.to(DEV).contiguous(), cam.to(DEV))
    vr, cr = verts.clone().requires_grad_(True), cam.clone().requires_grad_(True)
    ref = sp.soft_silhouette(vr, smpl_model_np['faces'], cr)[:, 0]
    cov_ref, cov = ref.detach() > 0, alpha.cpu() > 0
    assert cov_ref.sum() > 3000 * B
    # coverage may differ only on pixels whose centre sits on an edge to within fp32 rounding
    assert (cov_ref != cov).float().mean().item() < 2e-4
    both = cov_ref & cov
    # alpha = sigmoid(d/1e-4) amplifies fp32 noise in d (d ~ 1e-5): compare where both rasterisers agree
    assert (alpha.cpu()[both] - ref.detach()[both]).abs().mean().item() < 2e-3
    g = (ref.detach() - mask) * 2 / (B * 224 * 224)
    (ref * g).sum().backward()
    dv, dc = eng.silhouette_backward(g.to(DEV).contiguous())
    assert relerr(dc, cr.grad) < 2e-2
    num = (dv.cpu().double() - vr.grad.double()).norm() / vr.grad.double().norm()
    assert num.item() < 2e-2
    # module-level interface (scripts/mesh_renderer.py): (B,4,H,W), alpha in channel 3
    mr = _mod('mesh_renderer')
    out = mr.Mesh_Renderer(224, smpl_hip)({"cam": cam.to(DEV)}, (verts * torch.tensor([-2.0, -2.0, 2.0])).to(DEV))
    assert out.shape == (B, 4, 224, 224) and torch.equal(out[:, 3], alpha)


def test_refine_run_with_silhouette_term(smpl_hip, smpl_model_np, j_h36m_np):
    """BASELINE configs[4]: joint loss + silhouette loss (x100) in the fused loop vs the oracle"""
    eng_mod = _mod('engine')
    B, n = 3, 3
    sp, batch, x6, betas, cam0, verts, mask = _sil_setup(smpl_model_np, j_h36m_np, B, 52)
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    smpl = oracle.OracleSMPL(smpl_model_np)
    o, p, b, hist, c = oracle.refine_poses(smpl, T(j_h36m_np), x6[:, :1], x6[:, 1:], betas, gt_c, n, cam=cam0,
                                           sil_mask=mask[:, None], faces=smpl_model_np['faces'])
    eng = eng_mod.RefineEngine(smpl_hip.device_model, B, flags=eng_mod.FLAG_SILHOUETTE | eng_mod.FLAG_KEEP_VERTS)
    eng.set_j_regressor(T(j_h36m_np))
    xd, bd, cd = x6.clone().to(DEV), betas.clone().to(DEV), cam0.clone().to(DEV)
    cm, cv = torch.zeros(B, 3, device=DEV), torch.zeros(B, 3, device=DEV)
    eng.set_silhouette(mask.to(DEV).contiguous(), cd, cm, cv)
    m, v = torch.zeros(B, 154, device=DEV), torch.zeros(B, 154, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    eng.refine_run(xd, bd, gt_c.to(DEV).contiguous(), m, v, step, 1e-2, n)
    eng.set_silhouette(None)
    # Adam normalises the update: agreement to a fraction of lr unless a near-zero gradient flips sign
    dx = (xd.cpu() - torch.cat([o, p], 1)).abs()
    assert dx.max().item() < 2e-3 and dx.mean().item() < 1e-4
    assert (bd.cpu() - b).abs().max().item() < 2e-3
    assert (cd.cpu() - c).abs().max().item() < 2e-3
    assert (cd.cpu() - cam0).abs().max().item() > 5e-3
    assert 'silhouette_loss' in hist[0]
    # the silhouette adjoint accumulates in fixed point (integer LDS adds): the loop is bitwise reproducible
    outs = []
    for _ in range(2):
        x2, b2, c2 = x6.clone().to(DEV), betas.clone().to(DEV), cam0.clone().to(DEV)
        cm, cv = torch.zeros(B, 3, device=DEV), torch.zeros(B, 3, device=DEV)
        eng.set_silhouette(mask.to(DEV).contiguous(), c2, cm, cv)
        m, v = torch.zeros(B, 154, device=DEV), torch.zeros(B, 154, device=DEV)
        step = torch.zeros(1, dtype=torch.int32, device=DEV)
        eng.refine_run(x2, b2, gt_c.to(DEV).contiguous(), m, v, step, 1e-2, n)
        eng.set_silhouette(None)
        outs.append((x2.cpu(), b2.cpu(), c2.cpu()))
    assert all(torch.equal(a, b_) for a, b_ in zip(outs[0], outs[1]))
    assert torch.equal(outs[0][0], xd.cpu())
