"""Round-2 boundary rows, HIP path vs the CPU oracle (pytest -m gpu):
  * Rodrigues kernel and the pose2rot=True branch of the SMPL operator (smplx default; scripts/smpl.py:61-85)
  * Shape_Discriminator.forward / backward and Discriminator as trainable nn.Modules (scripts/discriminator.py:7-74,
    scripts/optimize.py:276-293) without an SMPL model
  * deferred autograd backward after a second forward on the same engine (the forward-generation counter)
  * adversarial loss values of the inner loop's last iteration (scripts/optimize.py:246-250,323-337)
  * config-1 evaluation report (scripts/test.py:33-138)
"""
import importlib

import numpy as np
import pytest
import torch

import oracle
from conftest import PKG_NAME

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


def test_rodrigues_forward_backward():
    eng = _mod('engine')
    gen = torch.Generator().manual_seed(4)
    aa = torch.randn(500, 3, generator=gen) * 0.8
    aa[:8] = 0.0                                     # theta -> 0: R = I, dR/daa = the generators
    aa[8:16] *= 1e-4
    aa[16:24] *= 1e-7
    aa[24] = torch.tensor([3.1, 0.0, 0.0])           # near pi
    R = eng.rodrigues_forward(aa.to(DEV))
    ref = oracle.rodrigues(aa.double())
    assert (R.cpu().double() - ref).abs().max().item() < 2e-6
    assert torch.equal(R[:8].cpu(), torch.eye(3).expand(8, 3, 3))
    dR = torch.randn(500, 3, 3, generator=gen)
    ar = aa.double().clone().requires_grad_(True)
    (oracle.rodrigues(ar) * dR.double()).sum().backward()
    daa = eng.rodrigues_backward(aa.to(DEV), dR.to(DEV))
    assert torch.isfinite(daa).all()
    # the fp64 oracle differentiates the same formula (autograd); small angles included: the kernel evaluates
    # 1 - cos(theta) as 2 sin^2(theta/2)
    np.testing.assert_allclose(daa.cpu().numpy(), ar.grad.numpy(), rtol=3e-4, atol=2e-5)
    # autograd wrapper
    ag = aa.to(DEV).requires_grad_(True)
    (_mod('smpl').batch_rodrigues(ag) * dR.to(DEV)).sum().backward()
    assert torch.equal(ag.grad, daa)


def test_smpl_operator_pose2rot(smpl_hip, smpl_model_np):
    """smpl(global_orient=(B,3), body_pose=(B,69), betas) with smplx's default pose2rot=True"""
    B = 6
    gen = torch.Generator().manual_seed(12)
    aa = torch.randn(B, 24, 3, generator=gen) * 0.35
    betas = torch.randn(B, 10, generator=gen)
    ag = aa.to(DEV).requires_grad_(True)
    bg = betas.to(DEV).requires_grad_(True)
    out = smpl_hip(global_orient=ag[:, 0], body_pose=ag[:, 1:].reshape(B, 69), betas=bg)
    osm = oracle.OracleSMPL(smpl_model_np, dtype=torch.float64)
    ar = aa.double().clone().requires_grad_(True)
    br = betas.double().clone().requires_grad_(True)
    Rr = oracle.rodrigues(ar.reshape(-1, 3)).view(B, 24, 3, 3)
    ref = osm(Rr[:, :1], Rr[:, 1:], br).vertices
    assert (out.vertices.detach().cpu().double() - ref.detach()).abs().max().item() < 2e-5
    w = torch.randn(B, 6890, 3, generator=gen)
    (out.vertices * w.to(DEV)).sum().backward()
    (ref * w.double()).sum().backward()
    assert relerr(ag.grad, ar.grad) < 3e-4
    assert relerr(bg.grad, br.grad) < 3e-4


def test_two_forwards_before_backward(smpl_hip, smpl_model_np, j_h36m_np):
    """autograd defers backward: find_joints(pred) then find_joints(other pose, other J) on the SAME cached engine,
    then one backward through both -- gradients must equal the oracle's (the wrappers restore the forward state)."""
    utils = _mod('utils')
    B = 9
    sm = _mod('smpl_model')
    b1 = sm.synthetic_batch(smpl_model_np, j_h36m_np, B, seed=41)
    b2 = sm.synthetic_batch(smpl_model_np, j_h36m_np, B, seed=42)
    gen = torch.Generator().manual_seed(2)
    J1 = T(j_h36m_np)
    J2 = (T(j_h36m_np) * (1 + 0.3 * torch.rand(17, 6890, generator=gen)))
    w1, w2 = torch.randn(B, 17, 3, generator=gen), torch.randn(B, 17, 3, generator=gen)

    def run(findj, smpl, rot, dev, dt):
        outs = []
        leaves = []
        for bt, J, w in ((b1, J1, w1), (b2, J2, w2)):
            x = T(bt['pose6d']).to(dev, dt).requires_grad_(True)
            be = T(bt['betas']).to(dev, dt).requires_grad_(True)
            Jl = J.to(dev, dt).clone().requires_grad_(True)
            R = rot(x.reshape(-1, 6)).view(B, 24, 3, 3)
            j = findj(smpl, be, R[:, :1], R[:, 1:], Jl)
            outs.append((j * w.to(dev, dt)).sum())
            leaves.append((x, be, Jl))
        (outs[0] + outs[1]).backward()       # both backward passes run AFTER both forwards
        return leaves

    got = run(utils.find_joints, smpl_hip, utils.rot6d_to_rotmat, DEV, torch.float32)
    ref = run(oracle.find_joints, oracle.OracleSMPL(smpl_model_np, dtype=torch.float64), oracle.rot6d_to_rotmat, 'cpu', torch.float64)
    for (x, be, Jl), (xr, br, Jr) in zip(got, ref):
        assert relerr(x.grad, xr.grad) < 5e-4
        assert relerr(be.grad, br.grad) < 5e-4
        assert relerr(Jl.grad, Jr.grad) < 5e-4


def test_discriminator_modules_train_like_the_reference():
    """D(fake), D(real) -> loss -> backward -> .grad of every parameter and of the input, with NO SMPL model;
    Shape_Discriminator likewise (scripts/optimize.py:276-293 through the nn.Modules)."""
    disc = _mod('discriminator')
    torch.manual_seed(3)
    D, SD = disc.Discriminator().to(DEV), disc.Shape_Discriminator().to(DEV)
    dsd = {k: v.detach().cpu().clone() for k, v in D.state_dict().items()}
    ssd = {k: v.detach().cpu().clone() for k, v in SD.state_dict().items()}
    assert [(k, tuple(v.shape)) for k, v in dsd.items()] == list(oracle.DISC_PARAM_SHAPES)
    B = 33
    gen = torch.Generator().manual_seed(6)
    fake, real = torch.randn(B, 24, 6, generator=gen) * 0.6, torch.randn(B, 24, 6, generator=gen) * 0.6
    fg = fake.to(DEV).requires_grad_(True)
    p_fake, p_real = D(fg), D(real.to(DEV))          # two forwards on the module's engine before one backward
    assert p_fake.shape == (B, 25, 1)
    loss = (p_fake ** 2).mean() + ((p_real - 1) ** 2).mean()
    loss.backward()
    sdr = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
    fr = fake.clone().requires_grad_(True)
    rf, rr = oracle.discriminator_forward(sdr, fr), oracle.discriminator_forward(sdr, real)
    np.testing.assert_allclose(p_fake.detach().cpu().numpy(), rf.detach().numpy(), rtol=0, atol=3e-6)
    lref = (rf ** 2).mean() + ((rr - 1) ** 2).mean()
    lref.backward()
    np.testing.assert_allclose(float(loss), float(lref), rtol=1e-5)
    assert relerr(fg.grad, fr.grad) < 5e-4
    for k, p in D.named_parameters():
        assert p.grad is not None and relerr(p.grad, sdr[k].grad) < 1e-3, k
    # one optimiser step through the module changes the next forward (the upload cache notices the new version)
    torch.optim.Adam(D.parameters(), lr=1e-3).step()
    p2 = D(real.to(DEV))
    with torch.no_grad():
        sd2 = {k: v.detach().cpu() for k, v in D.state_dict().items()}
    np.testing.assert_allclose(p2.detach().cpu().numpy(), oracle.discriminator_forward(sd2, real).numpy(), rtol=0, atol=3e-6)
    assert (p2 - p_real).abs().max().item() > 1e-5
    # shape discriminator
    be = torch.randn(B, 10, generator=gen)
    bg = be.to(DEV).requires_grad_(True)
    s = SD(bg)
    assert s.shape == (B, 1)
    ((s - 1) ** 2).mean().backward()
    ssr = {k: v.clone().requires_grad_(True) for k, v in ssd.items()}
    br = be.clone().requires_grad_(True)
    sr = oracle.shape_discriminator_forward(ssr, br)
    np.testing.assert_allclose(s.detach().cpu().numpy(), sr.detach().numpy(), rtol=0, atol=2e-6)
    ((sr - 1) ** 2).mean().backward()
    assert relerr(bg.grad, br.grad) < 3e-4
    for k, p in SD.named_parameters():
        assert relerr(p.grad, ssr[k].grad) < 1e-3, k


def test_model_less_engine_rejects_smpl_ops():
    eng_mod = _mod('engine')
    eng = eng_mod.RefineEngine(None, 8, flags=eng_mod.FLAG_POSE_DISC, device=DEV)
    with pytest.raises(_mod('_lib').JrrError):
        eng.set_j_regressor(torch.ones(17, 6890))
    with pytest.raises(_mod('_lib').JrrError):
        eng_mod.RefineEngine(None, 8, flags=eng_mod.FLAG_KEEP_VERTS, device=DEV)


def test_refine_aux_losses_match_oracle(smpl_hip, smpl_model_np, j_h36m_np):
    """pose_discriminated_loss / shape_discriminated_loss of the last inner iteration (optimize.py:246-250,327-328)"""
    eng_mod = _mod('engine')
    B, n = 70, 3
    batch = _mod('smpl_model').synthetic_batch(smpl_model_np, j_h36m_np, B, seed=61)
    x6, betas = T(batch['pose6d']), T(batch['betas'])
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    dsd = oracle.formula_state_dict(oracle.DISC_PARAM_SHAPES, seed=0)
    ssd = oracle.formula_state_dict(oracle.SHAPE_DISC_PARAM_SHAPES, seed=1)
    eng = eng_mod.RefineEngine(smpl_hip.device_model, B, flags=eng_mod.FLAG_POSE_DISC | eng_mod.FLAG_SHAPE_DISC)
    eng.set_j_regressor(T(j_h36m_np))
    eng.set_pose_disc(eng_mod.flatten_state_dict(dsd, eng_mod.DISC_KEYS))
    eng.set_shape_disc(eng_mod.flatten_state_dict(ssd, eng_mod.SHAPE_DISC_KEYS))
    xd, bd = x6.clone().to(DEV), betas.clone().to(DEV)
    m, v = torch.zeros(B, 154, device=DEV), torch.zeros(B, 154, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    eng.refine_run(xd, bd, gt_c.to(DEV).contiguous(), m, v, step, 1e-2, n)
    pd, sd = eng.refine_aux_losses(True, True)
    _, _, _, hist = oracle.refine_poses(oracle.OracleSMPL(smpl_model_np), T(j_h36m_np), x6[:, :1], x6[:, 1:], betas, gt_c, n,
                                        disc_sd=dsd, shape_disc_sd=ssd)
    np.testing.assert_allclose(float(pd.sum()) / (B * 25), hist[-1]['pose_discriminated_loss'], rtol=1e-4)
    np.testing.assert_allclose(float(sd.sum()) / B, hist[-1]['shape_discriminated_loss'], rtol=1e-4)


def test_mesh_renderer_takes_pretransformed_vertices(smpl_hip, smpl_model_np, j_h36m_np):
    """Mesh_Renderer.forward(batch, verts) receives the flipped / doubled vertices of render_mesh
    (scripts/optimize.py:80-82), as the reference's module does"""
    eng_mod, mr = _mod('engine'), _mod('mesh_renderer')
    B = 2
    batch = _mod('smpl_model').synthetic_batch(smpl_model_np, j_h36m_np, B, seed=51)
    x6, betas, cam = T(batch['pose6d']).to(DEV), T(batch['betas']).to(DEV), T(batch['cam']).to(DEV)
    eng = eng_mod.RefineEngine(smpl_hip.device_model, B, flags=eng_mod.FLAG_SILHOUETTE | eng_mod.FLAG_KEEP_VERTS)
    eng.set_j_regressor(T(j_h36m_np))
    _, verts = eng.find_joints_forward(betas, x6d=x6, return_verts=True)
    alpha = eng.silhouette_forward(verts, cam)
    renderer = mr.Mesh_Renderer(224, smpl_hip)
    out = renderer({'cam': cam}, verts * torch.tensor([-2.0, -2.0, 2.0], device=DEV))
    assert out.shape == (B, 4, 224, 224) and torch.equal(out[:, 3], alpha) and torch.equal(out[:, 0], torch.ones_like(alpha))
    R = _mod('utils').rot6d_to_rotmat(x6.reshape(-1, 6)).view(B, 24, 3, 3)
    img = mr.render_mesh(smpl_hip, renderer, betas, R[:, :1], R[:, 1:], {'cam': cam})
    # through rotation matrices instead of 6-D input: last-bit vertex differences, amplified by sigmoid(d / 1e-4)
    assert img.shape == (B, 1, 224, 224)
    assert ((img[:, 0] > 0) != (alpha > 0)).float().mean().item() < 1e-4 and (img[:, 0] - alpha).abs().mean().item() < 1e-5


def test_config1_eval_report(smpl_model_np, j_h36m_np, tmp_path, capsys):
    """BASELINE configs[0] / scripts/test.py:33-138: before/after MPJPE & PA-MPJPE of the initial vs a retrained
    regressor, batch of 4 poses, against the oracle's find_joints + evaluate on the same batches."""
    B = 4
    argsmod = _mod('args')
    ckpt = str(tmp_path / 'retrained_J_Regressor.pt')
    gen = torch.Generator().manual_seed(8)
    J_new = T(j_h36m_np) * (1 + 0.2 * torch.rand(17, 6890, generator=gen))       # a "retrained" regressor
    _mod('checkpoint').save_j_regressor(J_new, ckpt)
    argsmod._LazyArgs._ns = argsmod.get_args(['--batch_size', str(B), '--synthetic_batches', '3', '--device', DEV, '--synthetic',
                                               '--eval_j_regressor', ckpt])
    rep = _mod('test').test_pose_refiner_model()
    out = capsys.readouterr().out.split()
    assert out[0] == 'MPJPE' and out[2] == 'PAMPJPE' and out[4] == 'after' and out[5] == 'MPJPE' and out[7] == 'PAMPJPE'
    assert out[1] == f"{rep['mpjpe_before']:.4f}" and out[8] == f"{rep['pampjpe_after']:.4f}"
    sm = _mod('smpl_model')
    smpl = oracle.OracleSMPL(smpl_model_np)
    acc = {k: [] for k in ('mb', 'pb', 'ma', 'pa')}
    for it in range(3):
        full = sm.synthetic_batch(smpl_model_np, j_h36m_np, B, seed=7919 * 1000 + it)
        x6, betas, gt = T(full['pose6d']), T(full['betas']), oracle.move_pelvis(T(full['gt_j3d']))
        R = oracle.rot6d_to_rotmat(x6.reshape(-1, 6)).view(B, 24, 3, 3)
        mask = oracle.find_j_reg_mask(T(j_h36m_np))
        for J, (km, kp) in ((T(j_h36m_np), ('mb', 'pb')), (J_new, ('ma', 'pa'))):
            j = oracle.find_joints(smpl, betas, R[:, :1], R[:, 1:], J, mask=mask)
            m_, p_ = oracle.evaluate(j, gt)
            acc[km].append(float(m_)); acc[kp].append(float(p_))
    np.testing.assert_allclose(rep['mpjpe_before'], np.mean(acc['mb']), rtol=1e-4)
    np.testing.assert_allclose(rep['pampjpe_before'], np.mean(acc['pb']), rtol=1e-4)
    np.testing.assert_allclose(rep['mpjpe_after'], np.mean(acc['ma']), rtol=1e-4)
    np.testing.assert_allclose(rep['pampjpe_after'], np.mean(acc['pa']), rtol=1e-4)
    assert abs(rep['mpjpe_after'] - rep['mpjpe_before']) > 1e-3


def test_driver_on_dataset_tensors(smpl_model_np, j_h36m_np, tmp_path):
    """row f4 wired into the driver: `--data_root` makes optimize_pose_refiner iterate data.data_set (the reference's
    precomputed-tensor layout, scripts/data.py:49-86) -- here on files written from a synthetic batch, with the pose
    stored as AXIS-ANGLE so that the Rodrigues kernel is on the path -- and reproduces the synthetic-batch run."""
    sm, argsmod, opt = _mod('smpl_model'), _mod('args'), _mod('optimize')
    B = 24
    rng = np.random.RandomState(3)
    aa = rng.normal(0, 0.3, size=(B, 24, 3)).astype(np.float32)
    R = oracle.rodrigues(T(aa).reshape(-1, 3)).view(B, 24, 3, 3)
    x6 = R[..., :, :2].reshape(B, 24, 6)
    full = sm.synthetic_batch(smpl_model_np, j_h36m_np, B, seed=0)
    d = tmp_path / 'precomputed_val'
    d.mkdir()
    tensors = {'bboxes': torch.tensor([[100., 200., 700., 800.]]).repeat(B, 1), 'betas': T(full['betas']),
               'estimated_translation': T(full['cam']), 'gt_j2d': torch.rand(B, 17, 2) * 1000, 'gt_j3d': T(full['gt_j3d']),
               'intrinsics': torch.eye(3).repeat(B, 1, 1), 'orient': T(aa[:, 0]), 'pose': T(aa[:, 1:].reshape(B, 69))}
    for k, v in tensors.items():
        torch.save(v, str(d / f'{k}.pt'))
    argsmod._LazyArgs._ns = argsmod.get_args(['--batch_size', str(B), '--inner_iters', '2', '--device', DEV, '--synthetic',
                                               '--data_root', str(tmp_path)])
    res = opt.optimize_pose_refiner(log=lambda r: None)
    assert res['history'][0]['data'] == 'dataset' and len(res['history']) == 1
    # the same batch through the oracle (the loader shuffles: compare as sets of refined poses via a sort key)
    smpl = oracle.OracleSMPL(smpl_model_np)
    dsd = {k: v.detach() for k, v in _seeded_disc_sd().items()}
    gt_c = oracle.move_pelvis(T(full['gt_j3d']))
    o, p, b, hist = oracle.refine_poses(smpl, T(j_h36m_np), x6[:, :1], x6[:, 1:], T(full['betas']), gt_c, 2, disc_sd=dsd)
    ref = torch.cat([o, p], 1)
    got = res['x6d'].cpu()
    # match rows by their (unchanged to 2e-2) first coordinates
    d2 = ((got[:, None, :, :] - ref[None]) ** 2).sum((-1, -2))
    match = d2.argmin(1)
    assert sorted(match.tolist()) == list(range(B))
    assert (got - ref[match]).abs().max().item() < 3e-4
    np.testing.assert_allclose(res['history'][0]['joint_loss'], hist[-1]['joint_loss'], rtol=2e-3)


def _seeded_disc_sd():
    """the driver's discriminator: torch default init right after utils.set_seed(args.seed = 0)"""
    torch.manual_seed(0)
    return _mod('discriminator').Discriminator().state_dict()


def test_silhouette_gradients_where_both_rasterisers_agree(smpl_hip, smpl_model_np, j_h36m_np):
    """row f2 at a ragged batch of 67 poses: coverage agreement >= 99.98 % of the pixels, and -- with the upstream
    gradient restricted to the pixels where both rasterisers see the same face at the same distance -- vertex and camera
    gradients to 2e-3 of the oracle's (pytorch3d 0.3.0 restatement, scripts/mesh_renderer.py:23-79)."""
    from oracle import silhouette_port as sp
    eng_mod = _mod('engine')
    B = 67
    batch = _mod('smpl_model').synthetic_batch(smpl_model_np, j_h36m_np, B, seed=53)
    x6, betas, cam = T(batch['pose6d']), T(batch['betas']), T(batch['cam'])
    R = oracle.rot6d_to_rotmat(x6.reshape(-1, 6)).view(B, 24, 3, 3)
    verts = oracle.OracleSMPL(smpl_model_np)(R[:, :1], R[:, 1:], betas).vertices
    mask = (sp.soft_silhouette(verts, smpl_model_np['faces'], cam + torch.tensor([0.15, -0.1, 1.0]))[:, 0] > 0).float()
    eng = eng_mod.RefineEngine(smpl_hip.device_model, B, flags=eng_mod.FLAG_SILHOUETTE)
    alpha = eng.silhouette_forward(verts.to(DEV).contiguous(), cam.to(DEV).contiguous()).cpu()
    vr, cr = verts.clone().requires_grad_(True), cam.clone().requires_grad_(True)
    ref = sp.soft_silhouette(vr, smpl_model_np['faces'], cr)[:, 0]
    cov_ref, cov = ref.detach() > 0, alpha > 0
    assert cov_ref.sum() > 3000 * B
    assert (cov_ref != cov).float().mean().item() < 2e-4
    same = cov_ref & cov & ((alpha - ref.detach()).abs() < 2e-3)
    assert same.sum().item() > 0.97 * cov_ref.sum().item()
    g = ((ref.detach() - mask) * 2 / (B * 224 * 224)) * same
    (ref * g).sum().backward()
    dv, dc = eng.silhouette_backward(g.to(DEV).contiguous())
    rel_v = ((dv.cpu().double() - vr.grad.double()).norm() / vr.grad.double().norm()).item()
    rel_c = ((dc.cpu().double() - cr.grad.double()).norm() / cr.grad.double().norm()).item()
    assert rel_v < 2e-3, rel_v
    assert rel_c < 2e-3, rel_c


def test_refine_run_all_terms_mid_size(smpl_hip, smpl_model_np, j_h36m_np):
    """joint loss + pose-D + shape-D + 2-D reprojection term together (every term of scripts/optimize.py:252-253 except the
    silhouette) at a ragged batch of 150, 4 iterations, against the oracle -- poses, betas AND the camera."""
    eng_mod = _mod('engine')
    B, n = 150, 4
    batch = _mod('smpl_model').synthetic_batch(smpl_model_np, j_h36m_np, B, seed=81)
    x6, betas, cam0 = T(batch['pose6d']), T(batch['betas']), T(batch['cam'])
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    smpl = oracle.OracleSMPL(smpl_model_np)
    R = oracle.rot6d_to_rotmat(x6.reshape(-1, 6)).view(B, 24, 3, 3)
    j0 = oracle.find_joints(smpl, betas, R[:, :1], R[:, 1:], T(j_h36m_np))
    gen = torch.Generator().manual_seed(5)
    gt_j2d = oracle.project_joints(j0, cam0 + torch.tensor([0.2, -0.1, 2.0]))[..., :2] + torch.randn(B, 17, 2, generator=gen) * 2.0
    dsd = oracle.formula_state_dict(oracle.DISC_PARAM_SHAPES, seed=0)
    ssd = oracle.formula_state_dict(oracle.SHAPE_DISC_PARAM_SHAPES, seed=1)
    o, p, b, hist, c = oracle.refine_poses(smpl, T(j_h36m_np), x6[:, :1], x6[:, 1:], betas, gt_c, n, disc_sd=dsd, shape_disc_sd=ssd,
                                           gt_j2d=gt_j2d, cam=cam0)
    eng = eng_mod.RefineEngine(smpl_hip.device_model, B, flags=eng_mod.FLAG_POSE_DISC | eng_mod.FLAG_SHAPE_DISC)
    eng.set_j_regressor(T(j_h36m_np))
    eng.set_pose_disc(eng_mod.flatten_state_dict(dsd, eng_mod.DISC_KEYS))
    eng.set_shape_disc(eng_mod.flatten_state_dict(ssd, eng_mod.SHAPE_DISC_KEYS))
    xd, bd, cd = x6.clone().to(DEV), betas.clone().to(DEV), cam0.clone().to(DEV)
    cm, cv = torch.zeros(B, 3, device=DEV), torch.zeros(B, 3, device=DEV)
    eng.set_reprojection(gt_j2d.to(DEV).contiguous(), cd, cm, cv)
    m, v = torch.zeros(B, 154, device=DEV), torch.zeros(B, 154, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    sq = torch.zeros(B, device=DEV)
    eng.refine_run(xd, bd, gt_c.to(DEV).contiguous(), m, v, step, 1e-2, n, sqerr=sq)
    eng.set_reprojection(None)
    assert (xd.cpu() - torch.cat([o, p], 1)).abs().max().item() < 6e-4 and (xd.cpu() - torch.cat([o, p], 1)).abs().mean().item() < 5e-6
    assert (bd.cpu() - b).abs().max().item() < 3e-4
    assert (cd.cpu() - c).abs().max().item() < 3e-4
    np.testing.assert_allclose(float(sq.sum()) / (B * 51), hist[-1]['joint_loss'], rtol=2e-3)
    pd, sd = eng.refine_aux_losses(True, True)
    np.testing.assert_allclose(float(pd.sum()) / (B * 25), hist[-1]['pose_discriminated_loss'], rtol=1e-4)
    np.testing.assert_allclose(float(sd.sum()) / B, hist[-1]['shape_discriminated_loss'], rtol=1e-4)
