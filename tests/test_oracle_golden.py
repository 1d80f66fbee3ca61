"""The oracle (oracle/reference_port.py) against the golden vectors captured from the reference's
own importable modules (tests/golden/make_golden.py).  CPU only."""
import importlib

import numpy as np
import torch

import oracle
from conftest import load_golden, PKG_NAME

T = torch.from_numpy


class Stub:
    def __init__(self, v):
        self.v = v

    def __call__(self, **kw):
        return oracle.SMPLOutput(self.v)


def test_g1_rot6d():
    g = load_golden('g1_rot6d.npz')
    R = oracle.rot6d_to_rotmat(T(g['x']))
    assert torch.equal(R, T(g['R']))          # same torch ops, bit-exact
    assert not torch.isnan(R).any()           # degenerate rows give zero columns, not NaN


def test_g2_find_joints_and_grads():
    g = load_golden('g2_find_joints.npz')
    verts = T(g['verts']).float().requires_grad_(True)
    gt = T(g['gt'])
    tri = load_golden('j_regressor_triplets.npz')
    sm = importlib.import_module(PKG_NAME + '.smpl_model')
    Js = {'ck': T(sm.j_regressor_from_triplets(tri['rows'], tri['cols'], tri['vals'])),
          'syn': T(sm.synthetic_h36m_regressor(None, seed=7))}
    for tag, J0 in Js.items():
        J = J0.clone().requires_grad_(True)
        mask = oracle.find_j_reg_mask(J.detach())
        assert mask.unique().tolist() == g[f'{tag}_mask_unique'].tolist() == [1.0]
        joints = oracle.find_joints(Stub(verts), None, None, None, J, mask=mask)
        np.testing.assert_allclose(joints.detach().numpy(), g[f'{tag}_joints'], rtol=0, atol=1e-6)
        loss = ((oracle.move_pelvis(joints) - gt) ** 2).mean()
        np.testing.assert_allclose(float(loss), float(g[f'{tag}_loss']), rtol=1e-6)
        gJ, gV = torch.autograd.grad(loss, [J, verts])
        r, c = torch.nonzero(gJ, as_tuple=True)
        assert r.tolist() == g[f'{tag}_gJ_rows'].tolist() and c.tolist() == g[f'{tag}_gJ_cols'].tolist()
        np.testing.assert_allclose(gJ[r, c].numpy(), g[f'{tag}_gJ_vals'], rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(gV[:, ::689, :].numpy(), g[f'{tag}_gV_sample'], rtol=1e-5, atol=1e-10)
        np.testing.assert_allclose(float(gV.pow(2).sum().sqrt()), float(g[f'{tag}_gV_l2']), rtol=1e-5)


def test_g3_pelvis_and_loss():
    g = load_golden('g3_pelvis_loss.npz')
    assert torch.equal(oracle.move_pelvis(T(g['j'])), T(g['moved']))
    gt_c = oracle.move_pelvis(T(g['gt_mm']))
    assert torch.equal(gt_c, T(g['gt_moved']))
    d = oracle.move_pelvis(T(g['j'])) - gt_c / 1000
    np.testing.assert_allclose(float((d ** 2).sum() / d.numel() * oracle.W_JOINT), float(g['joint_loss_w']), rtol=1e-6)


def test_g4_discriminators():
    g = load_golden('g4_disc.npz')
    sd = oracle.formula_state_dict(oracle.DISC_PARAM_SHAPES, seed=0)
    sd = {k: v.requires_grad_(True) for k, v in sd.items()}
    x = T(g['x']).requires_grad_(True)
    out = oracle.discriminator_forward(sd, x)
    np.testing.assert_allclose(out.detach().numpy(), g['out'], rtol=0, atol=2e-6)
    loss = ((out - 1) ** 2).mean()
    np.testing.assert_allclose(float(loss), float(g['loss']), rtol=1e-6)
    grads = torch.autograd.grad(loss, [x] + list(sd.values()))
    np.testing.assert_allclose(grads[0].numpy(), g['gx'], rtol=1e-4, atol=1e-9)
    for i, gw in enumerate(grads[1:]):
        np.testing.assert_allclose(float(gw.pow(2).sum().sqrt()), float(g[f'gw_l2_{i}']), rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(float(gw.sum()), float(g[f'gw_sum_{i}']), rtol=1e-3, atol=1e-6)
    ssd = oracle.formula_state_dict(oracle.SHAPE_DISC_PARAM_SHAPES, seed=1)
    b = T(g['betas']).requires_grad_(True)
    so = oracle.shape_discriminator_forward(ssd, b)
    np.testing.assert_allclose(so.detach().numpy(), g['sout'], rtol=0, atol=1e-6)
    (gb,) = torch.autograd.grad(((so - 1) ** 2).mean(), [b])
    np.testing.assert_allclose(gb.numpy(), g['gbetas'], rtol=1e-4, atol=1e-9)


def test_g5_adam():
    g = load_golden('g5_adam.npz')
    p = T(g['traj'][0]).clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for s in range(3):
        oracle.adam_step(p, T(g['grads'][s]), m, v, s + 1, lr=1e-2)
        np.testing.assert_allclose(p.numpy(), g['traj'][s + 1], rtol=0, atol=2e-7)


def test_g6_evaluate():
    g = load_golden('g6_evaluate.npz')
    mpjpe, pa = oracle.evaluate(T(g['pred']), T(g['target_mm']))
    np.testing.assert_allclose(mpjpe, float(g['mpjpe']), rtol=1e-5)
    np.testing.assert_allclose(pa, float(g['pampjpe']), rtol=1e-4)
    s1 = oracle.batch_compute_similarity_transform_torch(T(g['pred']), T(g['target_mm']) / 1000)
    np.testing.assert_allclose(s1.numpy(), g['s1hat'], rtol=0, atol=1e-5)


def test_g7_inner_loop(smpl_model_np, j_h36m_np):
    g = load_golden('g7_inner_loop.npz')
    sm = importlib.import_module(PKG_NAME + '.smpl_model')
    batch = sm.synthetic_batch(smpl_model_np, j_h36m_np, 4, seed=3)
    smpl = oracle.OracleSMPL(smpl_model_np)
    pose6 = T(batch['pose6d'])
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    dsd = oracle.formula_state_dict(oracle.DISC_PARAM_SHAPES, seed=0)
    ssd = oracle.formula_state_dict(oracle.SHAPE_DISC_PARAM_SHAPES, seed=1)
    rec = {}
    o, p, b, hist = oracle.refine_poses(smpl, T(j_h36m_np), pose6[:, :1], pose6[:, 1:], T(batch['betas']), gt_c, 10,
                                        disc_sd=dsd, shape_disc_sd=ssd, record=lambda it, d: rec.setdefault(it, d))
    np.testing.assert_allclose(rec[0]['joints'].numpy(), g['joints0'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(rec[0]['g_pose'].numpy(), g['g_pose0'], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(rec[0]['g_orient'].numpy(), g['g_orient0'], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(rec[0]['g_betas'].numpy(), g['g_betas0'], rtol=2e-4, atol=1e-6)
    h = np.array([[x['loss'], x['joint_loss'], x['pose_discriminated_loss'], x['shape_discriminated_loss']] for x in hist])
    np.testing.assert_allclose(h, g['hist'], rtol=2e-4)
    np.testing.assert_allclose(p.numpy(), g['pose'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(o.numpy(), g['orient'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(b.numpy(), g['betas'], rtol=0, atol=2e-5)


def test_g8_jstep_support(j_h36m_np):
    g8 = load_golden('g8_jstep.npz')
    g2 = load_golden('g2_find_joints.npz')
    verts, gt = T(g2['verts']).float(), T(g2['gt'])
    J0 = T(j_h36m_np)
    J = J0.clone()
    m, v = torch.zeros_like(J), torch.zeros_like(J)
    for s in range(3):
        Jr = J.clone().requires_grad_(True)
        joints = oracle.find_joints(Stub(verts), None, None, None, Jr, mask=oracle.find_j_reg_mask(Jr.detach()))
        loss = ((oracle.move_pelvis(joints) - gt) ** 2).mean()
        (gJ,) = torch.autograd.grad(loss, [Jr])
        oracle.adam_step(J, gJ, m, v, s + 1, lr=1e-2)
    r, c = torch.nonzero(J != J0, as_tuple=True)
    assert r.tolist() == g8['changed_rows'].tolist() and c.tolist() == g8['changed_cols'].tolist()
    assert len(r) == int(g8['n_positive']) == 62
    assert bool(g8['zeros_stay_zero']) and bool((J[J0 == 0] == 0).all())
    assert bool(g8['negatives_unchanged']) and bool((J[J0 < 0] == J0[J0 < 0]).all())
    np.testing.assert_allclose(J[r, c].numpy(), g8['new_vals'], rtol=0, atol=1e-6)
