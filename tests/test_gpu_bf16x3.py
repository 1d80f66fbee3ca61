"""SIDE MODE `FLAG_BLEND_BF16X3` (include/jrr.h) -- the blend-basis adjoint as a split-bf16 product with fp32 accumulation.  It is NOT the
reference's arithmetic and never what bench.py's `value` runs; these are its OWN parity bounds (pytest -m gpu):

  * the pose / shape gradients of `find_joints` against the fp64 oracle's autograd (scripts/utils.py:85-98 backwards), beside the
    exact-fp32 engine on the same inputs: the split product may cost at most 2e-4 of the gradient's scale (the fp32 path's own bound);
  * a 100-iteration Adam trajectory (scripts/optimize.py:220-265, joint loss) against the exact-fp32 engine and against the oracle:
    the regressed 3-D joints of the refined poses within `north_star`'s 1e-4 m;
  * the flag is refused without a body model and leaves the support-tile iterations untouched (bit-identical to an engine without it).
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


@pytest.fixture(scope='module')
def eng_mod():
    return importlib.import_module(PKG_NAME + '.engine')


@pytest.fixture(scope='module')
def dmodel(eng_mod, smpl_model_np):
    return eng_mod.DeviceModel(smpl_model_np, DEV)


def _batch(smpl_model_np, j_h36m_np, B, seed):
    sm = importlib.import_module(PKG_NAME + '.smpl_model')
    return sm.synthetic_batch(smpl_model_np, j_h36m_np, B, seed=seed)


@pytest.mark.parametrize('B', [37, 256])
def test_bf16x3_gradients_vs_oracle_and_fp32(eng_mod, dmodel, smpl_model_np, j_h36m_np, B):
    batch = _batch(smpl_model_np, j_h36m_np, B, seed=12)
    x6d, betas = T(batch['pose6d']).double(), T(batch['betas']).double()
    dj = torch.randn(B, 17, 3, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    J = T(j_h36m_np).double()
    smpl = oracle.OracleSMPL(smpl_model_np, dtype=torch.float64)
    b_r, leaf = betas.clone().requires_grad_(True), x6d.clone().requires_grad_(True)
    R = oracle.rot6d_to_rotmat(leaf.reshape(-1, 6)).view(B, 24, 3, 3)
    joints = oracle.find_joints(smpl, b_r, R[:, :1], R[:, 1:], J, mask=oracle.find_j_reg_mask(J))
    (joints * dj).sum().backward()
    got = {}
    for name, extra in (('f32', 0), ('bf16x3', eng_mod.FLAG_BLEND_BF16X3)):
        eng = eng_mod.RefineEngine(dmodel, B, flags=eng_mod.FLAG_KEEP_VERTS | extra)
        eng.set_j_regressor(T(j_h36m_np))
        xd, bd = x6d.float().contiguous().to(DEV), betas.float().to(DEV)
        eng.find_joints_forward(bd, x6d=xd)
        dx, db, _ = eng.find_joints_backward(bd, dj.float().to(DEV), x6d=xd)
        got[name] = (dx.cpu().double(), db.cpu().double())

    def relerr(a, b):
        return ((a - b).abs().max() / b.abs().max()).item()
    for name in got:
        assert relerr(got[name][0], leaf.grad) < 2e-4, name
        assert relerr(got[name][1], b_r.grad) < 2e-4, name
    # the split product against the exact one: what the side mode itself costs (three bf16 products drop ~ 2^-16 of a term)
    assert relerr(got['bf16x3'][0], got['f32'][0]) < 5e-5
    assert relerr(got['bf16x3'][1], got['f32'][1]) < 5e-5
    assert not torch.equal(got['bf16x3'][0], got['f32'][0])          # (the mode was actually taken)


def test_bf16x3_hundred_iteration_trajectory(eng_mod, dmodel, smpl_model_np, j_h36m_np):
    """100 fused iterations (Adam, lr 1e-2 as scripts/optimize.py:201-202): the regressed joints of the refined poses within 1e-4 m of the
    exact-fp32 engine's and of the oracle's; the parameters themselves within the Adam-amplification bound of the other trajectory tests"""
    B, n = 64, 100
    batch = _batch(smpl_model_np, j_h36m_np, B, 31)
    x6d, betas = T(batch['pose6d']), T(batch['betas'])
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    res = {}
    for name, extra in (('f32', 0), ('bf16x3', eng_mod.FLAG_BLEND_BF16X3)):
        eng = eng_mod.RefineEngine(dmodel, B, flags=eng_mod.FLAG_KEEP_VERTS | extra)
        eng.set_j_regressor(T(j_h36m_np))
        xd, bd = x6d.clone().to(DEV), betas.clone().to(DEV)
        m, v = torch.zeros(B, 154, device=DEV), torch.zeros(B, 154, device=DEV)
        step = torch.zeros(1, dtype=torch.int32, device=DEV)
        sq = torch.zeros(B, device=DEV)
        eng.refine_run(xd, bd, gt_c.to(DEV).contiguous(), m, v, step, 1e-2, n, sqerr=sq)
        joints = eng.find_joints_forward(bd, x6d=xd)
        res[name] = (xd.cpu(), bd.cpu(), joints.cpu(), sq.cpu())
    smpl = oracle.OracleSMPL(smpl_model_np)
    o, p, b, hist = oracle.refine_poses(smpl, T(j_h36m_np), x6d[:, :1], x6d[:, 1:], betas, gt_c, n)
    Ro = oracle.rot6d_to_rotmat(torch.cat([o, p], 1).reshape(-1, 6)).view(B, 24, 3, 3)
    j_or = oracle.find_joints(smpl, b, Ro[:, :1], Ro[:, 1:], T(j_h36m_np), mask=oracle.find_j_reg_mask(T(j_h36m_np)))
    # (the exact-fp32 engine's own 100-iteration oracle tests live in tests/test_gpu_trajectory.py; here it is the yardstick)
    dj = (res['bf16x3'][2] - j_or).abs().max().item()
    assert dj < 1e-4, dj                                              # north_star: regressed 3-D joints within 1e-4 m
    np.testing.assert_allclose(float(res['bf16x3'][3].sum()) / (B * 51), hist[-1]['joint_loss'], rtol=5e-3)
    dj = (res['bf16x3'][2] - res['f32'][2]).abs().max().item()
    assert dj < 1e-4, dj
    d = (res['bf16x3'][0] - res['f32'][0]).abs()
    assert d.mean().item() < 2e-5, d.mean().item()                    # (max: Adam-amplified near-zero gradient entries; the mean pins the trajectory)


def test_bf16x3_flag_scope(eng_mod, dmodel, smpl_model_np, j_h36m_np):
    with pytest.raises(Exception):
        eng_mod.RefineEngine(None, 64, flags=eng_mod.FLAG_POSE_DISC | eng_mod.FLAG_NO_MODEL | eng_mod.FLAG_BLEND_BF16X3)
    # the support-tile iterations run the exact kernels whatever the flag says
    B, n = 128, 3
    batch = _batch(smpl_model_np, j_h36m_np, B, 7)
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    out = []
    for extra in (0, eng_mod.FLAG_BLEND_BF16X3):
        eng = eng_mod.RefineEngine(dmodel, B, flags=eng_mod.FLAG_KEEP_VERTS | eng_mod.FLAG_SUPPORT_TILES | extra)
        eng.set_j_regressor(T(j_h36m_np))
        eng.j_support_info()
        if not eng.support_tiles()[0]:
            pytest.skip('support tiles not engaged on this model / variant')
        xd, bd = T(batch['pose6d']).clone().to(DEV), T(batch['betas']).clone().to(DEV)
        m, v = torch.zeros(B, 154, device=DEV), torch.zeros(B, 154, device=DEV)
        step = torch.zeros(1, dtype=torch.int32, device=DEV)
        eng.refine_run(xd, bd, gt_c.to(DEV).contiguous(), m, v, step, 1e-2, n)
        out.append(xd.cpu())
    assert torch.equal(out[0], out[1])
