"""Known-answer tests for the parity-unpinned SMPL LBS restatement. CPU only."""
import numpy as np
import torch

import oracle

T = torch.from_numpy


def _model64(m):
    return {k: (T(np.asarray(v)).double() if k != 'parents' else T(np.asarray(v)).long()) for k, v in m.items()
            if k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights', 'parents')}


def _rand_rot(n, seed, scale=0.4):
    g = torch.Generator().manual_seed(seed)
    return oracle.rodrigues(torch.randn(n, 3, generator=g, dtype=torch.float64) * scale)


def test_k1_identity(smpl_model_np):
    m = _model64(smpl_model_np)
    R = torch.eye(3, dtype=torch.float64).expand(2, 24, 3, 3)
    verts, joints = oracle.smpl_lbs(m, R, torch.zeros(2, 10, dtype=torch.float64))
    assert torch.allclose(verts, m['v_template'].expand(2, -1, -1), atol=1e-12)
    assert torch.allclose(joints, (m['J_regressor'] @ m['v_template']).expand(2, -1, -1), atol=1e-12)


def test_k2_global_rotation_is_rigid_about_root(smpl_model_np):
    m = _model64(smpl_model_np)
    R = torch.eye(3, dtype=torch.float64).repeat(3, 24, 1, 1)
    R0 = _rand_rot(3, 5, 1.0)
    R[:, 0] = R0
    verts, _ = oracle.smpl_lbs(m, R, torch.zeros(3, 10, dtype=torch.float64))
    j0 = (m['J_regressor'] @ m['v_template'])[0]
    expect = torch.einsum('brc,vc->bvr', R0, m['v_template'] - j0) + j0
    assert torch.allclose(verts, expect, atol=1e-10)


def test_k3_shape_only_is_linear(smpl_model_np):
    m = _model64(smpl_model_np)
    g = torch.Generator().manual_seed(2)
    b = torch.randn(2, 10, generator=g, dtype=torch.float64)
    R = torch.eye(3, dtype=torch.float64).expand(2, 24, 3, 3)
    verts, joints, aux = oracle.smpl_lbs(m, R, b, return_all=True)
    expect = m['v_template'] + torch.einsum('bl,vkl->bvk', b, m['shapedirs'])
    assert torch.allclose(verts, expect, atol=1e-10)
    assert torch.allclose(aux['J'], torch.einsum('jv,bvk->bjk', m['J_regressor'], expect), atol=1e-10)


def test_k4_child_rotation_leaves_unweighted_vertices(smpl_model_np):
    m = _model64(smpl_model_np)
    m = dict(m)
    m['posedirs'] = torch.zeros_like(m['posedirs'])      # isolate the skinning term
    R = torch.eye(3, dtype=torch.float64).repeat(1, 24, 1, 1)
    R[:, 18] = _rand_rot(1, 9, 1.0)                       # left elbow: subtree {18,20,22}
    verts, _ = oracle.smpl_lbs(m, R, torch.zeros(1, 10, dtype=torch.float64))
    sub = m['lbs_weights'][:, [18, 20, 22]].sum(1)
    fixed = sub == 0
    assert fixed.sum() > 1000 and (~fixed).sum() > 10
    assert torch.allclose(verts[0, fixed], m['v_template'][fixed], atol=1e-12)
    assert (verts[0, ~fixed] - m['v_template'][~fixed]).abs().max() > 1e-3


def test_k5_translation_equivariance(smpl_model_np):
    m = _model64(smpl_model_np)
    assert torch.allclose(m['lbs_weights'].sum(1), torch.ones(6890, dtype=torch.float64), atol=1e-6)
    assert torch.allclose(m['J_regressor'].sum(1), torch.ones(24, dtype=torch.float64), atol=1e-6)
    R = _rand_rot(24, 11).view(1, 24, 3, 3)
    b = torch.zeros(1, 10, dtype=torch.float64)
    v1, _ = oracle.smpl_lbs(m, R, b)
    m2 = dict(m)
    t = torch.tensor([0.3, -0.2, 0.5], dtype=torch.float64)
    m2['v_template'] = m['v_template'] + t
    v2, _ = oracle.smpl_lbs(m2, R, b)
    # rows of W and J_regressor sum to 1 (to float32 rounding of the stored model)
    assert torch.allclose(v2, v1 + t, atol=1e-6)


def test_k6_gradcheck_fp64(smpl_model_np, j_h36m_np):
    m = _model64(smpl_model_np)
    Jn = oracle.normalize_j_regressor(T(j_h36m_np).double())
    g = torch.Generator().manual_seed(4)
    x6 = torch.randn(1, 24, 6, generator=g, dtype=torch.float64)
    b = torch.randn(1, 10, generator=g, dtype=torch.float64) * 0.5
    gt = torch.randn(1, 17, 3, generator=g, dtype=torch.float64) * 0.2

    def f(x6, b):
        R = oracle.rot6d_to_rotmat(x6.reshape(-1, 6)).view(1, 24, 3, 3)
        verts, _ = oracle.smpl_lbs(m, R, b)
        return ((oracle.move_pelvis(Jn @ verts) - gt) ** 2).mean()

    x6r, br = x6.clone().requires_grad_(True), b.clone().requires_grad_(True)
    gx, gb = torch.autograd.grad(f(x6r, br), [x6r, br])
    eps = 1e-6
    for idx in [(0, 0, 0), (0, 3, 4), (0, 18, 1), (0, 23, 5)]:
        d = torch.zeros_like(x6)
        d[idx] = eps
        fd = (f(x6 + d, b) - f(x6 - d, b)) / (2 * eps)
        assert abs(float(fd) - float(gx[idx])) < 1e-6 * max(1.0, abs(float(fd)))
    for l in (0, 7):
        d = torch.zeros_like(b)
        d[0, l] = eps
        fd = (f(x6, b + d) - f(x6, b - d)) / (2 * eps)
        assert abs(float(fd) - float(gb[0, l])) < 1e-6 * max(1.0, abs(float(fd)))


def test_k8_two_independent_restatements_agree(smpl_model_np):
    """The torch restatement (homogeneous 4x4 chain, smplx's op order) against the float64 numpy LBS of the data generator
    (smpl_model._lbs_np: rotation / translation parts kept apart, einsum contractions) -- written separately from the same published
    algorithm.  Agreement pins neither to smplx, but a slip in one of the two would show here; also on the
    capsule body (another skeleton-relative vertex layout, random file order)."""
    import importlib
    from conftest import PKG_NAME
    sm = importlib.import_module(PKG_NAME + '.smpl_model')
    for body in (smpl_model_np, sm.synthetic_smpl(1234, kind='capsules')):
        m = _model64(body)
        B = 5
        R = _rand_rot(B * 24, seed=21, scale=0.6).view(B, 24, 3, 3)
        betas = torch.randn(B, 10, generator=torch.Generator().manual_seed(22), dtype=torch.float64)
        verts, _ = oracle.smpl_lbs(m, R, betas)
        ref = sm._lbs_np(body, R.numpy(), betas.numpy())
        assert np.abs(verts.numpy() - ref).max() < 1e-9


def test_k9_joint_loss_reads_the_vertices_through_the_regressor_support_only(smpl_model_np, j_h36m_np):
    """What JRR_FLAG_SUPPORT_TILES rests on, stated on the REFERENCE algorithm (scripts/utils.py:85-103, scripts/optimize.py:228-239):
    pred_joints = normalise(relu(J * mask)) @ vertices, so (a) the joints -- hence every loss term of configs 2-4 and every gradient --
    do not change by one bit when the vertices OUTSIDE the regressor's positive columns are replaced by anything finite, and (b) the
    gradient of the joint loss w.r.t. the vertices is exactly zero there.  62 positive entries on 58 vertices for the shipped
    checkpoint's structure: the other 6832 vertices of a joint-loss iteration are computed for nothing."""
    B = 3
    smpl = oracle.OracleSMPL(smpl_model_np)
    J = T(j_h36m_np)
    support = (J > 0).any(0)
    assert int(support.sum()) == 58 and int((J > 0).sum()) == 62
    g = torch.Generator().manual_seed(11)
    x6 = torch.randn(B, 24, 6, generator=g)
    betas = torch.randn(B, 10, generator=g)
    R = oracle.rot6d_to_rotmat(x6.reshape(-1, 6)).view(B, 24, 3, 3)
    gt = torch.randn(B, 17, 3, generator=g)

    class Perturbed:
        def __call__(self, **kw):
            out = smpl(**kw)
            v = out.vertices.clone()
            v[:, ~support] = torch.randn(B, int((~support).sum()), 3, generator=g) * 100.0      # anything finite
            return oracle.SMPLOutput(vertices=v, joints=out.joints)
    mask = oracle.find_j_reg_mask(J)
    ja = oracle.find_joints(smpl, betas, R[:, :1], R[:, 1:], J, mask)
    jb = oracle.find_joints(Perturbed(), betas, R[:, :1], R[:, 1:], J, mask)
    assert torch.equal(ja, jb)
    # (b) d(joint loss) / d(vertices) is exactly zero outside the support
    _, verts = oracle.find_joints(smpl, betas, R[:, :1], R[:, 1:], J, mask, return_verts=True)
    v = verts.detach().clone().requires_grad_(True)
    joints = torch.matmul(oracle.normalize_j_regressor(J, mask)[None].expand(B, -1, -1), v)
    loss = ((oracle.move_pelvis(joints) - gt) ** 2).mean()
    loss.backward()
    assert (v.grad[:, ~support] == 0).all() and (v.grad[:, support] != 0).any()
