"""Round-4 parity rows, HIP path (through the C ABI) vs the CPU oracle (pytest -m gpu):
  * EVERY skinning / backward kernel family the library can select -- the 8-slot default, the 12-slot variant
    (JRR_SKIN_JOINTS=12), the dense kernels (JRR_DENSE_SKINNING=1), the role backward kernel (JRR_BWD16=0), the internally
    re-ordered body (JRR_VERTEX_ORDER=sorted: one tile stays wide), a capsule body in a random file order (the library's
    kinematic-chain order leaves no tile above 8 joints) and the PER-TILE classes (the benchmarked body with ONE 13-joint tile) -- each against the ORACLE,
    not against each other: find_joints forward + backward incl. dJ (B = 37, 130), the J step's gradient, a 3-iteration
    refinement with the pose discriminator (B = 200) and the benchmarked batch of 4096 on a strided subset.
    Reference: /root/reference/scripts/utils.py:85-103, scripts/optimize.py:220-265,300-312.
  * forward reuse after a J step at batches whose padded size is an odd multiple of 128 (B = 300, 600: the last 128 pose
    columns of the re-regression slab), against the sequence that repeats the forward
  * the J step over the regressor's support with the all-reduce payload restricted to it (jrr_j_regressor_grad_support /
    jrr_j_step_apply_support): bit-identical to the dense pair
  * a model FILE in the licensed distribution's format (chumpy objects, sparse J_regressor) -> SMPL(model_dir) -> kernels
"""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

import oracle
from conftest import PKG_NAME, write_chumpy_style_pickle

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = 'cuda:0'
KNOBS = ('JRR_DENSE_SKINNING', 'JRR_SKIN_JOINTS', 'JRR_VERTEX_ORDER', 'JRR_BWD16')


def _mod(name):
    return importlib.import_module(f'{PKG_NAME}.{name}')


def _fresh_state(B):
    return (torch.zeros(B, 154, device=DEV), torch.zeros(B, 154, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV))


# variant id -> (environment of jrr_model_create, body, expected joint slots, wide tiles expected?)
VARIANTS = {
    'default': ({}, 'surface', 8, False),
    'skin12': ({'JRR_SKIN_JOINTS': '12'}, 'surface', 12, False),
    'dense': ({'JRR_DENSE_SKINNING': '1'}, 'surface', 0, False),
    'role_bwd': ({'JRR_BWD16': '0'}, 'surface', 8, False),
    'role_bwd12': ({'JRR_BWD16': '0', 'JRR_SKIN_JOINTS': '12'}, 'surface', 12, False),
    'sorted': ({'JRR_VERTEX_ORDER': 'sorted'}, 'surface', 8, True),            # the surface body in the library's chain order: 1 tile of 9 joints
    'capsules': ({}, 'capsules', 8, False),                                    # random file order -> kinematic-chain order: no tile above 8
    'capsules_role_bwd': ({'JRR_BWD16': '0'}, 'capsules', 8, False),           # ... with the role backward kernel
    'wide13': ({}, 'wide13', 8, True),                                         # ONE 13-joint tile in the file order
    'wide13_role_bwd': ({'JRR_BWD16': '0'}, 'wide13', 8, True),                # a wide tile + role kernel: its dense form
    'wide13_skin12': ({'JRR_SKIN_JOINTS': '12'}, 'wide13', 12, True),          # ... under the 12-slot kernels: second pass over slots 12..
    'hinted': ({}, 'hinted', 8, False),                                        # the regressor's support stored FIRST (jrr_model_create_hinted), packed into tiles of <= 8 joints
    'hinted_capsules': ({}, 'hinted_capsules', 8, True),                       # ... on the capsule body (100 support vertices; one tile behind them ends up with 9 joints)
}


@pytest.fixture(scope='module', params=list(VARIANTS))
def variant(request, smpl_model_np, j_h36m_np):
    if any(k in os.environ for k in KNOBS):
        pytest.skip('the suite itself runs under a forced skinning variant')
    env, body, slots, wide = VARIANTS[request.param]
    sm, eng_mod = _mod('smpl_model'), _mod('engine')
    if body in ('capsules', 'hinted_capsules'):
        model = sm.synthetic_smpl(1234, kind='capsules')
    elif body == 'wide13':
        model = sm.with_wide_tile(smpl_model_np, 100, 13)
    else:
        model = smpl_model_np
    # a regressor with the shipped checkpoint's structure on THIS body's vertices (the capsule body has its own vertex order)
    J = j_h36m_np if 'capsules' not in body else sm.synthetic_h36m_regressor(model, seed=7, support=8)
    hint = np.nonzero((J > 0).any(0))[0] if body.startswith('hinted') else None
    os.environ.update(env)          # the knobs are read by jrr_model_create
    try:
        dm = eng_mod.DeviceModel(model, DEV, hint_vertices=hint)
    finally:
        for k in env:
            os.environ.pop(k, None)
    assert dm.info['joint_slots'] == slots, dm.info
    assert (dm.info['wide_tiles'] > 0) == wide, dm.info
    if body == 'capsules':
        assert dm.info['internal_vertex_order'] and dm.info['most_joints_per_tile'] <= 8
    if body == 'wide13':
        assert dm.info['most_joints_per_tile'] == 13 and dm.info['wide_tiles'] == 1 and not dm.info['internal_vertex_order']
    if hint is not None:
        assert dm.info['hinted_vertices_stored_first'] == len(hint) and dm.info['internal_vertex_order'], dm.info
    else:
        assert dm.info['hinted_vertices_stored_first'] == 0
    return dict(name=request.param, model=model, dm=dm, J=np.ascontiguousarray(J), eng_mod=eng_mod, sm=sm)


def _relerr(a, b):
    return ((a.cpu().double() - b.double()).abs().max() / b.double().abs().max()).item()


@pytest.mark.parametrize('B', [37, 130])
def test_find_joints_forward_backward_vs_oracle(variant, B):
    v = variant
    batch = v['sm'].synthetic_batch(v['model'], v['J'], B, seed=40 + B)
    x6d, betas = T(batch['pose6d']).double(), T(batch['betas']).double()
    dj = torch.randn(B, 17, 3, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    J = T(v['J']).double().requires_grad_(True)
    smpl = oracle.OracleSMPL(v['model'], dtype=torch.float64)
    xr, br = x6d.clone().requires_grad_(True), betas.clone().requires_grad_(True)
    R = oracle.rot6d_to_rotmat(xr.reshape(-1, 6)).view(B, 24, 3, 3)
    ref_j, ref_v = oracle.find_joints(smpl, br, R[:, :1], R[:, 1:], J, mask=oracle.find_j_reg_mask(J.detach()), return_verts=True)
    (ref_j * dj).sum().backward()
    eng = v['eng_mod'].RefineEngine(v['dm'], B, flags=v['eng_mod'].FLAG_KEEP_VERTS)
    eng.set_j_regressor(T(v['J']))
    xd, bd = x6d.float().contiguous().to(DEV), betas.float().contiguous().to(DEV)
    joints, verts = eng.find_joints_forward(bd, x6d=xd, return_verts=True)
    assert (verts.cpu().double() - ref_v.detach()).abs().max().item() < 2e-5          # results in FILE order of the vertices
    assert (joints.cpu().double() - ref_j.detach()).abs().max().item() < 2e-5         # north_star bar: 1e-4 m
    dx, db, dJ = eng.find_joints_backward(bd, dj.float().to(DEV), x6d=xd, want_dJ=True)
    assert _relerr(dx, xr.grad) < 2e-4 and _relerr(db, br.grad) < 2e-4 and _relerr(dJ, J.grad) < 2e-4
    assert (dJ.cpu()[T(v['J']) <= 0] == 0).all()


def test_j_step_gradient_vs_oracle(variant):
    v = variant
    B = 130
    batch = v['sm'].synthetic_batch(v['model'], v['J'], B, seed=61)
    x6, betas = T(batch['pose6d']), T(batch['betas'])
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    smpl = oracle.OracleSMPL(v['model'])
    _, dJ_ref, _ = oracle.j_regressor_loss_and_grad(smpl, T(v['J']), x6[:, :1], x6[:, 1:], betas, gt_c)
    eng = v['eng_mod'].RefineEngine(v['dm'], B, flags=v['eng_mod'].FLAG_KEEP_VERTS)
    eng.set_j_regressor(T(v['J']))
    dJ = eng.j_regressor_grad(x6.to(DEV).contiguous(), betas.to(DEV).contiguous(), gt_c.to(DEV).contiguous())
    assert _relerr(dJ, dJ_ref) < 5e-4
    assert torch.equal(dJ.cpu() != 0, dJ_ref != 0)


def _rounding_dominated_entries(records32, records64, rel=5e-3):
    """(B,144) mask of the pose entries whose fp32 ORACLE gradient is dominated by rounding: it differs from the float64 oracle's gradient of
    the same iteration by more than `rel` of its value (a sum of large terms that cancel to ~0; the two oracle trajectories themselves
    differ by ~1e-7 over three iterations, far below `rel`).  Adam's update is lr * m_hat / (sqrt(v_hat) + eps): a gradient entry known to
    a few per cent moves the parameter by a few per cent of a step -- 7.7e-4 was one such entry when round 5 changed a summation order --
    whatever the order of the sums; every other entry sees a rounding difference as ~1e-7 of a step."""
    mask = None
    for r32, r64 in zip(records32, records64):
        g32 = torch.cat([r32['g_orient'], r32['g_pose']], 1).reshape(r32['g_pose'].shape[0], -1).double()
        g64 = torch.cat([r64['g_orient'], r64['g_pose']], 1).reshape(r64['g_pose'].shape[0], -1)
        bad = (g32 - g64).abs() > rel * g64.abs()
        mask = bad if mask is None else (mask | bad)
    return mask


def test_three_iterations_with_pose_discriminator_vs_oracle(variant):
    v = variant
    B = 200
    batch = v['sm'].synthetic_batch(v['model'], v['J'], B, seed=62)
    x6, betas = T(batch['pose6d']), T(batch['betas'])
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    dsd = oracle.formula_state_dict(oracle.DISC_PARAM_SHAPES, seed=0)
    smpl = oracle.OracleSMPL(v['model'])
    records, records64 = [], []
    o, p, b_, _ = oracle.refine_poses(smpl, T(v['J']), x6[:, :1], x6[:, 1:], betas, gt_c, 3, disc_sd=dsd,
                                      record=lambda it, r: records.append(r))
    oracle.refine_poses(oracle.OracleSMPL(v['model'], dtype=torch.float64), T(v['J']).double(), x6[:, :1].double(), x6[:, 1:].double(),
                        betas.double(), gt_c.double(), 3, disc_sd={k: t.double() for k, t in dsd.items()},
                        record=lambda it, r: records64.append(r))
    em = v['eng_mod']
    eng = em.RefineEngine(v['dm'], B, flags=em.FLAG_POSE_DISC)
    eng.set_j_regressor(T(v['J']))
    eng.set_pose_disc(em.flatten_state_dict(dsd, em.DISC_KEYS))
    xd, bd = x6.to(DEV).contiguous(), betas.to(DEV).contiguous()
    m, vv, step = _fresh_state(B)
    eng.refine_run(xd, bd, gt_c.to(DEV).contiguous(), m, vv, step, 1e-2, 3)
    d = (xd.cpu() - torch.cat([o, p], 1)).abs().reshape(B, -1)
    # Adam's first steps are lr * g / (|g| + eps): last-bit differences of the summation order are amplified wherever a gradient
    # entry is ~ 0.  Those entries are identified EXPLICITLY (round 6; round 5 had widened the maximum for every
    # entry when the planner's split-K slab count changed a summation order): where the fp32 oracle's own gradient is off its float64
    # twin by more than 0.5 %.  They keep the wide bound of a tenth of a first Adam step, every other entry the strict 6e-4, and the
    # mean pins the trajectory.
    near = _rounding_dominated_entries(records, records64)
    assert near.float().mean().item() < 0.01, near.float().mean().item()         # a handful of entries, not a blanket
    strict = d[~near].max().item()
    worst = (d * (~near)).argmax().item()
    assert strict < 6e-4, (strict, worst, [torch.cat([r['g_orient'], r['g_pose']], 1).reshape(B, -1).flatten()[worst].item() for r in records])
    if near.any():
        assert d[near].max().item() < 1e-3, d[near].max().item()
    assert d.mean().item() < 5e-6, d.mean().item()
    assert (bd.cpu() - b_).abs().max().item() < 3e-4


def test_benchmarked_batch_4096_strided_subset_vs_oracle(variant):
    """BASELINE configs[2]'s own size: joints of all 4096 poses are finite and pelvis-consistent, and on a strided subset the
    joints, the gradients of a random joint adjoint and 2 refinement iterations equal the oracle's"""
    v = variant
    B, sub = 4096, slice(5, 4096, 128)
    batch = v['sm'].synthetic_batch(v['model'], v['J'], B, seed=63)
    x6, betas = T(batch['pose6d']), T(batch['betas'])
    em = v['eng_mod']
    eng = em.RefineEngine(v['dm'], B, flags=0)
    eng.set_j_regressor(T(v['J']))
    xd, bd = x6.to(DEV).contiguous(), betas.to(DEV).contiguous()
    joints = eng.find_joints_forward(bd, x6d=xd)
    dj = torch.randn(B, 17, 3, generator=torch.Generator().manual_seed(9))
    dx, db, _ = eng.find_joints_backward(bd, dj.to(DEV).contiguous(), x6d=xd)
    assert torch.isfinite(joints).all() and torch.isfinite(dx).all()
    xs, bs = x6[sub].clone().requires_grad_(True), betas[sub].clone().requires_grad_(True)
    smpl = oracle.OracleSMPL(v['model'])
    R = oracle.rot6d_to_rotmat(xs.reshape(-1, 6)).view(-1, 24, 3, 3)
    j = oracle.find_joints(smpl, bs, R[:, :1], R[:, 1:], T(v['J']))
    (j * dj[sub]).sum().backward()
    assert (joints.cpu()[sub] - j.detach()).abs().max().item() < 2e-5
    assert _relerr(dx[sub], xs.grad) < 5e-4 and _relerr(db[sub], bs.grad) < 5e-4
    # two iterations of the loop (joint loss; batch_norm = 4096 in both)
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    o, p, b_, _ = oracle.refine_poses(smpl, T(v['J']), x6[sub, :1], x6[sub, 1:], betas[sub], gt_c[sub], 2, batch_norm=B)
    m, vv, step = _fresh_state(B)
    eng.refine_run(xd, bd, gt_c.to(DEV).contiguous(), m, vv, step, 1e-2, 2)
    d = (xd.cpu()[sub] - torch.cat([o, p], 1)).abs()
    assert d.max().item() < 6e-4 and d.mean().item() < 5e-6, (d.max().item(), d.mean().item())


# ---- JRR_FLAG_SUPPORT_TILES: the joint-loss iteration on the tiles of the regressor's support only ----------------------------
def _tiles_engine(v, B, extra=0):
    em = v['eng_mod']
    eng = em.RefineEngine(v['dm'], B, flags=em.FLAG_KEEP_VERTS | em.FLAG_SUPPORT_TILES | extra)
    eng.set_j_regressor(T(v['J']))
    assert eng.support_tiles() == (False, 216)                    # not before the support has been asked for
    counts, fits = eng.j_support_info()
    assert fits
    on, n = eng.support_tiles()
    listable = v['dm'].info['joint_slots'] in (8, 12) and 'role_bwd' not in v['name']
    assert on == listable, (v['name'], on, n)
    if on:
        assert 0 < n <= min(216, sum(counts)), (n, counts)        # at most one tile per support entry
    if v['name'].startswith('hinted'):                            # the support was stored first, a few tiles of <= 8 joints each
        assert on and n <= 4 * ((v['dm'].info['hinted_vertices_stored_first'] + 31) // 32), (n, v['dm'].info)
    return eng, on


def test_support_tiles_three_iterations_with_pose_discriminator_vs_oracle(variant):
    """every other tile multiplies its vertices by a zero block of the regressor (scripts/utils.py:87-92) and gets a zero vertex
    adjoint: the iteration restricted to the support's tiles against the ORACLE (which computes all 6890 vertices), same bounds as
    the all-tiles test above; variants without the joint-sparse 16-pose backward kernel run all tiles under the same flag"""
    v = variant
    B = 200
    batch = v['sm'].synthetic_batch(v['model'], v['J'], B, seed=62)
    x6, betas = T(batch['pose6d']), T(batch['betas'])
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    dsd = oracle.formula_state_dict(oracle.DISC_PARAM_SHAPES, seed=0)
    smpl = oracle.OracleSMPL(v['model'])
    o, p, b_, _ = oracle.refine_poses(smpl, T(v['J']), x6[:, :1], x6[:, 1:], betas, gt_c, 3, disc_sd=dsd)
    em = v['eng_mod']
    eng, _ = _tiles_engine(v, B, em.FLAG_POSE_DISC)
    eng.set_pose_disc(em.flatten_state_dict(dsd, em.DISC_KEYS))
    xd, bd = x6.to(DEV).contiguous(), betas.to(DEV).contiguous()
    m, vv, step = _fresh_state(B)
    eng.refine_run(xd, bd, gt_c.to(DEV).contiguous(), m, vv, step, 1e-2, 3)
    d = (xd.cpu() - torch.cat([o, p], 1)).abs()
    assert d.max().item() < 6e-4 and d.mean().item() < 5e-6, (d.max().item(), d.mean().item())
    assert (bd.cpu() - b_).abs().max().item() < 3e-4


@pytest.mark.parametrize('every', [1, 2])
def test_support_tiles_with_j_steps_equal_the_all_tiles_run(variant, every):
    """in-call J steps (forward reuse after each) on the support's tiles against the same call on all tiles: poses within the
    suite's geometry bound (another order of the sums over the tiles), the regressor to 1e-6, the same entries moved"""
    v = variant
    em = v['eng_mod']
    B = 130
    batch = v['sm'].synthetic_batch(v['model'], v['J'], B, seed=64)
    gt_c = oracle.move_pelvis(T(batch['gt_j3d'])).to(DEV).contiguous()
    outs = []
    for tiles in (False, True):
        if tiles:
            eng, on = _tiles_engine(v, B)
        else:
            eng = em.RefineEngine(v['dm'], B, flags=em.FLAG_KEEP_VERTS)
            eng.set_j_regressor(T(v['J']))
            eng.j_support_info()
        J = T(v['J']).to(DEV).clone()
        eng.set_j_regressor(J)
        eng.j_support_info()
        Jm, Jv, Js = torch.zeros_like(J), torch.zeros_like(J), torch.zeros(1, dtype=torch.int32, device=DEV)
        xd, bd = T(batch['pose6d']).to(DEV).contiguous(), T(batch['betas']).to(DEV).contiguous()
        m, vv, step = _fresh_state(B)
        eng.refine_run_j_steps(xd, bd, gt_c, m, vv, step, 1e-2, 4, every, J, Jm, Jv, Js, 1e-2)
        eng.refine_run(xd, bd, gt_c, m, vv, step, 1e-2, 1, after_j_step=True)
        outs.append((xd.cpu(), bd.cpu(), J.cpu(), int(Js.item())))
    (xa, ba, Ja, na), (xt, bt, Jt, nt) = outs
    assert na == nt == 4 // every
    d = (xa - xt).abs()
    assert d.max().item() < 6e-4 and d.mean().item() < 5e-6, (d.max().item(), d.mean().item())
    assert (ba - bt).abs().max().item() < 3e-4
    assert (Ja - Jt).abs().max().item() < 1e-6
    assert torch.equal(Ja != T(v['J']), Jt != T(v['J']))


def test_support_wider_than_64_vertices_runs_the_tile_lists(smpl_model_np):
    """the per-vertex iteration (supk.h) is built for a support of at most 64 vertices: a regressor that reads 102 of them keeps the
    tile lists of round 4 -- reported by support_vertices() -- and both forms equal the oracle; with the shipped H36M regressor (56
    vertices) the per-vertex iteration is the one that runs"""
    import conftest
    sm, em = _mod('smpl_model'), _mod('engine')
    rng = np.random.RandomState(11)
    J_wide = np.zeros((17, 6890), np.float32)
    cols = rng.choice(6890, 17 * 6, replace=False).reshape(17, 6)
    for i in range(17):
        J_wide[i, cols[i]] = rng.dirichlet(np.ones(6)).astype(np.float32)
    dm = em.DeviceModel(smpl_model_np, DEV, hint_vertices=np.nonzero((J_wide > 0).any(0))[0])
    B = 70
    batch = sm.synthetic_batch(smpl_model_np, J_wide, B, seed=17)
    x6, betas = T(batch['pose6d']), T(batch['betas'])
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    smpl = oracle.OracleSMPL(smpl_model_np)
    o, p, b_, _ = oracle.refine_poses(smpl, T(J_wide), x6[:, :1], x6[:, 1:], betas, gt_c, 3)
    eng = em.RefineEngine(dm, B, flags=em.FLAG_KEEP_VERTS | em.FLAG_SUPPORT_TILES)
    eng.set_j_regressor(T(J_wide))
    counts, fits = eng.j_support_info()
    assert fits and sum(counts) == 102
    assert eng.support_tiles()[0] == conftest.support_tiles_available()
    assert eng.support_vertices() == (False, 0)                       # 102 > 64: the tile lists
    xd, bd = x6.to(DEV).contiguous(), betas.to(DEV).contiguous()
    m, vv, step = _fresh_state(B)
    eng.refine_run(xd, bd, gt_c.to(DEV).contiguous(), m, vv, step, 1e-2, 3)
    d = (xd.cpu() - torch.cat([o, p], 1)).abs()
    assert d.max().item() < 6e-4 and d.mean().item() < 1e-5, (d.max().item(), d.mean().item())
    # ... and the shipped regressor on the same body engages the per-vertex iteration
    t = conftest.load_golden('j_regressor_triplets.npz')
    J_h36m = sm.j_regressor_from_triplets(t['rows'], t['cols'], t['vals'])
    eng2 = em.RefineEngine(dm, B, flags=em.FLAG_KEEP_VERTS | em.FLAG_SUPPORT_TILES)
    eng2.set_j_regressor(T(J_h36m))
    eng2.j_support_info()
    on, n_sv = eng2.support_vertices()
    import os
    fused_off = os.environ.get('JRR_SUPPORT_FUSED') == '0'
    assert on == (conftest.support_tiles_available() and not fused_off) and (n_sv == int((J_h36m > 0).any(0).sum()) or not on)


@pytest.mark.parametrize('B', [1, 37])
def test_support_tiles_ragged_batches_vs_oracle(smpl_model_np, j_h36m_np, B):
    """batches far below one pose group (padded to 128 columns): three listed iterations + a J step against the oracle, on the body
    uploaded with the vertex-order hint (what optimize.py does)"""
    sm, em = _mod('smpl_model'), _mod('engine')
    dm = em.DeviceModel(smpl_model_np, DEV, hint_vertices=np.nonzero((j_h36m_np > 0).any(0))[0])
    batch = sm.synthetic_batch(smpl_model_np, j_h36m_np, B, seed=90 + B)
    x6, betas = T(batch['pose6d']), T(batch['betas'])
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    smpl = oracle.OracleSMPL(smpl_model_np)
    o, p, b_, _ = oracle.refine_poses(smpl, T(j_h36m_np), x6[:, :1], x6[:, 1:], betas, gt_c, 3)
    eng = em.RefineEngine(dm, B, flags=em.FLAG_KEEP_VERTS | em.FLAG_SUPPORT_TILES)
    J = T(j_h36m_np).to(DEV).clone()
    eng.set_j_regressor(J)
    import conftest
    assert eng.j_support_info()[1] and eng.support_tiles()[0] == conftest.support_tiles_available()
    xd, bd = x6.to(DEV).contiguous(), betas.to(DEV).contiguous()
    m, vv, step = _fresh_state(B)
    eng.refine_run(xd, bd, gt_c.to(DEV).contiguous(), m, vv, step, 1e-2, 3)
    d = (xd.cpu() - torch.cat([o, p], 1)).abs()
    assert d.max().item() < 6e-4 and d.mean().item() < 1e-5, (d.max().item(), d.mean().item())
    assert (bd.cpu() - b_).abs().max().item() < 3e-4
    # the J step's gradient on the refined poses (listed forward), against the oracle on the same poses
    _, dJ_ref, _ = oracle.j_regressor_loss_and_grad(smpl, T(j_h36m_np), xd.cpu()[:, :1], xd.cpu()[:, 1:], bd.cpu(), gt_c)
    dJ = eng.j_regressor_grad(xd, bd, gt_c.to(DEV).contiguous())
    assert _relerr(dJ, dJ_ref) < 5e-4
    assert torch.equal(dJ.cpu() != 0, dJ_ref != 0)


def test_support_tiles_batch_4096_strided_subset_vs_oracle(variant):
    v = variant
    if v['name'] not in ('default', 'capsules', 'wide13', 'skin12', 'hinted'):
        pytest.skip('the benchmarked size on the variants that differ in the listed kernels')
    B, sub = 4096, slice(5, 4096, 128)
    batch = v['sm'].synthetic_batch(v['model'], v['J'], B, seed=63)
    x6, betas = T(batch['pose6d']), T(batch['betas'])
    eng, on = _tiles_engine(v, B)
    assert on
    xd, bd = x6.to(DEV).contiguous(), betas.to(DEV).contiguous()
    smpl = oracle.OracleSMPL(v['model'])
    gt_c = oracle.move_pelvis(T(batch['gt_j3d']))
    o, p, b_, _ = oracle.refine_poses(smpl, T(v['J']), x6[sub, :1], x6[sub, 1:], betas[sub], gt_c[sub], 2, batch_norm=B)
    m, vv, step = _fresh_state(B)
    sq = torch.zeros(B, device=DEV)
    eng.refine_run(xd, bd, gt_c.to(DEV).contiguous(), m, vv, step, 1e-2, 2, sqerr=sq)
    assert torch.isfinite(xd).all() and torch.isfinite(sq).all()
    d = (xd.cpu()[sub] - torch.cat([o, p], 1)).abs()
    assert d.max().item() < 6e-4 and d.mean().item() < 5e-6, (d.max().item(), d.mean().item())


def test_vertex_order_hint_is_invisible_at_the_api(smpl_model_np, j_h36m_np):
    """jrr_model_create_hinted only changes the library's INTERNAL vertex order: vertices come back in the file's order and equal
    the un-hinted model's (same per-vertex arithmetic), joints to rounding (another order of the sum over the vertices); duplicate
    hint entries are ignored, an index outside the mesh is refused"""
    sm, em = _mod('smpl_model'), _mod('engine')
    hint = np.nonzero((j_h36m_np > 0).any(0))[0]
    plain = em.DeviceModel(smpl_model_np, DEV)
    hinted = em.DeviceModel(smpl_model_np, DEV, hint_vertices=np.concatenate([hint, hint[:7]]))
    assert hinted.info['hinted_vertices_stored_first'] == len(hint) and plain.info['hinted_vertices_stored_first'] == 0
    with pytest.raises(Exception):
        em.DeviceModel(smpl_model_np, DEV, hint_vertices=np.array([3, 6890]))
    B = 33
    batch = sm.synthetic_batch(smpl_model_np, j_h36m_np, B, seed=77)
    xd, bd = T(batch['pose6d']).to(DEV).contiguous(), T(batch['betas']).to(DEV).contiguous()
    outs = []
    for dm in (plain, hinted):
        eng = em.RefineEngine(dm, B, flags=em.FLAG_KEEP_VERTS)
        eng.set_j_regressor(T(j_h36m_np))
        outs.append(eng.find_joints_forward(bd, x6d=xd, return_verts=True))
    (j0, v0), (j1, v1) = outs
    assert (v0 - v1).abs().max().item() < 2e-6
    assert (j0 - j1).abs().max().item() < 2e-6


# ---- forward reuse after a J step where BP is an odd multiple of 128 ---------------------------------------------------------
@pytest.mark.parametrize('B', [300, 600])
def test_forward_reuse_after_j_step_at_odd_multiples_of_128(smpl_model_np, j_h36m_np, B):
    """The iteration after a J step re-regresses its joints from the J step's stored vertices into ONE slab of BP pose columns
    (support path).  BP = 384 / 640 is not a multiple of the kernel's 256-pose blocks: every pose column -- including the last
    128 -- must be written.  Against the sequence that repeats the SMPL forward (mode A), and the in-call J steps against both."""
    sm, em = _mod('smpl_model'), _mod('engine')
    dm = em.DeviceModel(smpl_model_np, DEV)
    batch = sm.synthetic_batch(smpl_model_np, j_h36m_np, B, seed=500 + B)
    gt_c = oracle.move_pelvis(T(batch['gt_j3d'])).to(DEV).contiguous()

    def run(mode):
        eng = em.RefineEngine(dm, B, flags=em.FLAG_KEEP_VERTS)
        assert eng.info['BP'] % 256 == 128 and eng.info['BP'] > 128
        J = T(j_h36m_np).to(DEV).clone()
        Jm, Jv, Js = torch.zeros_like(J), torch.zeros_like(J), torch.zeros(1, dtype=torch.int32, device=DEV)
        eng.set_j_regressor(J)
        xd, bd = T(batch['pose6d']).to(DEV).contiguous(), T(batch['betas']).to(DEV).contiguous()
        m, v, step = _fresh_state(B)
        sq = torch.zeros(B, device=DEV)
        if mode == 'in_call':
            eng.refine_run_j_steps(xd, bd, gt_c, m, v, step, 1e-2, 4, 1, J, Jm, Jv, Js, 1e-2, sqerr=sq)
        else:
            for _ in range(4):
                eng.refine_run(xd, bd, gt_c, m, v, step, 1e-2, 1, sqerr=sq, after_j_step=(mode == 'reuse' and int(Js.item()) > 0))
                dJ = eng.j_regressor_grad(xd, bd, gt_c)
                eng.j_step_apply(J, dJ, Jm, Jv, Js, 1e-2)
        return xd.cpu(), bd.cpu(), J.cpu(), sq.cpu()

    a, r, c = run('repeat'), run('reuse'), run('in_call')
    for x, y in zip(r, c):
        assert torch.equal(x, y)                                  # in-call J steps == the host-driven reuse sequence, bit for bit
    # reuse vs repeated forward: the regressor product is summed in another order (Adam-amplified rounding);
    # a stale slab for the last 128 poses would be off by the joints themselves
    d = (a[0] - r[0]).abs()
    assert d.max().item() < 6e-4 and d.mean().item() < 5e-6, (d.max().item(), d.mean().item())
    tail = slice(B - (B % 128 or 128), B)                          # poses in the last 128 columns
    np.testing.assert_allclose(r[3][tail].numpy(), a[3][tail].numpy(), rtol=2e-3, atol=1e-9)
    assert (a[2] - r[2]).abs().max().item() < 5e-5


# ---- the J step with its all-reduce payload restricted to the support -------------------------------------------------------
def test_support_sized_j_step_is_bit_identical_to_the_dense_pair(smpl_model_np, j_h36m_np):
    sm, em = _mod('smpl_model'), _mod('engine')
    dm = em.DeviceModel(smpl_model_np, DEV)
    B = 130
    batch = sm.synthetic_batch(smpl_model_np, j_h36m_np, B, seed=71)
    xd, bd = T(batch['pose6d']).to(DEV).contiguous(), T(batch['betas']).to(DEV).contiguous()
    gt_c = oracle.move_pelvis(T(batch['gt_j3d'])).to(DEV).contiguous()
    outs = []
    for compact in (False, True):
        eng = em.RefineEngine(dm, B, flags=em.FLAG_KEEP_VERTS)
        J = T(j_h36m_np).to(DEV).clone()
        Jm, Jv, Js = torch.zeros_like(J), torch.zeros_like(J), torch.zeros(1, dtype=torch.int32, device=DEV)
        eng.set_j_regressor(J)
        if compact:
            with pytest.raises(Exception):                         # not before the support has been asked for
                eng.j_regressor_grad_support(xd, bd, gt_c, out=torch.zeros(17, 128, device=DEV))
            counts, fits = eng.j_support_info()
            assert fits and sum(counts) == int((j_h36m_np > 0).sum()) == 62
        for _ in range(3):                                         # three J steps: the support may only shrink
            if compact:
                buf = torch.full((17, 128), 7.0, device=DEV)
                eng.j_regressor_grad_support(xd, bd, gt_c, out=buf)
                assert int((buf != 0).sum()) <= 62
                eng.j_step_apply_support(J, buf, Jm, Jv, Js, 1e-2)
            else:
                dJ = eng.j_regressor_grad(xd, bd, gt_c)
                eng.j_step_apply(J, dJ, Jm, Jv, Js, 1e-2)
        joints = eng.find_joints_forward(bd, x6d=xd)
        outs.append((J.cpu(), Jm.cpu(), Jv.cpu(), joints.cpu()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # the support-sized step's forward keeps the vertex tiles of the support only: a regressor from outside between it and the
    # reusing iteration drops the cached forward (the engine refuses instead of regressing joints from vertices it did not store);
    # without such an intervention the reuse works
    eng = em.RefineEngine(dm, B, flags=em.FLAG_KEEP_VERTS)
    J = T(j_h36m_np).to(DEV).clone()
    Jm, Jv, Js = torch.zeros_like(J), torch.zeros_like(J), torch.zeros(1, dtype=torch.int32, device=DEV)
    eng.set_j_regressor(J)
    assert eng.j_support_info()[1]
    x2, b2 = xd.clone(), bd.clone()
    am, av = torch.zeros(B, 154, device=DEV), torch.zeros(B, 154, device=DEV)
    st = torch.zeros(1, dtype=torch.int32, device=DEV)
    buf = torch.zeros(17, 128, device=DEV)
    eng.j_regressor_grad_support(x2, b2, gt_c, out=buf)
    eng.j_step_apply_support(J, buf, Jm, Jv, Js, 1e-2)
    eng.refine_run(x2, b2, gt_c, am, av, st, 1e-3, 1, after_j_step=True)
    eng.j_regressor_grad_support(x2, b2, gt_c, out=buf)
    J2 = T(j_h36m_np).to(DEV).clone()
    J2[5, 4000:4010] = 0.02                                       # support outside the stored tiles
    eng.set_j_regressor(J2)
    with pytest.raises(Exception):
        eng.refine_run(x2, b2, gt_c, am, av, st, 1e-3, 1, after_j_step=True)
    eng.refine_run(x2, b2, gt_c, am, av, st, 1e-3, 1)              # a plain run is fine
    # a regressor whose rows do not fit the lists: fits = 0, the support entry points refuse, the dense pair works
    Jwide = j_h36m_np.copy()
    Jwide[3, :200] = 0.01
    eng = em.RefineEngine(dm, B, flags=em.FLAG_KEEP_VERTS)
    eng.set_j_regressor(T(Jwide).to(DEV))
    counts, fits = eng.j_support_info()
    assert not fits
    with pytest.raises(Exception):
        eng.j_regressor_grad_support(xd, bd, gt_c, out=torch.zeros(17, 128, device=DEV))


# ---- a model file in the licensed distribution's format, end to end ----------------------------------------------------------
def test_model_file_with_chumpy_objects_end_to_end(tmp_path):
    """SMPL('<dir>', batch_size=1) on a file in the distribution's format (scripts/optimize.py:96-99, scripts/smpl.py:7-9):
    loaded without chumpy, uploaded (random file order: the library sorts it, per-tile classes), joints vs the oracle"""
    sm, em = _mod('smpl_model'), _mod('engine')
    body = sm.synthetic_smpl(1234, kind='capsules')
    write_chumpy_style_pickle(body, str(tmp_path / 'SMPL_NEUTRAL.pkl'))
    smpl = _mod('smpl').SMPL(str(tmp_path), batch_size=1, allow_synthetic=False).to(DEV)
    assert not any(k == 'chumpy' or k.startswith('chumpy.') for k in sys.modules)
    assert smpl.provenance.startswith('file:')
    for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights', 'parents', 'faces'):
        assert np.array_equal(smpl.model_np[k], body[k]), k
    B = 21
    J = sm.synthetic_h36m_regressor(body, seed=7, support=8)
    batch = sm.synthetic_batch(body, J, B, seed=3)
    x6, betas = T(batch['pose6d']), T(batch['betas'])
    R = _mod('utils').rot6d_to_rotmat(x6.reshape(-1, 6).to(DEV)).view(B, 24, 3, 3)
    joints = _mod('utils').find_joints(smpl, betas.to(DEV), R[:, :1], R[:, 1:], T(J).to(DEV))
    so = oracle.OracleSMPL(body, dtype=torch.float64)
    Ro = oracle.rot6d_to_rotmat(x6.double().reshape(-1, 6)).view(B, 24, 3, 3)
    ref = oracle.find_joints(so, betas.double(), Ro[:, :1], Ro[:, 1:], T(J).double())
    assert (joints.cpu().double() - ref).abs().max().item() < 2e-5


def test_driver_on_a_model_file_and_a_regressor_file(tmp_path):
    """the reference's two file inputs end to end (/root/reference/scripts/optimize.py:96-99,105-107): `--smpl_dir` with a
    SMPL_NEUTRAL.pkl in the distribution's format (chumpy objects; random vertex order -> internal sort, per-tile classes) and
    `--j_regressor_init` with a (17,6890) .npy; two inner iterations + the outer step (discriminator update, J step) against the
    oracle on the same body; the log record names the file as the body model"""
    sm, argsmod, opt = _mod('smpl_model'), _mod('args'), _mod('optimize')
    body = sm.synthetic_smpl(1234, kind='capsules')
    write_chumpy_style_pickle(body, str(tmp_path / 'SMPL_NEUTRAL.pkl'))
    J0 = sm.synthetic_h36m_regressor(body, seed=7, support=8)
    np.save(tmp_path / 'J_regressor_h36m.npy', J0)
    B = 48
    argsmod._LazyArgs._ns = argsmod.get_args(['--batch_size', str(B), '--inner_iters', '2', '--device', DEV, '--synthetic_batches', '1',
                                               '--smpl_dir', str(tmp_path), '--j_regressor_init', str(tmp_path / 'J_regressor_h36m.npy')])
    res = opt.optimize_pose_refiner(log=lambda r: None)
    rec = res['history'][0]
    assert rec['body_model'] == f"file:{tmp_path / 'SMPL_NEUTRAL.pkl'}" and rec['data'] == 'synthetic'
    # the same batch (the driver's seed 0 -> synthetic_batch seed 0) through the oracle
    full = sm.synthetic_batch(body, J0, B, seed=0)
    torch.manual_seed(0)
    dsd = {k: v.detach() for k, v in _mod('discriminator').Discriminator().state_dict().items()}
    x6 = T(full['pose6d'])
    gt_c = oracle.move_pelvis(T(full['gt_j3d']))
    smpl = oracle.OracleSMPL(body)
    o, p, b_, hist = oracle.refine_poses(smpl, T(J0), x6[:, :1], x6[:, 1:], T(full['betas']), gt_c, 2, disc_sd=dsd)
    assert (res['x6d'].cpu() - torch.cat([o, p], 1)).abs().max().item() < 3e-4
    np.testing.assert_allclose(rec['joint_loss'], hist[-1]['joint_loss'], rtol=2e-3)
    # the J step moved exactly the positive support of the file's regressor
    moved = res['J_regressor'].cpu().numpy() != J0
    assert moved.sum() == (J0 > 0).sum() and not moved[J0 <= 0].any()


def test_silhouette_at_the_reference_constructors_default_size(smpl_model_np, j_h36m_np):
    """Mesh_Renderer() -- image_size = 256, scripts/mesh_renderer.py:25, focal length 5000 / 256 (:52-53) -- beside the 224 of the loop
    (scripts/optimize.py:110): alpha and the winning faces against the restated rasteriser (oracle/silhouette_port.py) at 256 x 256,
    the fused loss / gradient kernel against the stand-alone rasteriser + adjoint, and the module's output shape."""
    from oracle import silhouette_port as sp
    sm, em = _mod('smpl_model'), _mod('engine')
    B = 9
    batch = sm.synthetic_batch(smpl_model_np, j_h36m_np, B, seed=91)
    x6, betas, cam = T(batch['pose6d']), T(batch['betas']), T(batch['cam'])
    smpl_hip = _mod('smpl').SMPL(model=smpl_model_np).to(DEV)
    eng = em.RefineEngine(smpl_hip.device_model, B, flags=em.FLAG_SILHOUETTE | em.FLAG_KEEP_VERTS | em.FLAG_SIL_256)
    assert eng.sil == 256
    eng.set_j_regressor(T(j_h36m_np))
    xd, bd, cd = x6.to(DEV).contiguous(), betas.to(DEV).contiguous(), cam.to(DEV).contiguous()
    _, verts = eng.find_joints_forward(bd, x6d=xd, return_verts=True)
    alpha = eng.silhouette_forward(verts, cd)
    assert alpha.shape == (B, 256, 256)
    p2f = eng.silhouette_pix_to_face().cpu()
    ref, p2f_o = sp.soft_silhouette(verts.cpu(), smpl_model_np['faces'], cam, image_size=256, return_pix_to_face=True)
    ref, p2f_o = ref[:, 0], torch.from_numpy(p2f_o)
    agree = (p2f == p2f_o) & ((alpha.cpu() - ref).abs() < 2e-3)
    covered = (p2f_o >= 0).sum().item()
    assert covered > 4000 * B * (256 / 224) ** 2 * 0.8                      # the same body covers (256 / 224)^2 as many pixels
    assert (~agree).sum().item() < 5e-3 * covered, ((~agree).sum().item(), covered)
    assert torch.equal(p2f >= 0, alpha.cpu() > 0)
    # the fused kernel (in-kernel projection with focal 5000 / 256, fixed-point adjoint) == stand-alone rasteriser + adjoint
    mask = (eng.silhouette_forward(verts, (cd + torch.tensor([0.15, -0.1, 1.0], device=DEV)).contiguous()) > 0).float().contiguous()
    sq_f, dv_f, dc_f = eng.silhouette_loss_grad(xd, bd, cd, mask)
    al = eng.silhouette_forward(verts, cd)
    sq_s = ((al - mask) ** 2).sum((1, 2))
    np.testing.assert_allclose(sq_f.cpu().numpy(), sq_s.cpu().numpy(), rtol=2e-3)
    dv_s, dc_s = eng.silhouette_backward(((al - mask) * (2.0 * 100.0 / (B * 256 * 256))).contiguous())
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()      # noqa: E731
    assert rel(dv_f, dv_s) < 2e-5 and rel(dc_f, dc_s) < 2e-5
    # the module with the reference constructor's default
    mr = _mod('mesh_renderer')
    out = mr.Mesh_Renderer(smpl=smpl_hip)({'cam': cd}, verts * verts.new_tensor([-2.0, -2.0, 2.0]))
    assert out.shape == (B, 4, 256, 256) and torch.equal(out[:, 3], al)
    with pytest.raises(NotImplementedError):
        mr.Mesh_Renderer(300, smpl_hip)


@pytest.mark.parametrize('S', [96, 160])
def test_mesh_renderer_at_other_image_sizes(smpl_model_np, j_h36m_np, S):
    """Mesh_Renderer(image_size) takes any size in the reference (scripts/mesh_renderer.py:25,34-38; focal length 5000 / size, :52-53):
    the stand-alone rasteriser and its adjoint at the multiples of 32 up to 256 -- alpha and the winning faces against the restated
    rasteriser, the adjoint against the oracle's autograd through the module, the in-loop term refused at sizes it is not built for"""
    from oracle import silhouette_port as sp
    sm, em, mr = _mod('smpl_model'), _mod('engine'), _mod('mesh_renderer')
    B = 5
    batch = sm.synthetic_batch(smpl_model_np, j_h36m_np, B, seed=93)
    x6, betas, cam = T(batch['pose6d']), T(batch['betas']), T(batch['cam'])
    smpl_hip = _mod('smpl').SMPL(model=smpl_model_np).to(DEV)
    eng = em.RefineEngine(smpl_hip.device_model, B, flags=em.FLAG_SILHOUETTE | em.FLAG_KEEP_VERTS | em.FLAG_SIL_SIZE(S))
    assert eng.sil == S
    eng.set_j_regressor(T(j_h36m_np))
    xd, bd, cd = x6.to(DEV).contiguous(), betas.to(DEV).contiguous(), cam.to(DEV).contiguous()
    _, verts = eng.find_joints_forward(bd, x6d=xd, return_verts=True)
    alpha = eng.silhouette_forward(verts, cd)
    assert alpha.shape == (B, S, S)
    p2f = eng.silhouette_pix_to_face().cpu()
    ref, p2f_o = sp.soft_silhouette(verts.cpu(), smpl_model_np['faces'], cam, image_size=S, return_pix_to_face=True)
    ref, p2f_o = ref[:, 0], torch.from_numpy(p2f_o)
    agree = (p2f == p2f_o) & ((alpha.cpu() - ref).abs() < 2e-3)
    covered = (p2f_o >= 0).sum().item()
    assert covered > 4000 * B * (S / 224) ** 2 * 0.8
    assert (~agree).sum().item() < 5e-3 * covered, ((~agree).sum().item(), covered)
    assert torch.equal(p2f >= 0, alpha.cpu() > 0)
    # the module: forward + backward (stand-alone adjoint kernel) against the oracle's autograd on the pixels both rasterisers agree on
    w = torch.randn(B, S, S, generator=torch.Generator().manual_seed(S)) * agree.float()
    vd = verts.clone().requires_grad_(True)
    out = mr.Mesh_Renderer(S, smpl_hip)({'cam': cd}, vd * vd.new_tensor([-2.0, -2.0, 2.0]))
    assert out.shape == (B, 4, S, S) and torch.equal(out[:, 3].detach(), alpha)
    (out[:, 3] * w.to(DEV)).sum().backward()
    vo = verts.cpu().clone().requires_grad_(True)
    img = sp.soft_silhouette(vo, smpl_model_np['faces'], cam, image_size=S)
    (img[:, 0] * w).sum().backward()
    rel = ((vd.grad.cpu().double() - vo.grad.double()).norm() / vo.grad.double().norm()).item()
    assert rel < 2e-2, rel
    # the silhouette term INSIDE the loop exists at 224 and 256 only
    lib_mod = _mod('_lib')
    cm, cv = torch.zeros(B, 3, device=DEV), torch.zeros(B, 3, device=DEV)
    with pytest.raises(lib_mod.JrrError, match='224'):
        eng.set_silhouette((alpha > 0).float().contiguous(), cd, cm, cv)
    with pytest.raises(ValueError):
        em.FLAG_SIL_SIZE(100)
