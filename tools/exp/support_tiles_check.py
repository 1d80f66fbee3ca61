"""JRR_FLAG_SUPPORT_TILES against the dense-vertex iteration on one box: results, tile count, ms per iteration.
usage: python tools/exp/support_tiles_check.py [B] [iters]"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = 'joint-regressor-refinement_amd'
sm = importlib.import_module(PKG + '.smpl_model')
em = importlib.import_module(PKG + '.engine')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device('cuda:0')
model = sm.synthetic_smpl(1234)
J_np = sm.default_h36m_regressor()
batch = sm.synthetic_batch(model, J_np, B, seed=5)
dm = em.DeviceModel(model, dev)
import numpy as np
dm_hint = em.DeviceModel(model, dev, hint_vertices=np.nonzero((J_np > 0).any(0))[0])
print('hinted model', dm_hint.info)
disc = torch.randn(em.DISC_PARAMS if hasattr(em, 'DISC_PARAMS') else 1840153, device=dev) * 0.02
res = {}
for name, fl in (('dense', 0), ('tiles', em.FLAG_SUPPORT_TILES), ('hinted', em.FLAG_SUPPORT_TILES)):
    eng = em.RefineEngine(dm_hint if name == 'hinted' else dm, B, flags=em.FLAG_KEEP_VERTS | em.FLAG_POSE_DISC | fl)
    J = torch.from_numpy(J_np).to(dev).clone()
    eng.set_j_regressor(J)
    eng.set_pose_disc(disc)
    counts, fits = eng.j_support_info()
    print(name, 'fits', fits, 'support', sum(counts), 'tiles', eng.support_tiles())
    x = torch.from_numpy(batch['pose6d']).to(dev).contiguous()
    b = torch.from_numpy(batch['betas']).to(dev).contiguous()
    gt = torch.from_numpy(batch['gt_j3d']).to(dev)
    gt = (gt - gt[:, :1]).contiguous()
    m, v = torch.zeros(B, 154, device=dev), torch.zeros(B, 154, device=dev)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    Jm, Jv, Js = torch.zeros_like(J), torch.zeros_like(J), torch.zeros(1, dtype=torch.int32, device=dev)
    eng.refine_run_j_steps(x, b, gt, m, v, st, 1e-2, 6, 2, J, Jm, Jv, Js, 1e-2)
    torch.cuda.synchronize()
    res[name] = (x.clone(), b.clone(), J.clone())
    eng.refine_run(x, b, gt, m, v, st, 1e-2, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.refine_run(x, b, gt, m, v, st, 1e-2, iters)
    torch.cuda.synchronize()
    print(name, 'ms/iteration', (time.perf_counter() - t0) / iters * 1e3)
    eng.set_profiling(True)
    eng.refine_run(x, b, gt, m, v, st, 1e-2, 10)
    print(name, {k: round(t, 4) for k, (t, n) in eng.profile_read().items() if n})
    eng.set_profiling(False)
for other in ('tiles', 'hinted'):
    for i, k in enumerate(('x6d', 'betas', 'J')):
        d = (res['dense'][i] - res[other][i]).abs()
        print(other, k, 'max abs diff', d.max().item(), 'mean', d.mean().item())
