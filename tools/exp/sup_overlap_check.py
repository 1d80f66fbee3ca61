"""The default (support-vertex) iteration with the pose discriminator at one batch size: ms per iteration, and the poses after
6 iterations with J steps + 40 more written to a file -- run once plain and once under JRR_SUP_OVERLAP=1, then compare the files
(tools/exp/sup_overlap_ab.sh).   usage: python tools/exp/sup_overlap_check.py <B> <iters> <out.npy>"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = 'joint-regressor-refinement_amd'
sm = importlib.import_module(PKG + '.smpl_model')
em = importlib.import_module(PKG + '.engine')
B, iters, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dev = torch.device('cuda:0')
model = sm.synthetic_smpl(1234)
J_np = sm.default_h36m_regressor()
batch = sm.synthetic_batch(model, J_np, B, seed=5)
dm = em.DeviceModel(model, dev, hint_vertices=np.nonzero((J_np > 0).any(0))[0])
torch.manual_seed(3)
disc = torch.randn(1840153, device=dev) * 0.02
eng = em.RefineEngine(dm, B, flags=em.FLAG_KEEP_VERTS | em.FLAG_POSE_DISC | em.FLAG_SHAPE_DISC | em.FLAG_SUPPORT_TILES)
J = torch.from_numpy(J_np).to(dev).clone()
eng.set_j_regressor(J)
eng.set_pose_disc(disc)
eng.set_shape_disc(torch.randn(em.SHAPE_DISC_PARAMS, device=dev) * 0.1)
counts, fits = eng.j_support_info()
x = torch.from_numpy(batch['pose6d']).to(dev).contiguous()
b = torch.from_numpy(batch['betas']).to(dev).contiguous()
gt = torch.from_numpy(batch['gt_j3d']).to(dev)
gt = (gt - gt[:, :1]).contiguous()
m, v = torch.zeros(B, 154, device=dev), torch.zeros(B, 154, device=dev)
st = torch.zeros(1, dtype=torch.int32, device=dev)
Jm, Jv, Js = torch.zeros_like(J), torch.zeros_like(J), torch.zeros(1, dtype=torch.int32, device=dev)
eng.refine_run_j_steps(x, b, gt, m, v, st, 1e-2, 6, 2, J, Jm, Jv, Js, 1e-2)
eng.refine_run(x, b, gt, m, v, st, 1e-2, 40)
torch.cuda.synchronize()
np.save(out, np.concatenate([x.cpu().numpy().ravel(), b.cpu().numpy().ravel(), J.cpu().numpy().ravel(), st.cpu().numpy().astype(np.float32)]))
best = 1e9
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.refine_run(x, b, gt, m, v, st, 1e-2, iters)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / iters * 1e3)
print('B=%d overlap=%s support=%s  ms/iteration %.4f (best of 5 x %d)' % (B, os.environ.get('JRR_SUP_OVERLAP', '0'), eng.support_vertices(), best, iters))
