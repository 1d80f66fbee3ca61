"""How launch-bound is the inner loop?  Host enqueue time of jrr_refine_run (returns after the last launch) vs GPU time."""
import importlib, sys, time, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
PKG = 'joint-regressor-refinement_amd'
sm = importlib.import_module(PKG + '.smpl_model'); eng_mod = importlib.import_module(PKG + '.engine')
dev = torch.device('cuda:0')
model_np = sm.synthetic_smpl(1234); J_np = sm.default_h36m_regressor()
dm = eng_mod.DeviceModel(model_np, dev)
flat, _ = bench.default_disc_flat(0)
for B in (128, 1024, 4096):
    batch = sm.synthetic_batch(model_np, J_np, B, seed=1000)
    e = eng_mod.RefineEngine(dm, B, flags=eng_mod.FLAG_POSE_DISC | eng_mod.FLAG_KEEP_VERTS)
    e.set_j_regressor(torch.from_numpy(J_np).to(dev)); e.set_pose_disc(flat.to(dev))
    x = torch.from_numpy(batch['pose6d']).to(dev).contiguous(); b = torch.from_numpy(batch['betas']).to(dev).contiguous()
    gt = torch.from_numpy(batch['gt_j3d']); gt_c = (gt - gt[:, :1]).to(dev).contiguous()
    m = torch.zeros(B, 154, device=dev); v = torch.zeros(B, 154, device=dev); st = torch.zeros(1, dtype=torch.int32, device=dev)
    e.refine_run(x, b, gt_c, m, v, st, 1e-2, 10); torch.cuda.synchronize()
    n = 100
    t0 = time.perf_counter(); e.refine_run(x, b, gt_c, m, v, st, 1e-2, n); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'B={B}: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/iter, total {1e3 * (t2 - t0) / n:.3f} ms/iter')
