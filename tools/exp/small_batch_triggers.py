"""Which host-side activity BETWEEN two 100-iteration jrr_refine_run calls at 256 poses makes some of the calls read 90 ms instead of 25?
(the driver shows it, the bare loop does not: tools/exp/small_batch_modes.py).  One candidate per run:
    python tools/exp/small_batch_triggers.py none|nocpu|spin|h2d|d2h|zeros|sleep|all"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
PKG = 'joint-regressor-refinement_amd'
sm = importlib.import_module(PKG + '.smpl_model'); eng_mod = importlib.import_module(PKG + '.engine')
dev = torch.device('cuda:0')
what = sys.argv[1] if len(sys.argv) > 1 else 'none'
B = 256
model_np = sm.synthetic_smpl(1234); J_np = sm.default_h36m_regressor()
dm = eng_mod.DeviceModel(model_np, dev, hint_vertices=np.nonzero((J_np > 0).any(0))[0])
flat, _ = bench.default_disc_flat(0)
e = eng_mod.RefineEngine(dm, B, flags=eng_mod.FLAG_POSE_DISC | eng_mod.FLAG_KEEP_VERTS | eng_mod.FLAG_SUPPORT_TILES)
e.set_j_regressor(torch.from_numpy(J_np).to(dev)); e.set_pose_disc(flat.to(dev)); e.j_support_info()
big = torch.zeros(2_000_000, device=dev)
times = []
for rep in range(30):
    if rep == 0 or what != 'nocpu':
        batch = sm.synthetic_batch(model_np, J_np, B, seed=1000 + rep)       # ~0.1 s of multi-threaded numpy on the host
    if what == 'spin':
        t_ = time.perf_counter()
        while time.perf_counter() - t_ < 0.1: pass                         # a single busy host thread instead
    if what in ('h2d', 'all'):
        x = torch.from_numpy(batch['pose6d']).to(dev).float().contiguous(); b = torch.from_numpy(batch['betas']).to(dev).float().contiguous()
        gt = torch.from_numpy(batch['gt_j3d']).to(dev).float(); gt_c = (gt - gt[:, :1]).contiguous()
    elif rep == 0 or what == 'none' or True:
        if rep == 0:
            x0 = torch.from_numpy(batch['pose6d']).to(dev).contiguous(); b0 = torch.from_numpy(batch['betas']).to(dev).contiguous()
            g0 = torch.from_numpy(batch['gt_j3d']).to(dev); gt_c = (g0 - g0[:, :1]).contiguous()
        x, b = x0.clone(), b0.clone()
    if what in ('zeros', 'all'):
        m = torch.zeros(B, 154, device=dev); v = torch.zeros(B, 154, device=dev); st = torch.zeros(1, dtype=torch.int32, device=dev)
    else:
        if rep == 0:
            m0 = torch.zeros(B, 154, device=dev); v0 = torch.zeros(B, 154, device=dev); s0 = torch.zeros(1, dtype=torch.int32, device=dev)
        m0.zero_(); v0.zero_(); s0.zero_(); m, v, st = m0, v0, s0
    if what in ('sleep', 'all'):
        time.sleep(0.02)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e.refine_run(x, b, gt_c, m, v, st, 1e-2, 100)
    torch.cuda.synchronize(); times.append(1e3 * (time.perf_counter() - t0))
    if what in ('d2h', 'all'):
        big.cpu()
print(what, ' '.join('%.0f' % t for t in times))
