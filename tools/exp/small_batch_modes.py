"""The same 100-iteration jrr_refine_run on the SAME 256 poses, twenty times: is the fast / slow split of the reference-default block a
property of the data, of one kernel, or of the device's state?  Prints each call's wall time and, with profiling on, the per-launch
means of a fast and of a slow call."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
PKG = 'joint-regressor-refinement_amd'
sm = importlib.import_module(PKG + '.smpl_model'); eng_mod = importlib.import_module(PKG + '.engine')
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N_IT = int(sys.argv[2]) if len(sys.argv) > 2 else 100
TILES = len(sys.argv) > 3 and sys.argv[3] == 'tiles'
model_np = sm.synthetic_smpl(1234); J_np = sm.default_h36m_regressor()
dm = eng_mod.DeviceModel(model_np, dev)
flat, _ = bench.default_disc_flat(0)
batch = sm.synthetic_batch(model_np, J_np, B, seed=1000)
e = eng_mod.RefineEngine(dm, B, flags=eng_mod.FLAG_POSE_DISC | eng_mod.FLAG_KEEP_VERTS | (eng_mod.FLAG_SUPPORT_TILES if TILES else 0))
e.set_j_regressor(torch.from_numpy(J_np).to(dev)); e.set_pose_disc(flat.to(dev))
if TILES:
    e.j_support_info(); print('support tiles:', e.support_tiles())
x0 = torch.from_numpy(batch['pose6d']).to(dev).contiguous(); b0 = torch.from_numpy(batch['betas']).to(dev).contiguous()
gt = torch.from_numpy(batch['gt_j3d']); gt_c = (gt - gt[:, :1]).to(dev).contiguous()
times, profs = [], []
for rep in range(24):
    x, b = x0.clone(), b0.clone()
    m = torch.zeros(B, 154, device=dev); v = torch.zeros(B, 154, device=dev); st = torch.zeros(1, dtype=torch.int32, device=dev)
    e.set_profiling(rep >= 12)
    if rep % 3 == 2:
        time.sleep(0.05)                      # an idle gap before every third call
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e.refine_run(x, b, gt_c, m, v, st, 1e-2, N_IT)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    times.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    if rep >= 12:
        profs.append((1e3 * (t2 - t0), {k: round(t, 4) for k, (t, n) in e.profile_read().items() if n}))
print('host enqueue / total ms per call:', ' '.join('%.1f/%.1f' % t for t in times))
profs.sort(key=lambda p: p[0])
print('fastest profiled call', profs[0]); print('slowest profiled call', profs[-1])
