"""Time the fused inner loop alone (config 3, batch 4096) and print per-kernel HIP-event means.
usage: [ENV=...] python tools/exp/loop_time.py [B] [iters]"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
PKG = 'joint-regressor-refinement_amd'
eng_mod = importlib.import_module(PKG + '.engine'); sm = importlib.import_module(PKG + '.smpl_model')
disc = importlib.import_module(PKG + '.discriminator')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
model = sm.synthetic_smpl(1234); J = sm.default_h36m_regressor()
batch = sm.synthetic_batch(model, J, 512, seed=1)
rep = B // 512
T = lambda k: torch.from_numpy(batch[k]).repeat(rep, *([1] * (batch[k].ndim - 1))).cuda().contiguous()
dm = eng_mod.DeviceModel(model, 'cuda:0')
eng = eng_mod.RefineEngine(dm, B, flags=eng_mod.FLAG_POSE_DISC)
eng.set_j_regressor(torch.from_numpy(J))
torch.manual_seed(0); eng.set_pose_disc(disc.Discriminator().flat_parameters())
x, b = T('pose6d'), T('betas'); gt = T('gt_j3d'); gt = (gt - gt[:, :1]).contiguous()
m, v = torch.zeros(B, 154, device='cuda'), torch.zeros(B, 154, device='cuda'); st = torch.zeros(1, dtype=torch.int32, device='cuda')
eng.refine_run(x, b, gt, m, v, st, 1e-2, 10); torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); eng.refine_run(x, b, gt, m, v, st, 1e-2, n); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
eng.set_profiling(True); eng.refine_run(x, b, gt, m, v, st, 1e-2, 20); prof = eng.profile_read(); eng.set_profiling(False)
tag = ' '.join(f'{k}={v}' for k, v in os.environ.items() if k.startswith('JRR_'))
print(f'[{tag}] B={B}: median {sorted(ts)[2]:.4f} ms/iter (min {min(ts):.4f}); ' + ' '.join(f'{k}={t:.4f}' for k, (t, c) in prof.items() if c))
import ctypes
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), PKG, 'libjrr_hip.so'))
if hasattr(lib, 'jrr_debug_read'):
    buf = (ctypes.c_longlong * 16)(); lib.jrr_debug_read(buf); print("phases:", list(buf))
