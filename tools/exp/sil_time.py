import importlib, sys, time, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = 'joint-regressor-refinement_amd'
sm = importlib.import_module(PKG + '.smpl_model'); eng_mod = importlib.import_module(PKG + '.engine')
dev = torch.device('cuda:0'); B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
model_np = sm.synthetic_smpl(1234); J_np = sm.default_h36m_regressor()
batch = sm.synthetic_batch(model_np, J_np, B, seed=1000)
dm = eng_mod.DeviceModel(model_np, dev)
e = eng_mod.RefineEngine(dm, B, flags=eng_mod.FLAG_SILHOUETTE)
e.set_j_regressor(torch.from_numpy(J_np).to(dev))
x = torch.from_numpy(batch['pose6d']).to(dev).contiguous(); b = torch.from_numpy(batch['betas']).to(dev).contiguous()
cam = torch.from_numpy(batch['cam']).to(dev).contiguous()
_, verts = e.find_joints_forward(b, x6d=x, return_verts=True)
for _ in range(2): a = e.silhouette_forward(verts, cam)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): a = e.silhouette_forward(verts, cam)
torch.cuda.synchronize(); print('B', B, 'silhouette_forward ms', (time.perf_counter() - t0) / 5 * 1e3, 'covered/pose', float((a > 0).float().sum() / B))
