import importlib, sys, time, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = 'joint-regressor-refinement_amd'
sm = importlib.import_module(PKG + '.smpl_model'); eng_mod = importlib.import_module(PKG + '.engine')
dev = torch.device('cuda:0'); B = 4096
model_np = sm.synthetic_smpl(1234); J_np = sm.default_h36m_regressor()
batch = sm.synthetic_batch(model_np, J_np, B, seed=1000)
dm = eng_mod.DeviceModel(model_np, dev)
e = eng_mod.RefineEngine(dm, B, flags=eng_mod.FLAG_KEEP_VERTS)
J = torch.from_numpy(J_np).to(dev)
e.set_j_regressor(J)
x = torch.from_numpy(batch['pose6d']).to(dev).contiguous(); b = torch.from_numpy(batch['betas']).to(dev).contiguous()
gt = torch.from_numpy(batch['gt_j3d']); gt_c = (gt - gt[:, :1]).to(dev).contiguous()
Jm, Jv = torch.zeros_like(J), torch.zeros_like(J); st = torch.zeros(1, dtype=torch.int32, device=dev)
def T(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('j_regressor_grad ms', T(lambda: e.j_regressor_grad(x, b, gt_c)))
dJ = e.j_regressor_grad(x, b, gt_c)
print('adam ms', T(lambda: eng_mod.adam_step(J, dJ, Jm, Jv, st, 1e-2)))
print('set_j_regressor ms', T(lambda: e.set_j_regressor(J)))
print('find_joints_forward ms', T(lambda: e.find_joints_forward(b, x6d=x)))
print('find_joints_forward+verts ms', T(lambda: e.find_joints_forward(b, x6d=x, return_verts=True)))
