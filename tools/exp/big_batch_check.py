"""Maximum-size check (not in the test-suite: 10 GB workspace): a 32768-pose engine (single-round geometry with 2 vertex
chunks, paired 5 : 4) against 512-pose engines on the same random inputs.  Run on the GPU box: python tools/exp/big_batch_check.py"""
import importlib, sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
PKG = 'joint-regressor-refinement_amd'
sm = importlib.import_module(PKG + '.smpl_model'); em = importlib.import_module(PKG + '.engine')
dev = torch.device('cuda:0')
model = sm.synthetic_smpl(1234); J = sm.default_h36m_regressor()
dm = em.DeviceModel(model, dev)
B = 32768
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 24, 6, generator=g).to(dev) * 0.5 + torch.tensor([1., 0, 0, 1, 0, 0], device=dev)
b = torch.randn(B, 10, generator=g).to(dev)
big = em.RefineEngine(dm, B); big.set_j_regressor(torch.from_numpy(J).to(dev))
print('geometry', big.info)
jb = big.find_joints_forward(b, x6d=x.contiguous())
small = em.RefineEngine(dm, 512, batch_norm=B); small.set_j_regressor(torch.from_numpy(J).to(dev))
worst = 0.0
for k in (0, 17, 63):
    sl = slice(k * 512, (k + 1) * 512)
    js = small.find_joints_forward(b[sl].contiguous(), x6d=x[sl].contiguous())
    worst = max(worst, (jb[sl] - js).abs().max().item())
print('max |joints(32768 engine) - joints(512 engine)| =', worst)
assert worst < 5e-6
