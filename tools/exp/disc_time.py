"""Time the pose-discriminator branch alone (forward + input gradient, stand-alone: 6 launches) at batch 4096.
usage: python tools/exp/disc_time.py [B]"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
PKG = 'joint-regressor-refinement_amd'
eng_mod = importlib.import_module(PKG + '.engine')
disc = importlib.import_module(PKG + '.discriminator')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
flat = disc.Discriminator().flat_parameters().cuda()
eng = eng_mod.RefineEngine(None, B, flags=eng_mod.FLAG_POSE_DISC, device='cuda:0')
eng.set_pose_disc(flat)
x = torch.randn(B, 24, 6, device='cuda') * 0.6
for _ in range(5):
    eng.pose_disc_forward(x); eng.pose_disc_backward_input(x, 10.0, 1.0)
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for _ in range(n):
    eng.pose_disc_forward(x); eng.pose_disc_backward_input(x, 10.0, 1.0)
torch.cuda.synchronize()
print(f'B={B}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per forward + input-gradient (incl. the operator-level finish kernel)')
