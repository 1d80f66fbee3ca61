"""Experiment (round 6): the default (support-vertex) iteration on TWO engines of B/2 poses on two HIP streams of one process -- poses are
independent, so half A's k_sup_step (128 -> 64 workgroups, latency-bound) can overlap half B's discriminator GEMMs with no cross-stream
event at all -- against ONE engine of B poses.  usage: python tools/exp/two_stream_halves.py [B] [iters]"""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device('cuda:0')
model = sm.synthetic_smpl(1234)
J_np = sm.default_h36m_regressor()
batch = sm.synthetic_batch(model, J_np, B, seed=5)
dm = em.DeviceModel(model, dev, hint_vertices=np.nonzero((J_np > 0).any(0))[0])
disc = torch.randn(1840153, device=dev) * 0.02
flags = em.FLAG_KEEP_VERTS | em.FLAG_POSE_DISC | em.FLAG_SUPPORT_TILES


def make(lo, hi, stream):
    n = hi - lo
    with torch.cuda.stream(stream):
        eng = em.RefineEngine(dm, n, batch_norm=B, flags=flags)
        eng.set_j_regressor(torch.from_numpy(J_np).to(dev))
        eng.set_pose_disc(disc)
        eng.j_support_info()
        x = torch.from_numpy(batch['pose6d'][lo:hi]).to(dev).contiguous()
        b = torch.from_numpy(batch['betas'][lo:hi]).to(dev).contiguous()
        gt = torch.from_numpy(batch['gt_j3d'][lo:hi]).to(dev)
        gt = (gt - gt[:, :1]).contiguous()
        st = dict(eng=eng, x=x, b=b, gt=gt, m=torch.zeros(n, 154, device=dev), v=torch.zeros(n, 154, device=dev),
                  step=torch.zeros(1, dtype=torch.int32, device=dev), stream=stream)
    return st


def run(parts, n):
    for p in parts:
        with torch.cuda.stream(p['stream']):
            p['eng'].refine_run(p['x'], p['b'], p['gt'], p['m'], p['v'], p['step'], 1e-2, n)


def timed(parts, n, reps=5):
    run(parts, 10)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        run(parts, n)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[len(ts) // 2]


one = [make(0, B, torch.cuda.Stream())]
print('one engine of', B, 'poses:            %.4f ms per iteration' % timed(one, iters))
x_one = one[0]['x'].clone()
del one
torch.cuda.empty_cache()
two = [make(0, B // 2, torch.cuda.Stream()), make(B // 2, B, torch.cuda.Stream())]
print('two engines of', B // 2, 'on two streams: %.4f ms per iteration' % timed(two, iters))
same = [make(0, B // 2, torch.cuda.current_stream()), make(B // 2, B, torch.cuda.current_stream())]
print('two engines of', B // 2, 'on ONE stream:   %.4f ms per iteration' % timed(same, iters))
# identical results: the same number of iterations ran on each (10 + 5 x iters); per-pose arithmetic does not depend on the split
x_two = torch.cat([two[0]['x'], two[1]['x']])
print('max |x(two streams) - x(one engine)| =', (x_two - x_one).abs().max().item())
