"""Experiment: does splitting the batch into G independent pose groups on G streams hide the latency-bound
per-pose kernels behind the other group's MFMA kernels?"""
import importlib, sys, time, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
PKG = 'joint-regressor-refinement_amd'
sm = importlib.import_module(PKG + '.smpl_model'); eng_mod = importlib.import_module(PKG + '.engine')
dev = torch.device('cuda:0')
B = 4096
model_np = sm.synthetic_smpl(1234); J_np = sm.default_h36m_regressor()
batch = sm.synthetic_batch(model_np, J_np, B, seed=1000)
dm = eng_mod.DeviceModel(model_np, dev)
flat, _ = bench.default_disc_flat(0)
x6d = torch.from_numpy(batch['pose6d']).to(dev).contiguous(); betas = torch.from_numpy(batch['betas']).to(dev).contiguous()
gt = torch.from_numpy(batch['gt_j3d']); gt_c = (gt - gt[:, :1]).to(dev).contiguous()
for G in (1, 2, 4):
    Bg = B // G
    engs, streams, state = [], [], []
    for g in range(G):
        e = eng_mod.RefineEngine(dm, Bg, batch_norm=B, flags=eng_mod.FLAG_POSE_DISC)
        e.set_j_regressor(torch.from_numpy(J_np).to(dev)); e.set_pose_disc(flat.to(dev))
        engs.append(e); streams.append(torch.cuda.Stream())
        sl = slice(g * Bg, (g + 1) * Bg)
        state.append((x6d[sl].clone(), betas[sl].clone(), gt_c[sl].contiguous(), torch.zeros(Bg, 154, device=dev), torch.zeros(Bg, 154, device=dev),
                      torch.zeros(1, dtype=torch.int32, device=dev)))
    def run(n):
        for g in range(G):
            with torch.cuda.stream(streams[g]):
                x, b, t, m, v, st = state[g]
                engs[g].refine_run(x, b, t, m, v, st, 1e-2, n)
    torch.cuda.synchronize(); run(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(30); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'G={G}: {dt / 30 * 1e3:.3f} ms/iter  {30 / dt:.1f} it/s')
