"""Create / use / destroy models and engines repeatedly and watch the free device memory (leak check)."""
import gc, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
PKG = 'joint-regressor-refinement_amd'
eng_mod = importlib.import_module(PKG + '.engine'); sm = importlib.import_module(PKG + '.smpl_model')
disc = importlib.import_module(PKG + '.discriminator')
model = sm.synthetic_smpl(1234); J = sm.default_h36m_regressor()
free0 = None
for it in range(30):
    B = [1, 37, 129, 512][it % 4]
    batch = sm.synthetic_batch(model, J, B, seed=it)
    T = lambda k: torch.from_numpy(batch[k]).cuda().contiguous()
    dm = eng_mod.DeviceModel(model, 'cuda:0')
    flags = eng_mod.FLAG_POSE_DISC | eng_mod.FLAG_KEEP_VERTS | (eng_mod.FLAG_SILHOUETTE if it % 2 else 0) | (eng_mod.FLAG_FOLDED if it % 3 == 0 else 0)
    if it % 5 in (1, 2):      # the default mode of optimize.py: support tiles / support vertices (k_sup_step), round 6
        flags |= eng_mod.FLAG_SUPPORT_TILES
    eng = eng_mod.RefineEngine(dm, B, flags=flags)
    eng.set_j_regressor(torch.from_numpy(J))
    if flags & eng_mod.FLAG_SUPPORT_TILES:
        eng.j_support_info()
    eng.set_pose_disc(disc.Discriminator().flat_parameters())
    x, b = T('pose6d'), T('betas'); gt = T('gt_j3d'); gt = (gt - gt[:, :1]).contiguous()
    m, v = torch.zeros(B, 154, device='cuda'), torch.zeros(B, 154, device='cuda'); st = torch.zeros(1, dtype=torch.int32, device='cuda')
    eng.refine_run(x, b, gt, m, v, st, 1e-2, 3)
    dJ = eng.j_regressor_grad(x, b, gt)
    torch.cuda.synchronize()
    assert torch.isfinite(x).all() and torch.isfinite(dJ[0] if isinstance(dJ, tuple) else dJ).all()
    del eng, dm, x, b, gt, m, v, st, dJ
    gc.collect(); torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if it == 3: free0 = free
    if it >= 3 and it % 4 == 3: print(f'iteration {it}: free {free / 2**20:.0f} MiB (delta since iteration 3: {(free - free0) / 2**20:+.1f} MiB)')
print('ok')
