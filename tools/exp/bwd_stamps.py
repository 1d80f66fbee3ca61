"""shader-clock stamps of k_lbs_bwd16 (variant library tools/probe/libjrr_stamps.so copied over the in-tree one): prologue, per
tile, flush of four workgroups"""
import ctypes, importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
PKG = 'joint-regressor-refinement_amd'
sm = importlib.import_module(PKG + '.smpl_model'); em = importlib.import_module(PKG + '.engine')
lib = importlib.import_module(PKG + '._lib').load()
B = 4096
model = sm.synthetic_smpl(1234); J = sm.default_h36m_regressor(); batch = sm.synthetic_batch(model, J, B, seed=1)
dm = em.DeviceModel(model, 'cuda:0'); eng = em.RefineEngine(dm, B, flags=0)
eng.set_j_regressor(torch.from_numpy(J))
x = torch.from_numpy(batch['pose6d']).cuda(); b = torch.from_numpy(batch['betas']).cuda()
gt = torch.from_numpy(batch['gt_j3d']); gt = (gt - gt[:, :1]).cuda().contiguous()
m = torch.zeros(B, 154, device='cuda'); v = torch.zeros_like(m); st = torch.zeros(1, dtype=torch.int32, device='cuda')
for rep in range(3):
    eng.refine_run(x, b, gt, m, v, st, 1e-2, 5)
    torch.cuda.synchronize()
    out = (ctypes.c_longlong * 256)()
    lib.jrr_debug_read_stamps.restype = ctypes.c_int
    assert lib.jrr_debug_read_stamps(out) == 0
    s = np.array(out[:]).reshape(4, 64)
    for w in range(4):
        n = int(s[w, 63]); t = s[w]
        tiles = np.diff(t[1:n])
        print(f'rep {rep} wg{w}: prologue {t[1]-t[0]}  tiles n={n-1} mean {tiles.mean():.0f} min {tiles.min()} max {tiles.max()} first3 {tiles[:3]} last3 {tiles[-3:]}  pre-flush {t[61]-t[n-1]} flush {t[62]-t[61]} total {t[62]-t[0]}')
