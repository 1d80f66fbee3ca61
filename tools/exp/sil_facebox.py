"""Distribution of pixel-centre candidates per face bounding box (what a lane of the face sweep iterates over)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
PKG = 'joint-regressor-refinement_amd'
eng_mod = importlib.import_module(PKG + '.engine'); sm = importlib.import_module(PKG + '.smpl_model')
B = 64
model = sm.synthetic_smpl(1234); J = sm.default_h36m_regressor()
batch = sm.synthetic_batch(model, J, B, seed=1000)
dm = eng_mod.DeviceModel(model, 'cuda:0')
eng = eng_mod.RefineEngine(dm, B, flags=eng_mod.FLAG_KEEP_VERTS)
eng.set_j_regressor(torch.from_numpy(J))
T = lambda k: torch.from_numpy(batch[k]).cuda().contiguous()
_, verts = eng.find_joints_forward(T('betas'), x6d=T('pose6d'), return_verts=True)
v = verts.cpu().numpy().astype(np.float64); cam = batch['cam'].astype(np.float64)
X = -2 * v[..., 0] + cam[:, None, 0]; Y = -2 * v[..., 1] + cam[:, None, 1]; Z = 2 * v[..., 2] + cam[:, None, 2]
f = 5000.0 / 224.0
x = f * X / Z; y = f * Y / Z
F = model['faces']
S = 224
def rng(lo, hi):
    a = np.ceil((S * (1 - hi) - 1) * 0.5 - 1e-3); b = np.floor((S * (1 - lo) - 1) * 0.5 + 1e-3)
    return np.clip(a, 0, S - 1), np.clip(b, 0, S - 1)
fx = x[:, F]; fy = y[:, F]
xlo, xhi = rng(fx.min(-1), fx.max(-1)); ylo, yhi = rng(fy.min(-1), fy.max(-1))
nx = np.maximum(xhi - xlo + 1, 0); ny = np.maximum(yhi - ylo + 1, 0)
n = (nx * ny).astype(int)
print('faces', F.shape[0], 'mean candidates %.2f' % n.mean(), 'hist:', np.bincount(np.minimum(n, 12).ravel()) / n.size)
# wave-level cost: lanes = 64 consecutive faces; trips = max over the wave
nw = n[:, :13760].reshape(B, -1, 64)
print('sum over waves of max-trip per pose: %.0f   sum of mean-trip: %.0f   (ratio %.2f)' % (nw.max(-1).sum(-1).mean(), nw.mean(-1).sum(-1).mean(), nw.max(-1).sum(-1).mean() / nw.mean(-1).sum(-1).mean()))
for cap in (2, 4, 6):
    capped = np.minimum(nw, cap).max(-1).sum(-1).mean(); rest = np.maximum(n - cap, 0).sum(-1).mean()
    print('cap', cap, ': capped wave trips %.0f + %.0f leftover pixel tests (= %.0f wave-iterations of 64)' % (capped, rest, rest / 64))
