"""SMPL model constants: optional real-model loader and the seeded synthetic generator.

The reference builds its body model with `SMPL("SPIN/data/smpl", batch_size=1)`
(/root/reference/scripts/optimize.py:96-99); the model file is licence-gated and absent, so
benchmarks and tests use a synthetic model of identical shapes.

Arrays (float32 unless noted), named as in smplx:
  v_template (6890,3)  shapedirs (6890,3,10)  posedirs (207, 20670)
  J_regressor (24,6890)  lbs_weights (6890,24)  parents (24,) int32  faces (F,3) int32
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, Optional

import numpy as np

NUM_VERTS = 6890
NUM_JOINTS = 24
NUM_BETAS = 10
NUM_H36M = 17
SMPL_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
                        dtype=np.int32)

# Rough rest-pose joint centres (metres) of a T-posed body, used only to place synthetic vertices.
_REST_JOINTS = np.array([
    [0.00, -0.22, 0.03], [0.07, -0.31, 0.01], [-0.07, -0.31, 0.01], [0.00, -0.10, 0.00],
    [0.10, -0.69, 0.01], [-0.10, -0.69, 0.01], [0.00, 0.03, 0.00], [0.09, -1.09, -0.03],
    [-0.09, -1.09, -0.03], [0.00, 0.09, 0.02], [0.11, -1.15, 0.09], [-0.11, -1.15, 0.09],
    [0.00, 0.30, -0.01], [0.08, 0.21, -0.01], [-0.08, 0.21, -0.01], [0.00, 0.38, 0.03],
    [0.17, 0.23, -0.02], [-0.17, 0.23, -0.02], [0.43, 0.22, -0.03], [-0.43, 0.22, -0.03],
    [0.68, 0.23, -0.03], [-0.68, 0.23, -0.03], [0.76, 0.22, -0.03], [-0.76, 0.22, -0.03]], dtype=np.float64)


N_RINGS, N_SEGS = 84, 82          # 84*82 + 2 poles = 6890 vertices; 2*84*82 = 13776 faces (SMPL's counts)


def _body_surface():
    """A closed, star-shaped (about the vertical axis) body-scale surface sampled on a ring/segment grid in
    ring-major order, with T-pose "arm" fins at shoulder height: gives the synthetic model a genuine
    genus-0 triangle mesh with SMPL's vertex and face counts (V - E + F = 2  =>  F = 2V - 4 = 13776)."""
    y_top, y_bot = 0.52, -1.18
    yc, H = 0.5 * (y_top + y_bot), 0.5 * (y_top - y_bot)
    verts = [np.array([0.0, y_top, 0.0])]
    for u in range(N_RINGS):
        th = np.pi * (u + 1) / (N_RINGS + 1)
        y = yc + H * np.cos(th)
        girth = np.sin(th) ** 0.6 * (0.75 + 0.25 * np.cos(2.2 * th))      # shoulders wider than ankles
        for w in range(N_SEGS):
            ph = 2 * np.pi * w / N_SEGS
            arm = 2.6 * np.exp(-((y - 0.22) / 0.055) ** 2) * np.abs(np.cos(ph)) ** 10
            verts.append(np.array([0.21 * girth * (1 + arm) * np.cos(ph), y, 0.12 * girth * np.sin(ph)]))
    verts.append(np.array([0.0, y_bot, 0.0]))
    verts = np.stack(verts)
    faces = []
    ring = lambda u, w: 1 + u * N_SEGS + (w % N_SEGS)
    for w in range(N_SEGS):
        faces.append((0, ring(0, w), ring(0, w + 1)))
    for u in range(N_RINGS - 1):
        for w in range(N_SEGS):
            faces.append((ring(u, w), ring(u + 1, w), ring(u + 1, w + 1)))
            faces.append((ring(u, w), ring(u + 1, w + 1), ring(u, w + 1)))
    last = 1 + N_RINGS * N_SEGS
    for w in range(N_SEGS):
        faces.append((last, ring(N_RINGS - 1, w + 1), ring(N_RINGS - 1, w)))
    return verts, np.asarray(faces, dtype=np.int32)


def _smooth_fields(rng, n: int) -> np.ndarray:
    """n smooth scalar fields over the surface, (n, 6890), unit RMS: sums of 3 low-order harmonics in the ring
    angle theta and the segment angle phi with random amplitudes and phases."""
    th = np.concatenate([[0.0], np.repeat(np.pi * (np.arange(N_RINGS) + 1) / (N_RINGS + 1), N_SEGS), [np.pi]])
    ph = np.concatenate([[0.0], np.tile(2 * np.pi * np.arange(N_SEGS) / N_SEGS, N_RINGS), [0.0]])
    out = np.zeros((n, NUM_VERTS))
    for _ in range(3):
        p = rng.randint(0, 4, size=(n, 1))
        q = rng.randint(0, 4, size=(n, 1))
        a = rng.normal(0.0, 1.0, size=(n, 1))
        out += a * np.cos(p * th[None] * 2 + rng.uniform(0, 2 * np.pi, size=(n, 1))) * \
            np.cos(q * ph[None] + rng.uniform(0, 2 * np.pi, size=(n, 1))) * np.sin(th[None]) ** (q > 0)
    return out / np.sqrt((out ** 2).mean(axis=1, keepdims=True) + 1e-12)


def synthetic_smpl(seed: int = 1234, max_influences: int = 4) -> Dict[str, np.ndarray]:
    """Seeded synthetic body model with SMPL's shapes and structural properties: a closed body-scale
    triangle mesh (6890 vertices / 13776 faces, vertex order spatially coherent), shapedirs ~ N(0, 0.01^2),
    posedirs ~ N(0, 0.002^2), sparse non-negative J_regressor rows summing to 1, <= `max_influences`
    smooth skinning weights per vertex summing to 1, canonical SMPL parents."""
    rng = np.random.RandomState(seed)
    V = NUM_VERTS
    v_template, faces = _body_surface()
    assert v_template.shape == (V, 3) and faces.shape == (13776, 3)
    v_template = v_template + rng.normal(0.0, 0.002, size=(V, 3))       # break the exact symmetry
    # blend shapes: spatially SMOOTH displacement fields (low-order harmonics over the surface
    # parametrisation), like real SMPL's -- independent per-vertex noise would crumple the mesh
    shapedirs = _smooth_fields(rng, 3 * NUM_BETAS).T.reshape(V, 3, NUM_BETAS) * 0.01
    posedirs = _smooth_fields(rng, 207 * 3).reshape(207, 3, V).transpose(0, 2, 1).reshape(207, V * 3) * 0.002
    # skinning weights: the nearest rest joints, Gaussian fall-off
    d2 = ((v_template[:, None, :] - _REST_JOINTS[None]) ** 2).sum(-1)           # (V, 24)
    W = np.zeros((V, NUM_JOINTS))
    near = np.argsort(d2, axis=1)[:, :max_influences]
    for v in range(V):
        w = np.exp(-d2[v, near[v]] / (2 * 0.12 ** 2)) + 1e-3
        W[v, near[v]] = w / w.sum()
    # rest-joint regressor: each joint regresses from its 12 nearest vertices
    J_regressor = np.zeros((NUM_JOINTS, V))
    for j in range(NUM_JOINTS):
        pick = np.argsort(d2[:, j])[:12]
        w = rng.uniform(0.2, 1.0, size=len(pick))
        J_regressor[j, pick] = w / w.sum()
    return dict(v_template=v_template.astype(np.float32), shapedirs=shapedirs.astype(np.float32),
                posedirs=posedirs.astype(np.float32), J_regressor=J_regressor.astype(np.float32),
                lbs_weights=W.astype(np.float32), parents=SMPL_PARENTS.copy(), faces=faces)


def shuffled_vertex_order(model: Dict[str, np.ndarray], seed: int = 5, scope: str = 'parts'):
    """The same body with the vertex order of an arbitrary mesh FILE: vertices stay grouped by body part (their dominant
    skinning joint, parts in order of first appearance) but are shuffled (seeded) inside each part -- what a real model
    file looks like to the 32-vertex tiles, as opposed to the synthetic generator's ring-major order.  Every
    per-vertex array and the face indices are permuted consistently; returns the new model and `perm` with
    new_vertex[k] = old_vertex[perm[k]].  scope='all': one shuffle of ALL vertices (no locality left at all: only the
    library's internal joint-sorted order makes such a file fit the joint-sparse kernels)."""
    rng = np.random.RandomState(seed)
    W = model['lbs_weights']
    dom = W.argmax(1)
    parts = []
    for j in dom:
        if j not in parts:
            parts.append(int(j))
    if scope == 'interleaved':
        # joint-sorted order cut into 32-vertex tiles, tiles dealt round-robin from three distant thirds of the body: every tile
        # is coherent (few joints) but CONSECUTIVE tiles share no joints, so the backward kernel's 16-joint segments (jrr_common.h)
        # last one or two tiles and every joint's slab rows are revisited many times -- the stress case of its flush path
        keys = []
        for v in range(NUM_VERTS):
            nz = np.nonzero(W[v])[0]
            o = nz[np.argsort(-W[v][nz], kind='stable')]
            keys.append(tuple(list(o) + [NUM_JOINTS] * (4 - len(o)))[:4] + (v,))
        srt = np.array(sorted(range(NUM_VERTS), key=lambda v: keys[v]), dtype=np.int64)
        ntile = (NUM_VERTS + 31) // 32
        third = (ntile + 2) // 3
        tiles = [t for k in range(third) for t in (k, k + third, k + 2 * third) if t < ntile]
        perm = np.concatenate([srt[32 * t:32 * t + 32] for t in tiles])
    elif scope == 'all':
        perm = rng.permutation(NUM_VERTS).astype(np.int64)
    else:
        perm = np.concatenate([rng.permutation(np.nonzero(dom == j)[0]) for j in parts]).astype(np.int64)
    assert sorted(perm.tolist()) == list(range(NUM_VERTS))
    inv = np.empty_like(perm)
    inv[perm] = np.arange(NUM_VERTS)
    out = dict(model)
    out['v_template'] = model['v_template'][perm]
    out['shapedirs'] = model['shapedirs'][perm]
    out['posedirs'] = np.ascontiguousarray(model['posedirs'].reshape(207, NUM_VERTS, 3)[:, perm].reshape(207, NUM_VERTS * 3))
    out['J_regressor'] = np.ascontiguousarray(model['J_regressor'][:, perm])
    out['lbs_weights'] = model['lbs_weights'][perm]
    if model.get('faces') is not None:
        out['faces'] = inv[model['faces']].astype(np.int32)
    out['provenance'] = f"{model.get('provenance', 'synthetic')}+shuffled(seed={seed})"
    return out, perm


def load_smpl_model(model_dir: Optional[str], allow_synthetic: bool = True) -> Dict[str, np.ndarray]:
    """Load a real SMPL model if the user supplies one (`<dir>/SMPL_NEUTRAL.pkl` or `.npz`, the
    layout smplx expects at scripts/optimize.py:96-99).  If no model file is found: the seeded synthetic
    model WITH a prominent warning when `allow_synthetic`, otherwise FileNotFoundError (a mistyped
    --smpl_dir must not silently train a regressor on a fake body).  The returned dict carries
    `provenance` ('file:<path>' or 'synthetic(seed=1234)')."""
    if model_dir:
        for name in ('SMPL_NEUTRAL.npz', 'SMPL_NEUTRAL.pkl', 'smpl_neutral.npz'):
            path = os.path.join(model_dir, name)
            if not os.path.exists(path):
                continue
            if path.endswith('.npz'):
                d = dict(np.load(path, allow_pickle=True))
            else:
                with open(path, 'rb') as f:
                    d = pickle.load(f, encoding='latin1')
            V = NUM_VERTS
            shapedirs = np.asarray(d['shapedirs'], dtype=np.float32)[:, :, :NUM_BETAS]
            posedirs = np.asarray(d['posedirs'], dtype=np.float32)            # (6890,3,207)
            if posedirs.ndim == 3:
                posedirs = posedirs.reshape(V * 3, -1).T                       # smplx: (207, 20670)
            Jr = d['J_regressor']
            Jr = np.asarray(Jr.todense() if hasattr(Jr, 'todense') else Jr, dtype=np.float32)
            parents = np.asarray(d['kintree_table'])[0].astype(np.int32) if 'kintree_table' in d \
                else SMPL_PARENTS.copy()
            parents[0] = -1
            return dict(v_template=np.asarray(d['v_template'], dtype=np.float32), shapedirs=shapedirs,
                        posedirs=np.ascontiguousarray(posedirs), J_regressor=Jr,
                        lbs_weights=np.asarray(d['weights'], dtype=np.float32), parents=parents,
                        faces=np.asarray(d['f'], dtype=np.int32), provenance=f'file:{path}')
        if not allow_synthetic:
            raise FileNotFoundError(f'no SMPL_NEUTRAL.{{npz,pkl}} under {model_dir!r}; pass --synthetic to run on the '
                                    f'seeded synthetic body model instead')
        import warnings
        warnings.warn(f'SMPL model not found under {model_dir!r}: using the SYNTHETIC body model (seed 1234). '
                      f'A J_regressor trained on it is not a SMPL regressor.', RuntimeWarning, stacklevel=2)
    m = synthetic_smpl()
    m['provenance'] = 'synthetic(seed=1234)'
    return m


# The 107 non-zero entries of the shipped /root/reference/models/retrained_J_Regressor.pt are the
# default H36M regressor initialisation when no SPIN/data/J_regressor_h36m.npy is supplied
# (scripts/optimize.py:105-107).  They are stored as (row, col, value) triplets in
# tests/golden/j_regressor_triplets.npz (data, not code) and mirrored in assets/.
def j_regressor_from_triplets(rows, cols, vals) -> np.ndarray:
    J = np.zeros((NUM_H36M, NUM_VERTS), dtype=np.float32)
    J[np.asarray(rows), np.asarray(cols)] = np.asarray(vals, dtype=np.float32)
    return J


def default_h36m_regressor(path: Optional[str] = None, allow_default: bool = True) -> np.ndarray:
    """H36M regressor initialisation (scripts/optimize.py:105-107): `SPIN/data/J_regressor_h36m.npy`
    if the user supplies it, else (with a warning, or FileNotFoundError when `allow_default` is False)
    the 107 non-zeros of the shipped checkpoint (assets/)."""
    if path and os.path.exists(path):
        return np.load(path).astype(np.float32)
    if path:
        if not allow_default:
            raise FileNotFoundError(f'{path!r} not found; pass --synthetic to start from the shipped checkpoint\'s support')
        import warnings
        warnings.warn(f'{path!r} not found: initialising the H36M regressor from the 107 non-zeros of the shipped checkpoint',
                      RuntimeWarning, stacklevel=2)
    t = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets', 'j_regressor_h36m_init.npz'))
    return j_regressor_from_triplets(t['rows'], t['cols'], t['vals'])


def synthetic_h36m_regressor(model: Dict[str, np.ndarray], seed: int = 7, support: int = 6,
                             with_negatives: bool = True) -> np.ndarray:
    """Seeded sparse (17,6890) regressor with the shipped checkpoint's structure: a few positive
    entries per row (not normalised) plus a few negative entries that ReLU removes."""
    rng = np.random.RandomState(seed)
    J = np.zeros((NUM_H36M, NUM_VERTS), dtype=np.float32)
    for i in range(NUM_H36M):
        idx = rng.choice(NUM_VERTS, size=support, replace=False)
        J[i, idx[:support - 2]] = rng.uniform(0.1, 0.6, size=support - 2)
        if with_negatives:
            J[i, idx[support - 2:]] = -rng.uniform(0.01, 0.2, size=2)
        else:
            J[i, idx[support - 2:]] = rng.uniform(0.1, 0.6, size=2)
    return J


def synthetic_batch(model: Dict[str, np.ndarray], J_h36m: np.ndarray, B: int, seed: int = 0):
    """Seeded "SPIN-init" batch: per-joint axis-angle ~ N(0,0.3^2) -> R ->
    6-D (first two columns interleaved as x[2i+k] = R[i,k]) + N(0,0.01^2) off-manifold noise;
    betas ~ N(0,1) clipped to +-3; cam = (0,0,2*5000/(224*0.9)); gt_j3d (mm) = 1000 x the
    pelvis-centred H36M joints of a perturbed pose (axis-angle noise 0.1) + N(0,10^2 mm).

    Ground truth is produced with a float64 numpy LBS of the same synthetic model (data
    generation, not the product path)."""
    rng = np.random.RandomState(seed)
    aa = rng.normal(0.0, 0.3, size=(B, NUM_JOINTS, 3))
    R = _rodrigues_np(aa.reshape(-1, 3)).reshape(B, NUM_JOINTS, 3, 3)
    x6 = R[:, :, :, :2].reshape(B, NUM_JOINTS, 6) + rng.normal(0.0, 0.01, size=(B, NUM_JOINTS, 6))
    betas = np.clip(rng.normal(0.0, 1.0, size=(B, NUM_BETAS)), -3, 3)
    cam = np.tile(np.array([[0.0, 0.0, 2 * 5000 / (224 * 0.9)]]), (B, 1))
    aa_gt = aa + rng.normal(0.0, 0.1, size=aa.shape)
    R_gt = _rodrigues_np(aa_gt.reshape(-1, 3)).reshape(B, NUM_JOINTS, 3, 3)
    Jn = np.maximum(J_h36m.astype(np.float64), 0)
    Jn = Jn / Jn.sum(1, keepdims=True)
    gt = np.empty((B, NUM_H36M, 3))
    cols = np.nonzero(Jn.any(axis=0))[0]          # only the regressor's support vertices need skinning
    if len(cols) > NUM_VERTS // 2:
        cols = None
    Jc = Jn if cols is None else Jn[:, cols]
    for s in range(0, B, 256):
        verts = _lbs_np(model, R_gt[s:s + 256], betas[s:s + 256], cols)
        j = np.einsum('iv,bvc->bic', Jc, verts)
        gt[s:s + 256] = (j - j[:, :1]) * 1000.0
    gt = gt + rng.normal(0.0, 10.0, size=gt.shape)
    gt = gt - gt[:, :1]
    return dict(pose6d=x6.astype(np.float32), betas=betas.astype(np.float32), cam=cam.astype(np.float32),
                gt_j3d=gt.astype(np.float32))


def _rodrigues_np(aa: np.ndarray) -> np.ndarray:
    angle = np.linalg.norm(aa + 1e-8, axis=1, keepdims=True)
    axis = aa / angle
    c, s = np.cos(angle)[:, :, None], np.sin(angle)[:, :, None]
    K = np.zeros((aa.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -axis[:, 2], axis[:, 1]
    K[:, 1, 0], K[:, 1, 2] = axis[:, 2], -axis[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -axis[:, 1], axis[:, 0]
    return np.eye(3)[None] + s * K + (1 - c) * (K @ K)


def _lbs_np(model, R, betas, cols=None):
    """float64 numpy LBS used only to synthesise ground-truth joints for benchmark batches; `cols` restricts the
    skinned vertices (the rest joints always use every vertex)."""
    B = R.shape[0]
    vt = model['v_template'].astype(np.float64)
    sdirs = model['shapedirs'].astype(np.float64)
    v_shaped = vt[None] + (betas @ sdirs.reshape(-1, sdirs.shape[-1]).T).reshape(B, -1, 3)
    J = np.matmul(model['J_regressor'].astype(np.float64)[None], v_shaped)
    pf = (R[:, 1:] - np.eye(3)).reshape(B, -1)
    pd = model['posedirs'].astype(np.float64)
    W = model['lbs_weights'].astype(np.float64)
    if cols is not None:
        v_shaped = v_shaped[:, cols]
        pd = pd.reshape(pd.shape[0], -1, 3)[:, cols].reshape(pd.shape[0], -1)
        W = W[cols]
    v_posed = v_shaped + (pf @ pd).reshape(B, -1, 3)
    parents = model['parents']
    G_R = np.empty((B, NUM_JOINTS, 3, 3))
    G_t = np.empty((B, NUM_JOINTS, 3))
    G_R[:, 0], G_t[:, 0] = R[:, 0], J[:, 0]
    for i in range(1, NUM_JOINTS):
        p = parents[i]
        G_R[:, i] = G_R[:, p] @ R[:, i]
        G_t[:, i] = np.einsum('brc,bc->br', G_R[:, p], J[:, i] - J[:, p]) + G_t[:, p]
    A_t = G_t - np.einsum('bjrc,bjc->bjr', G_R, J)
    T_R = np.einsum('vj,bjrc->bvrc', W, G_R)
    T_t = np.einsum('vj,bjr->bvr', W, A_t)
    return np.einsum('bvrc,bvc->bvr', T_R, v_posed) + T_t
