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


def _capsule_surface(rng):
    """6890 vertices on capsules around the 23 bones of the rest skeleton (265 rings of 26 vertices, shared out by bone length,
    at least 3 per bone), one open triangle strip grid per bone (faces <= 13776), returned in a seeded RANDOM vertex order --
    what an arbitrary mesh file looks like to the 32-vertex tiles: no locality at all in the file order."""
    S = 26
    n_rings_total = NUM_VERTS // S
    assert n_rings_total * S == NUM_VERTS
    bones = [(int(SMPL_PARENTS[j]), j) for j in range(1, NUM_JOINTS)]
    length = np.array([np.linalg.norm(_REST_JOINTS[c] - _REST_JOINTS[p]) for p, c in bones])
    rings = np.maximum(3, np.floor(length / length.sum() * n_rings_total).astype(int))
    while rings.sum() > n_rings_total:
        rings[np.argmax(rings)] -= 1
    while rings.sum() < n_rings_total:
        rings[np.argmax(length / rings)] += 1
    verts, faces, base = [], [], 0
    for (p, c), nr in zip(bones, rings):
        a, b = _REST_JOINTS[p], _REST_JOINTS[c]
        ax = (b - a) / (np.linalg.norm(b - a) + 1e-12)
        ref = np.array([0.0, 0.0, 1.0]) if abs(ax[2]) < 0.9 else np.array([1.0, 0.0, 0.0])
        u = np.cross(ax, ref); u /= np.linalg.norm(u)
        w = np.cross(ax, u)
        rad = 0.035 + 0.06 * np.exp(-np.linalg.norm(0.5 * (a + b) - np.array([0.0, 0.0, 0.0])) / 0.35)   # thicker near the trunk
        for k in range(nr):
            t = (k + 0.5) / nr
            taper = np.sqrt(max(1e-3, 1.0 - (2 * t - 1) ** 4))                    # rounded ends
            for q in range(S):
                ph = 2 * np.pi * q / S
                verts.append(a + t * (b - a) + rad * taper * (np.cos(ph) * u + np.sin(ph) * w))
        for k in range(nr - 1):
            for q in range(S):
                v00, v01 = base + k * S + q, base + k * S + (q + 1) % S
                v10, v11 = v00 + S, v01 + S
                faces.append((v00, v10, v11)); faces.append((v00, v11, v01))
        base += nr * S
    verts = np.stack(verts)
    faces = np.asarray(faces, dtype=np.int64)
    perm = rng.permutation(NUM_VERTS)              # file position k holds grid vertex perm[k]
    inv = np.empty_like(perm); inv[perm] = np.arange(NUM_VERTS)
    return verts[perm], inv[faces].astype(np.int32)


def synthetic_smpl(seed: int = 1234, max_influences: int = 4, kind: str = 'surface') -> Dict[str, np.ndarray]:
    """Seeded synthetic body model with SMPL's shapes and structural properties.
    kind='surface' (default; the benchmarked body): a closed body-scale triangle mesh (6890 vertices / 13776 faces, vertex order
    spatially coherent: ring-major), shapedirs ~ N(0, 0.01^2), posedirs ~ N(0, 0.002^2), sparse non-negative J_regressor rows
    summing to 1, <= `max_influences` smooth skinning weights per vertex summing to 1, canonical SMPL parents.
    kind='capsules': vertices on capsules around the 23 bones in a seeded RANDOM file order, the 4 nearest joints skin each
    vertex -- the stand-in for a real model file whose vertex order means nothing to the 32-vertex tiles of the LBS kernels
    (the library's internal joint-sorted order then decides the tile classes)."""
    if kind == 'capsules':
        return _synthetic_from_surface(seed, max_influences, *_capsule_surface(np.random.RandomState(seed + 99)), smooth=False,
                                       provenance=f'synthetic-capsules(seed={seed})')
    if kind != 'surface':
        raise ValueError(f'unknown synthetic body kind {kind!r}')
    v_template, faces = _body_surface()
    assert v_template.shape == (NUM_VERTS, 3) and faces.shape == (13776, 3)
    return _synthetic_from_surface(seed, max_influences, v_template, faces, smooth=True, provenance=None)


def _synthetic_from_surface(seed, max_influences, v_template, faces, smooth, provenance):
    rng = np.random.RandomState(seed)
    V = NUM_VERTS
    v_template = v_template + rng.normal(0.0, 0.002, size=(V, 3))       # break the exact symmetry
    if smooth:
        # blend shapes: spatially SMOOTH displacement fields (low-order harmonics over the surface
        # parametrisation), like real SMPL's -- independent per-vertex noise would crumple the mesh
        shapedirs = _smooth_fields(rng, 3 * NUM_BETAS).T.reshape(V, 3, NUM_BETAS) * 0.01
        posedirs = _smooth_fields(rng, 207 * 3).reshape(207, 3, V).transpose(0, 2, 1).reshape(207, V * 3) * 0.002
    else:
        # no surface parametrisation in a random file order: smooth functions of the rest POSITION instead
        def fields(n):
            out = np.zeros((n, V))
            for _ in range(3):
                k = rng.normal(0.0, 4.0, size=(n, 3))
                out += rng.normal(0.0, 1.0, size=(n, 1)) * np.cos(k @ v_template.T + rng.uniform(0, 2 * np.pi, size=(n, 1)))
            return out / np.sqrt((out ** 2).mean(axis=1, keepdims=True) + 1e-12)
        shapedirs = fields(3 * NUM_BETAS).T.reshape(V, 3, NUM_BETAS) * 0.01
        posedirs = fields(207 * 3).reshape(207, 3, V).transpose(0, 2, 1).reshape(207, V * 3) * 0.002
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
    out = dict(v_template=v_template.astype(np.float32), shapedirs=shapedirs.astype(np.float32),
               posedirs=posedirs.astype(np.float32), J_regressor=J_regressor.astype(np.float32),
               lbs_weights=W.astype(np.float32), parents=SMPL_PARENTS.copy(), faces=faces)
    if provenance:
        out['provenance'] = provenance
    return out


def with_wide_tile(model: Dict[str, np.ndarray], tile: int = 100, joints: int = 13, seed: int = 3) -> Dict[str, np.ndarray]:
    """The same body with ONE 32-vertex tile of the file order skinned by `joints` joints in total (<= 4 influences per vertex
    kept, rows still sum to 1): the test / bench case of the per-tile skinning classes -- one wide tile must cost itself a
    second pass, not move the whole model to slower kernels."""
    rng = np.random.RandomState(seed)
    W = model['lbs_weights'].copy()
    rows = np.arange(tile * 32, min(tile * 32 + 32, NUM_VERTS))
    used = set(np.nonzero((W[rows] != 0).any(0))[0].tolist())
    spare = [j for j in range(NUM_JOINTS) if j not in used]
    rng.shuffle(spare)
    need = joints - len(used)
    assert 0 <= need <= len(spare), (len(used), joints)
    k = 0
    for v in rows:
        if need <= 0:
            break
        take = spare[k:k + min(3, need)]
        k += len(take); need -= len(take)
        keep = int(np.argmax(W[v]))
        w = np.zeros(NUM_JOINTS, dtype=W.dtype)
        w[keep] = 0.55
        w[take] = 0.45 / len(take)
        W[v] = w
    out = dict(model)
    out['lbs_weights'] = W
    out['provenance'] = f"{model.get('provenance', 'synthetic')}+wide_tile({tile},{joints})"
    return out


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


# ---- the licensed model file --------------------------------------------------------------------------------------
# SMPL_NEUTRAL.pkl as distributed (and as SPIN's data/smpl holds it; smplx reads it at scripts/smpl.py:7-9 through
# pickle.load(..., encoding='latin1')) is a Python-2 pickle whose arrays are `chumpy.ch.Ch` objects (v_template, shapedirs,
# posedirs, weights, J), whose J_regressor is a scipy.sparse matrix and whose kintree_table is uint32.  Unpickling it the plain
# way imports chumpy -- which is absent here and on the GPU boxes (and does not import under numpy >= 1.24 anyway).  The
# loader below never imports it: a restricted Unpickler maps every `chumpy.*` class to a stand-in that just keeps the pickled
# state, and the numeric content is taken from that state (`x` of a leaf Ch; the few re-ordering nodes SMPL files have been
# seen with are resolved recursively).  Everything that is not numpy / scipy.sparse / a plain container is refused.
class _ChumpyStandIn:
    """Holds the pickled state of a chumpy object; np.asarray(obj) gives its value (no chumpy import)."""
    _jrr_chumpy_class = 'chumpy.ch.Ch'

    def __new__(cls, *args, **kwargs):
        return object.__new__(cls)

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        self.__dict__['_state'] = state if isinstance(state, dict) else {'x': state}

    def _value(self):
        st = self.__dict__.get('_state', {})
        name = self._jrr_chumpy_class.rsplit('.', 1)[-1].lower()
        if 'x' in st and not isinstance(st['x'], _ChumpyStandIn):
            return np.asarray(st['x'])
        inner = st.get('a', st.get('x'))
        if inner is None:
            raise ValueError(f'cannot take the value of a pickled {self._jrr_chumpy_class} (state keys: {sorted(st)})')
        a = np.asarray(inner)
        if name == 'transpose':
            return np.transpose(a, st.get('axes'))
        if name == 'reshape':
            return np.reshape(a, st.get('newshape'))
        if name == 'select':
            return a.ravel()[np.asarray(st['idxs'])]
        if name == 'ch':
            return a
        raise ValueError(f'unsupported chumpy node {self._jrr_chumpy_class} in the model file (state keys: {sorted(st)})')

    def __array__(self, dtype=None, copy=None):
        v = self._value()
        return v.astype(dtype) if dtype is not None else v


_SAFE_GLOBALS = {
    ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'), ('numpy', 'ndarray'), ('numpy', 'dtype'),
    ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'), ('numpy.core.numeric', '_frombuffer'),
    ('numpy._core.numeric', '_frombuffer'), ('copy_reg', '_reconstructor'), ('copyreg', '_reconstructor'),
    ('__builtin__', 'object'), ('builtins', 'object'), ('__builtin__', 'set'), ('builtins', 'set'), ('__builtin__', 'frozenset'),
    ('builtins', 'frozenset'), ('__builtin__', 'slice'), ('builtins', 'slice'), ('__builtin__', 'complex'), ('builtins', 'complex'),
    ('collections', 'OrderedDict'), ('_codecs', 'encode'),
}


class _SMPLUnpickler(pickle.Unpickler):
    _standins: Dict[str, type] = {}

    def find_class(self, module, name):
        if module == 'chumpy' or module.startswith('chumpy.'):
            full = f'{module}.{name}'
            if full not in self._standins:
                self._standins[full] = type('Chumpy_' + name, (_ChumpyStandIn,), {'_jrr_chumpy_class': full})
            return self._standins[full]
        if module == 'scipy.sparse' or module.startswith('scipy.sparse.'):
            import scipy.sparse as sp
            if hasattr(sp, name) and name.endswith(('_matrix', '_array')):
                return getattr(sp, name)        # (py2-era files name the defining sub-module, e.g. scipy.sparse.csc)
        if (module, name) in _SAFE_GLOBALS:
            if module in ('copy_reg', '__builtin__'):
                module = {'copy_reg': 'copyreg', '__builtin__': 'builtins'}[module]
            if module.startswith('numpy.core') and not _has_module(module):
                module = module.replace('numpy.core', 'numpy._core')
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f'SMPL model file refers to {module}.{name}: not a numpy / scipy.sparse / chumpy object')


def _has_module(name: str) -> bool:
    import importlib.util
    try:
        return importlib.util.find_spec(name) is not None
    except (ImportError, ValueError):
        return False


def load_smpl_pickle(path: str) -> Dict:
    """SMPL_NEUTRAL.pkl -> dict of plain arrays, without importing chumpy (see above)."""
    with open(path, 'rb') as f:
        d = _SMPLUnpickler(f, encoding='latin1').load()
    if not isinstance(d, dict):
        raise ValueError(f'{path}: expected a pickled dict of SMPL arrays, got {type(d).__name__}')
    return {(k.decode('latin1') if isinstance(k, bytes) else k): v for k, v in d.items()}


def _model_from_file_dict(d: Dict, path: str) -> Dict[str, np.ndarray]:
    """The arrays smplx.SMPL takes from the model file (smplx 0.1.26 body_models.py: v_template, shapedirs[:, :, :10],
    posedirs reshaped to (207, 20670), J_regressor, weights, kintree_table[0], f) in this package's layout."""
    V = NUM_VERTS

    def arr(key, dtype=np.float32):
        if key not in d:
            raise KeyError(f'{path}: no {key!r} in the model file (keys: {sorted(map(str, d))})')
        x = d[key]
        x = x.toarray() if hasattr(x, 'toarray') else np.asarray(x)
        return np.ascontiguousarray(x.astype(dtype))

    v_template = arr('v_template')
    shapedirs = arr('shapedirs')
    if shapedirs.ndim != 3 or shapedirs.shape[:2] != (V, 3) or shapedirs.shape[2] < NUM_BETAS:
        raise ValueError(f'{path}: shapedirs has shape {shapedirs.shape}, expected (6890, 3, >= {NUM_BETAS})')
    shapedirs = np.ascontiguousarray(shapedirs[:, :, :NUM_BETAS])          # the 300-component files: first 10, as smplx does
    posedirs = arr('posedirs')
    if posedirs.ndim == 3:                                                  # file layout (6890, 3, 207)
        posedirs = np.ascontiguousarray(posedirs.reshape(V * 3, -1).T)      # smplx: (207, 20670)
    Jr = arr('J_regressor')
    W = arr('weights')
    if 'kintree_table' in d:
        parents = np.asarray(d['kintree_table'])[0].astype(np.int64).astype(np.int32)    # uint32 with 2^32 - 1 at the root
    else:
        parents = SMPL_PARENTS.copy()
    parents[0] = -1
    faces = arr('f', np.int64).astype(np.int32)
    for name, a, shape in (('v_template', v_template, (V, 3)), ('posedirs', posedirs, (207, V * 3)), ('J_regressor', Jr, (NUM_JOINTS, V)),
                           ('weights', W, (V, NUM_JOINTS)), ('kintree_table[0]', parents, (NUM_JOINTS,))):
        if a.shape != shape:
            raise ValueError(f'{path}: {name} has shape {a.shape}, expected {shape}')
    if faces.ndim != 2 or faces.shape[1] != 3:
        raise ValueError(f'{path}: f has shape {faces.shape}, expected (F, 3)')
    return dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, J_regressor=Jr, lbs_weights=W, parents=parents,
                faces=faces, provenance=f'file:{path}')


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
                # (smplx reads .npz model files with allow_pickle=True too: a scipy-sparse J_regressor arrives as a 0-d object array)
                d = {k: (v.item() if isinstance(v, np.ndarray) and v.dtype == object and v.ndim == 0 else v)
                     for k, v in np.load(path, allow_pickle=True).items()}
            else:
                d = load_smpl_pickle(path)
            return _model_from_file_dict(d, path)
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
    with _few_blas_threads():
        return _synthetic_batch(model, J_h36m, B, seed)


class _few_blas_threads:
    """numpy's OpenBLAS starts one worker per logical CPU (256 on the GPU box) and the workers keep SPINNING for tens of
    milliseconds after a product.  A 100-iteration jrr_refine_run that starts right after a batch was generated then shares the
    host with 256 busy threads exactly when it has filled the device's queue and waits for room in it: the device was seen idle
    for ~60 ms in the middle of 40 % of such calls at 256 poses (25 ms calls reading 90 ms; DESIGN.md section 0,
    tools/exp/small_batch_triggers.py).  Data generation therefore runs its BLAS calls on at most 8 threads."""

    def __enter__(self):
        self._ctx = None
        try:
            import threadpoolctl
            self._ctx = threadpoolctl.threadpool_limits(limits=8, user_api='blas')
            self._ctx.__enter__()
        except Exception:
            self._ctx = None
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
        return False


def _synthetic_batch(model, J_h36m, B, seed):
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
