"""CPU-only checks: flag surface, checkpoint format, C-ABI export table, data-parallel host logic."""
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
from conftest import PKG_NAME, ROOT

T = torch.from_numpy


def _mod(name):
    return importlib.import_module(f'{PKG_NAME}.{name}')


def test_reference_flags_kept():
    """the 15 flags of the reference's scripts/args.py:5-21: same names, types, defaults"""
    a = _mod('args')
    ns = a.get_args([])
    for k, v in a.REFERENCE_FLAGS.items():
        assert hasattr(ns, k), k
        assert getattr(ns, k) == v, (k, getattr(ns, k), v)
    ns2 = a.get_args(['--batch_size', '32', '--j_reg_lr', '0.5', '--wandb_log', '--device', 'cuda:1', '--unknown_flag', '3'])
    assert ns2.batch_size == 32 and ns2.j_reg_lr == 0.5 and ns2.wandb_log is True and ns2.device == 'cuda:1'


def test_checkpoint_format_roundtrip(tmp_path, j_h36m_np):
    ck = _mod('checkpoint')
    J = T(j_h36m_np)
    p = str(tmp_path / 'retrained_J_Regressor.pt')
    ck.save_j_regressor(J, p)
    raw = torch.load(p, weights_only=True)               # what the reference's readers do (test.py:46-47)
    assert torch.is_tensor(raw) and raw.shape == (17, 6890) and raw.dtype == torch.float32
    assert raw.stride() == (1, 17) and raw.requires_grad      # layout of the shipped artefact
    assert torch.equal(raw.detach(), J)
    assert torch.equal(ck.load_j_regressor(p), J)
    # arbitrary strides / requires_grad written by someone else
    weird = torch.zeros(6890, 17)
    weird.t()[:] = J
    torch.save(weird.t().requires_grad_(True), p)
    assert torch.equal(ck.load_j_regressor(p), J)
    torch.save(torch.zeros(3, 3), p)
    with pytest.raises(ValueError):
        ck.load_j_regressor(p)
    # 107 non-zeros, 62 positive: the shipped checkpoint's support
    assert int((J != 0).sum()) == 107 and int((J > 0).sum()) == 62


def test_c_abi_exports_every_declared_symbol():
    """include/jrr.h <-> libjrr_hip.so <-> the ctypes table agree (no compute: no GPU needed)."""
    hdr = open(os.path.join(ROOT, 'include', 'jrr.h')).read()
    declared = set(re.findall(r'\b(jrr_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'jrr_status'}
    lib_mod = _mod('_lib')
    _mod('build').build(verbose=False)
    lib = lib_mod.load()
    assert declared == set(lib_mod.SIGNATURES), declared ^ set(lib_mod.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.jrr_version() >= 100
    assert lib.jrr_engine_workspace_bytes(4096, 5) > 2 * 3 * 6912 * 4096 * 4
    # error behaviour: negative status + message, no exception across the boundary
    import ctypes
    rc = lib.jrr_engine_create(None, 4, 4, None, 0, 0, ctypes.byref(ctypes.c_void_p()))
    assert rc == -1 and b'bad argument' in lib.jrr_last_error()
    # a caller-owned model buffer that is too small is refused before anything touches the device (JRR_ERR_WORKSPACE)
    assert lib.jrr_model_bytes() > 3 * 3 * 6912 * 224 * 4
    z = np.zeros(8, dtype=np.float32)
    rc = lib.jrr_model_create_in(z.ctypes.data, z.ctypes.data, z.ctypes.data, z.ctypes.data, z.ctypes.data, z.ctypes.data,
                                 ctypes.c_void_p(256), 1024, ctypes.byref(ctypes.c_void_p()))
    assert rc == -3 and b'jrr_model_bytes' in lib.jrr_last_error()


def test_product_path_has_no_oracle_import():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/"""
    pkg_dir = os.path.join(ROOT, PKG_NAME)
    for dp, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(import|from)\s+oracle', src, re.M), os.path.join(dp, f)


def test_shard_bounds_cover_batch():
    d = _mod('dist')
    for n, w in [(32768, 8), (4096, 2), (10, 3), (5, 8)]:
        edges = [d.shard_bounds(n, r, w) for r in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in edges) - min(h - l for l, h in edges) <= 1


_DP_WORKER = r'''
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import oracle
PKG = "joint-regressor-refinement_amd"
d = importlib.import_module(PKG + ".dist")
sm = importlib.import_module(PKG + ".smpl_model")
import torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
T = torch.from_numpy
model = sm.synthetic_smpl(1234)
J0 = T(sm.default_h36m_regressor())
B = 12
full = sm.synthetic_batch(model, J0.numpy(), B, seed=77)
lo, hi = d.shard_bounds(B, rank, world)
smpl = oracle.OracleSMPL(model)
x6, betas = T(full["pose6d"]), T(full["betas"])
gt_c = oracle.move_pelvis(T(full["gt_j3d"]))
# rank-local inner loop with the GLOBAL batch as the MSE normaliser, then the shared J step
o, p, b, _ = oracle.refine_poses(smpl, J0, x6[lo:hi, :1], x6[lo:hi, 1:], betas[lo:hi], gt_c[lo:hi], 2, batch_norm=B)
_, gJ, _ = oracle.j_regressor_loss_and_grad(smpl, J0, o, p, b, gt_c[lo:hi], batch_norm=B)
J = J0.clone(); m = torch.zeros_like(J); v = torch.zeros_like(J)
# the replicated shared-parameter step: every rank's gradient is normalised by the GLOBAL batch, so the all-reduced sum is
# the single-process gradient; then the identical Adam update on every rank (what optimize.py does with its flat bucket)
d.all_reduce_sum_(gJ)
oracle.adam_step(J, gJ, m, v, 1, 1e-2)
a = torch.full((3,), float(rank + 1))
d.all_reduce_sum_(a)
assert a.tolist() == [sum(range(1, world + 1))] * 3
if rank == 0:
    np.savez(sys.argv[2], J=J.numpy(), o=o.numpy(), lo=lo, hi=hi)
dist.barrier()
dist.destroy_process_group()
'''


def test_data_parallel_j_step_equals_single_process(tmp_path, smpl_model_np, j_h36m_np):
    """world_size-2 gloo run of the sharded inner loop + all-reduced J step == the single-process run."""
    script = tmp_path / 'dp_worker.py'
    script.write_text(_DP_WORKER)
    out = str(tmp_path / 'dp.npz')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', OMP_NUM_THREADS='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', str(script), ROOT, out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = np.load(out)
    sm = _mod('smpl_model')
    B = 12
    full = sm.synthetic_batch(smpl_model_np, j_h36m_np, B, seed=77)
    smpl = oracle.OracleSMPL(smpl_model_np)
    x6, betas = T(full['pose6d']), T(full['betas'])
    gt_c = oracle.move_pelvis(T(full['gt_j3d']))
    J0 = T(j_h36m_np)
    o, p, b, _ = oracle.refine_poses(smpl, J0, x6[:, :1], x6[:, 1:], betas, gt_c, 2)
    _, gJ, _ = oracle.j_regressor_loss_and_grad(smpl, J0, o, p, b, gt_c)
    J = J0.clone()
    oracle.adam_step(J, gJ, torch.zeros_like(J), torch.zeros_like(J), 1, 1e-2)
    np.testing.assert_allclose(got['o'], o[int(got['lo']):int(got['hi'])].numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(got['J'], J.numpy(), rtol=0, atol=1e-6)


def test_dataset_tensor_layout(tmp_path):
    """row f4: the reference's precomputed-tensor directory layout and the crop re-positioning of gt_j2d
    (scripts/data.py:49-86,134-158,220-247) on synthetic files."""
    d = _mod('data')
    loc = tmp_path / 'precomputed_val'
    loc.mkdir()
    g = torch.Generator().manual_seed(0)
    n = 5
    bb = torch.tensor([[100., 200., 700., 600.], [0., 0., 1000., 1000.], [300., 100., 500., 900.],
                       [50., 60., 950., 940.], [400., 400., 600., 600.]])
    files = dict(bboxes=bb, betas=torch.randn(n, 10, generator=g), estimated_translation=torch.randn(n, 3, generator=g),
                 gt_j2d=torch.rand(n, 17, 2, generator=g) * 1000, gt_j3d=torch.randn(n, 17, 3, generator=g) * 300,
                 intrinsics=torch.eye(3).repeat(n, 1, 1), orient=torch.randn(n, 1, 6, generator=g),
                 pose=torch.randn(n, 23, 6, generator=g))
    for k, v in files.items():
        torch.save(v, str(loc / f'{k}.pt'))
    ds = d.data_set('validation', root=str(tmp_path))
    assert len(ds) == n
    s = ds[0]
    assert set(s) == {'bboxes', 'betas', 'cam', 'gt_j2d', 'gt_j3d', 'intrinsics', 'orient', 'pose', 'inc_gt'}
    # sample 0: bbox y 100..700, x 200..600 -> square crop of side 600 centred at (400, 400): min (100, 100); scale is
    # the half side in units of 500 px (0.6), so (j - min)/scale spans 0..1000 and /(1000/224) spans 0..224
    min_x, min_y, scale = d.crop_params(bb)
    assert abs(float(min_x[0]) - 100) < 1e-3 and abs(float(min_y[0]) - 100) < 1e-3 and abs(float(scale[0]) - 0.6) < 1e-6
    np.testing.assert_allclose(s['gt_j2d'].numpy(), ((files['gt_j2d'][0] - 100) / 0.6 / (1000 / 224)).numpy(), rtol=1e-5, atol=1e-3)
    # full-frame bbox: identity crop up to the 1000 -> 224 resolution change
    np.testing.assert_allclose(ds[1]['gt_j2d'].numpy(), (files['gt_j2d'][1] / (1000 / 224)).numpy(), rtol=1e-5, atol=1e-3)
    with pytest.raises(FileNotFoundError):
        d.data_set('train', root=str(tmp_path))


def test_pixel_centre_f64_product_equals_fp32_division():
    """csrc/sil.hip pix_x<S>(): fl32((2i+1) * (1/S) in f64) must equal the IEEE fp32 quotient (2i+1)/S for every pixel index of every
    image size the rasteriser is instantiated for (the multiples of 32 up to 256), so that its pixel centres are those of the fp32
    formula (pytorch3d 0.3.0 arithmetic)."""
    for S in range(32, 257, 32):
        i = np.arange(S)
        ref = np.float32(1) - (np.float32(2) * i.astype(np.float32) + np.float32(1)) / np.float32(S)
        got = np.float32(1) - ((2 * i + 1).astype(np.float64) * (1.0 / float(S))).astype(np.float32)
        assert ref.dtype == np.float32 and got.dtype == np.float32
        assert np.array_equal(ref, got), S


def test_shared_bucket_layout():
    """the flat all-reduce bucket of the outer step: [dJ | dD | dShapeD | sums | history], every section
    256-byte aligned (the C ABI takes 16-byte aligned pointers), one contiguous tail to read back"""
    opt = _mod('optimize')
    eng = _mod('engine')
    b = opt.SharedBucket('cpu', True, True, 10)
    assert b.dJ.shape == (17, 6890) and b.dD.numel() == eng.DISC_PARAMS and b.dS.numel() == eng.SHAPE_DISC_PARAMS
    base = b.flat.data_ptr()
    for t in (b.dJ, b.dD, b.dS):
        assert (t.data_ptr() - base) % 256 == 0 and t.is_contiguous()
    assert b.tail.numel() == opt.N_SCALARS + 50 and b.hist.shape == (10, 5)
    assert 7.8e6 < b.nbytes < 7.9e6                              # section 8e: 7.83 MB
    b.put(3, torch.arange(5.0))
    b.hist[2, 1] = 7.0
    tail = b.tail.numpy()
    assert tail[3] == 10.0 and tail[opt.N_SCALARS + 2 * 5 + 1] == 7.0
    b.dD[5] = 2.0                                               # views alias the flat buffer
    assert b.flat[(b.dD.data_ptr() - base) // 4 + 5] == 2.0
    nb = opt.SharedBucket('cpu', False, False, 0)
    assert nb.dD.numel() == 0 and nb.dS.numel() == 0 and nb.tail.numel() == opt.N_SCALARS


def test_bench_gpus_n_starts_n_ranks_before_touching_the_gpu():
    """`python bench.py --gpus 2` outside torchrun becomes a torchrun CHILD of two rank processes (the driver's command line for
    N > 1); here, without a GPU, each rank stops at the availability check -- which it reaches only as a rank of a world of 2,
    i.e. after the launch -- and the parent relays the failure (exit status != 0, no JSON line)."""
    if torch.cuda.device_count() > 0:       # (counting devices does not initialise the GPU)
        pytest.skip('a GPU is present: the N-rank launch itself is exercised by tests/test_gpu_dp.py')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '2', '--warmup', '1'],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    # a rank that got as far as the check says so as a rank of a world of 2 (the launcher ends the other rank as soon as the first
    # one fails: one such line is guaranteed, two are usual)
    import re
    assert re.search(r'bench\.py needs an MI355X .*\[rank [01] of 2\]', r.stderr), r.stderr[-1500:]
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]


def test_bench_provenance_and_experiment_library_guard(monkeypatch):
    """bench.py names the binary it measured (path + sha256[:16] of the loaded library, every JRR_* knob of the environment) and refuses a
    library named by JRR_LIB unless --allow_experiment_lib says the line is an experiment -- before anything touches the GPU"""
    import hashlib
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.delenv('JRR_LIB', raising=False)
    monkeypatch.setenv('JRR_DISC_NARROW', '0')
    prov = bench.provenance(False)
    lib = os.path.join(ROOT, PKG_NAME, 'libjrr_hip.so')
    assert prov['in_tree_lib'] is True and prov['lib_path'] == os.path.relpath(lib, ROOT)
    assert prov['lib_sha16'] == hashlib.sha256(open(lib, 'rb').read()).hexdigest()[:16]
    assert prov['jrr_env'] == {'JRR_DISC_NARROW': '0'}
    monkeypatch.setenv('JRR_LIB', lib)
    with pytest.raises(SystemExit, match='allow_experiment_lib'):
        bench.provenance(False)
    assert bench.provenance(True)['jrr_env']['JRR_LIB'] == lib
    # the separately reported blocks are functions of their own, and the line is assembled in one place
    for name in ('block_folded', 'block_bf16x3', 'block_support_tiles', 'block_config5', 'block_config2', 'block_skin_variants', 'block_driver',
                 'build_line', 'cpu_baseline', 'provenance', 'self_launch'):
        assert callable(getattr(bench, name)), name


@pytest.mark.parametrize('protocol', [0, 2])
def test_smpl_model_file_with_chumpy_objects_loads_without_chumpy(tmp_path, protocol):
    """SMPL('SPIN/data/smpl', batch_size=1) reads SMPL_NEUTRAL.pkl (/root/reference/scripts/optimize.py:96-99,
    scripts/smpl.py:7-9): a Python-2 pickle of chumpy.ch.Ch arrays, a scipy.sparse J_regressor, (6890,3,207) posedirs, 300
    shape components, a uint32 kintree_table.  chumpy is absent here and on the GPU boxes: the loader maps chumpy classes to a
    stand-in that keeps the pickled state, refuses anything that is not numpy / scipy.sparse / a plain container."""
    import pickle
    from conftest import write_chumpy_style_pickle
    sm = _mod('smpl_model')
    body = sm.synthetic_smpl(1234, kind='capsules')
    write_chumpy_style_pickle(body, str(tmp_path / 'SMPL_NEUTRAL.pkl'), protocol=protocol)
    assert not any(k == 'chumpy' or k.startswith('chumpy.') for k in sys.modules)
    got = sm.load_smpl_model(str(tmp_path), allow_synthetic=False)
    assert not any(k == 'chumpy' or k.startswith('chumpy.') for k in sys.modules)          # still not imported
    assert got['provenance'] == f"file:{tmp_path / 'SMPL_NEUTRAL.pkl'}"
    for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights', 'parents', 'faces'):
        assert got[k].dtype == body[k].dtype and np.array_equal(got[k], body[k]), k
    assert got['parents'][0] == -1 and got['shapedirs'].shape == (6890, 3, 10)
    # the same arrays as a plain .npz (with a sparse regressor stored as an object, as np.savez does)
    import scipy.sparse as sp
    (tmp_path / 'SMPL_NEUTRAL.pkl').unlink()
    np.savez(tmp_path / 'SMPL_NEUTRAL.npz', v_template=body['v_template'], shapedirs=body['shapedirs'],
             posedirs=body['posedirs'].T.reshape(6890, 3, 207), J_regressor=np.array(sp.csr_matrix(body['J_regressor']), dtype=object),
             weights=body['lbs_weights'], kintree_table=np.stack([body['parents'].astype(np.int64), np.arange(24)]), f=body['faces'])
    got2 = sm.load_smpl_model(str(tmp_path), allow_synthetic=False)
    for k in ('v_template', 'posedirs', 'J_regressor', 'lbs_weights', 'parents', 'faces'):
        assert np.array_equal(got2[k], body[k]), k
    # a pickle that names anything else is refused before it can run
    class Evil:
        def __reduce__(self):
            return (os.system, ('true',))
    (tmp_path / 'SMPL_NEUTRAL.npz').unlink()
    with open(tmp_path / 'SMPL_NEUTRAL.pkl', 'wb') as f:
        pickle.dump({'v_template': Evil()}, f, protocol=2)
    with pytest.raises(pickle.UnpicklingError):
        sm.load_smpl_model(str(tmp_path), allow_synthetic=False)
    # a wrong shape is an error with the file's name in it, not a crash further down
    bad = dict(body); bad['v_template'] = body['v_template'][:100]
    write_chumpy_style_pickle(bad, str(tmp_path / 'SMPL_NEUTRAL.pkl'))
    with pytest.raises(ValueError, match='v_template'):
        sm.load_smpl_model(str(tmp_path), allow_synthetic=False)


def test_synthetic_bodies_and_tile_classes():
    """the two synthetic bodies and the wide-tile variant: SMPL's shapes, <= 4 influences, rows summing to 1; the capsule body's
    file order has no locality (every tile sees nearly every joint) while its joint-sorted order fits 16-joint windows"""
    sm = _mod('smpl_model')
    for kind in ('surface', 'capsules'):
        m = sm.synthetic_smpl(1234, kind=kind)
        W = m['lbs_weights']
        assert m['v_template'].shape == (6890, 3) and m['posedirs'].shape == (207, 20670) and m['faces'].max() == 6889
        assert (W != 0).sum(1).max() <= 4 and np.abs(W.sum(1) - 1).max() < 1e-6 and m['faces'].shape[0] <= 14336
        per_tile = [int((W[32 * t:32 * t + 32] != 0).any(0).sum()) for t in range(216)]
        assert (max(per_tile) <= 8) == (kind == 'surface')
    w = sm.with_wide_tile(sm.synthetic_smpl(1234), 100, 13)
    W = w['lbs_weights']
    per_tile = [int((W[32 * t:32 * t + 32] != 0).any(0).sum()) for t in range(216)]
    assert per_tile[100] == 13 and sum(n > 8 for n in per_tile) == 1 and (W != 0).sum(1).max() <= 4


def test_integration_table_names_every_entry_point():
    """INTEGRATION.md's call-site table has a row for every exported symbol of include/jrr.h (rows abbreviate families as
    `jrr_model_create / _create_in / _destroy`)"""
    hdr = open(os.path.join(ROOT, 'include', 'jrr.h')).read()
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    declared = set(re.findall(r'\b(jrr_[a-z0-9_]+)\s*\(', hdr)) - {'jrr_status'}
    named = set(re.findall(r'\bjrr_[a-z0-9_]+', doc))
    for cell in re.findall(r'`([^`]*)`', doc):
        parts = [p.strip() for p in cell.split('/')]
        if len(parts) > 1 and parts[0].startswith('jrr_'):
            stem = parts[0].rsplit('_', 1)[0]
            named |= {stem + p for p in parts[1:] if p.startswith('_')}
    assert not (declared - named), sorted(declared - named)


def test_no_kernel_spills_registers():
    """the compiler's resource report of the CURRENT sources (hipcc cross-compiles gfx950 here): no kernel of the library keeps a
    register in scratch memory -- the matrix kernels run at 200+ registers, where one more live value in a hot loop turns into
    scratch traffic nobody timed (round 4: six 12-slot WIDE instantiations of k_lbs_fwd, k_evaluate); and every matrix kernel keeps
    the occupancy its design states (two waves per SIMD)"""
    rows = _mod('build').resource_rows()
    assert len(rows) > 100                      # every instantiation reports
    bad = [(r['name'], r['vspill'], r['scratch']) for r in rows if r.get('vspill', 0) or r.get('scratch', 0)]
    assert not bad, bad
    for r in rows:
        if any(k in r['name'] for k in ('k_lbs_fwd', 'k_lbs_bwd', 'k_blend_adjoint')):
            if 'k_lbs_bwd16' in r['name'] and r['name'].rstrip().endswith(', 2>'):
                assert r['occ'] == 1                        # the JRR_BWD16_NG=2 experiment: ONE 512-register wave per SIMD, by design
                continue
            assert r['occ'] >= 2 and r['vgpr'] <= 256, r


def test_every_environment_knob_is_documented():
    """every JRR_* variable the library or the host side reads appears in DESIGN.md (section 3's list of verification knobs)"""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, 'joint-regressor-refinement_amd')
    knobs = set()
    for f in glob.glob(os.path.join(pkg, 'csrc', '*.h*')):
        knobs |= set(re.findall(r'getenv\("(JRR_[A-Z0-9_]+)"\)', open(f).read()))
    for f in glob.glob(os.path.join(pkg, '*.py')) + [os.path.join(root, 'bench.py')]:
        knobs |= set(re.findall(r"environ\.get\(['\"](JRR_[A-Z0-9_]+)", open(f).read()))
    assert len(knobs) > 10
    design = open(os.path.join(root, 'DESIGN.md')).read()
    missing = sorted(k for k in knobs if k not in design)
    assert not missing, missing
