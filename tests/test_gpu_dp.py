"""The product's data-parallel path, executed: 2 rank processes through `torch.distributed.run` running the real
driver (`optimize_pose_refiner()`: jdist.init, sharded engines with batch_norm = global batch, all-reduces of dJ /
discriminator gradients / loss sums) and the real `bench.py --gpus 2` on the HIP engine, against the 1-rank run on the
same global batch.  On the 1-GPU test box both ranks share cuda:0 and talk over gloo (`--single_device`); the
collectives' call sites are the ones RCCL serves with one rank per GPU.

The rank processes are launched by tests/conftest.py when collection finishes (before this pytest process
initialises the GPU); the tests below read their outputs.  Reference semantics under sharding:
/root/reference/scripts/optimize.py:220-265 (inner loop), :276-293 (discriminator updates), :300-312 (J step).
"""
import json
import os

import numpy as np
import pytest

import conftest

pytestmark = [pytest.mark.gpu, pytest.mark.dp_gpu]


def _run(name):
    r = conftest.DP_RUNS.get(name)
    assert r is not None, 'multi-rank runs were not launched (no GPU visible at collection time?)'
    assert r['rc'] == 0, f"{name} failed (rc {r['rc']}):\n{r['out'][-3000:]}\n{r['err']}"
    return r


def _load(prefix, world):
    d = conftest.DP_RUNS['dir']
    return [dict(np.load(os.path.join(d, f'{prefix}.rank{r}.npz'))) for r in range(world)]


def test_two_rank_driver_equals_single_process():
    _run('w1'); _run('w2')
    (one,) = _load('w1', 1)
    two = _load('w2', 2)
    assert (int(two[0]['lo']), int(two[0]['hi']), int(two[1]['lo']), int(two[1]['hi'])) == (0, 128, 128, 256)
    # per-pose state: the shards concatenate to the single-process result (different launch geometry at 128 vs 256
    # poses: fp32 summation order differs in the last bit, Adam's normalised step amplifies it to a fraction of lr)
    x2 = np.concatenate([two[0]['x6d'], two[1]['x6d']])
    b2 = np.concatenate([two[0]['betas'], two[1]['betas']])
    assert x2.shape == one['x6d'].shape == (256, 24, 6)
    assert np.abs(x2 - one['x6d']).max() < 2e-4
    assert np.abs(b2 - one['betas']).max() < 2e-4
    # shared parameters: identical on both ranks (replicated Adam after the all-reduce) and equal to the 1-rank run
    for k in ('J', 'disc', 'sdisc'):
        assert np.array_equal(two[0][k], two[1][k]), k
        assert np.abs(two[0][k] - one[k]).max() < 5e-5, k
    # the log record (all-reduced sums / global batch) agrees; both J steps ran (one inside the loop, one after)
    h1, h2 = json.loads(str(one['history']))[0], json.loads(str(two[0]['history']))[0]
    for k in ('joint_loss', 'pose_discriminated_loss', 'shape_discriminated_loss', 'pose_discriminator_loss',
              'shape_discriminator_loss', 'j_regressor_error', 'mpjpe', 'pampjpe'):
        assert h1[k] is not None and h2[k] is not None, k
        np.testing.assert_allclose(h2[k], h1[k], rtol=2e-3, err_msg=k)


def test_eight_rank_driver_equals_single_process():
    """world size 8 (the node of BASELINE configs[3]) on the same global batch of 256: eight contiguous shards of 32 poses"""
    _run('w1'); _run('w8')
    (one,) = _load('w1', 1)
    eight = _load('w8', 8)
    assert [(int(r['lo']), int(r['hi'])) for r in eight] == [(32 * k, 32 * k + 32) for k in range(8)]
    x8 = np.concatenate([r['x6d'] for r in eight])
    b8 = np.concatenate([r['betas'] for r in eight])
    assert np.abs(x8 - one['x6d']).max() < 2e-4 and np.abs(b8 - one['betas']).max() < 2e-4
    for k in ('J', 'disc', 'sdisc'):
        for r in eight[1:]:
            assert np.array_equal(eight[0][k], r[k]), k              # replicated Adam after the ONE all-reduce: bit-identical
        assert np.abs(eight[0][k] - one[k]).max() < 5e-5, k
    h1, h8 = json.loads(str(one['history']))[0], json.loads(str(eight[0]['history']))[0]
    for k in ('joint_loss', 'pose_discriminated_loss', 'shape_discriminated_loss', 'pose_discriminator_loss',
              'shape_discriminator_loss', 'j_regressor_error', 'mpjpe', 'pampjpe'):
        np.testing.assert_allclose(h8[k], h1[k], rtol=2e-3, err_msg=k)
    np.testing.assert_allclose(np.array(h8['loss_history']), np.array(h1['loss_history']), rtol=2e-3, atol=1e-6)


def test_two_rank_driver_all_vertex_tiles_equals_single_process():
    """--all_vertex_tiles (what the bench headline runs: every iteration skins all 216 vertex tiles) through the driver on two
    ranks against one rank, and against the default run on the regressor's support tiles (same results up to summation order)"""
    _run('w1a'); _run('w2a'); _run('w1')
    (one,) = _load('w1a', 1)
    two = _load('w2a', 2)
    (dflt,) = _load('w1', 1)
    h1 = json.loads(str(one['history']))[0]
    nd = json.loads(str(dflt['history']))[0]['vertex_tiles_run']
    assert h1['vertex_tiles_run'] == 216 and (0 < nd < 60 if conftest.support_tiles_available() else nd == 216)
    x2 = np.concatenate([two[0]['x6d'], two[1]['x6d']])
    b2 = np.concatenate([two[0]['betas'], two[1]['betas']])
    assert np.abs(x2 - one['x6d']).max() < 2e-4 and np.abs(b2 - one['betas']).max() < 2e-4
    for k in ('J', 'disc', 'sdisc'):
        assert np.array_equal(two[0][k], two[1][k]), k
        assert np.abs(two[0][k] - one[k]).max() < 5e-5, k
    h2 = json.loads(str(two[0]['history']))[0]
    for k in ('joint_loss', 'pose_discriminated_loss', 'shape_discriminated_loss', 'pose_discriminator_loss',
              'shape_discriminator_loss', 'j_regressor_error', 'mpjpe', 'pampjpe'):
        np.testing.assert_allclose(h2[k], h1[k], rtol=2e-3, err_msg=k)
    # all tiles vs the support's tiles: the sums over the tiles associate differently (1e-4 max / 2e-8 mean measured)
    dx = np.abs(one['x6d'] - dflt['x6d'])
    assert dx.max() < 6e-4 and dx.mean() < 1e-6, (dx.max(), dx.mean())
    assert np.abs(one['J'] - dflt['J']).max() < 5e-6


def test_two_rank_driver_with_silhouette_reprojection_shape_disc_equals_single_process():
    """BASELINE configs[4] through the driver under sharding: --silhouette --reprojection --shape_disc, 64 poses, a J step inside the
    loop; the synthetic masks / 2-D targets are drawn for the global batch and sliced, so the two shards see the 1-rank targets"""
    _run('w1s'); _run('w2s')
    (one,) = _load('w1s', 1)
    two = _load('w2s', 2)
    assert (int(two[0]['lo']), int(two[0]['hi']), int(two[1]['lo']), int(two[1]['hi'])) == (0, 32, 32, 64)
    h1, h2 = json.loads(str(one['history']))[0], json.loads(str(two[0]['history']))[0]
    assert h1['vertex_tiles_run'] == h2['vertex_tiles_run'] == 216
    # per pose the rasteriser and its fixed-point adjoint do not depend on the batch: only the launch geometry of the LBS
    # kernels differs between 32 and 64 poses (last-bit summation order, Adam-amplified)
    for k in ('x6d', 'betas', 'cam'):
        d = np.abs(np.concatenate([two[0][k], two[1][k]]) - one[k])
        assert d.max() < 2e-3 and d.mean() < 2e-5, (k, d.max(), d.mean())
    for k in ('J', 'disc', 'sdisc'):
        assert np.array_equal(two[0][k], two[1][k]), k
        assert np.abs(two[0][k] - one[k]).max() < 5e-5, k
    for k in ('joint_loss', 'pose_discriminated_loss', 'shape_discriminated_loss', 'pose_discriminator_loss',
              'shape_discriminator_loss', 'j_regressor_error', 'mpjpe', 'pampjpe'):
        np.testing.assert_allclose(h2[k], h1[k], rtol=5e-3, err_msg=k)
    np.testing.assert_allclose(np.array(h2['loss_history'])[0], np.array(h1['loss_history'])[0], rtol=2e-4)


def test_one_rank_over_rccl_runs_the_collective_call_sites():
    """backend nccl (= RCCL) with a one-rank group on this box's GPU: the driver takes its N > 1 branch (refine_run ->
    jrr_j_regressor_grad_support -> all-reduce -> jrr_j_step_apply_support per J step; ONE flat all-reduce per outer step) and
    RCCL executes every collective on the device buffers.  A sum over one rank is the identity: results equal the plain run's
    (the in-call J steps of the plain run and the host-driven ones are the same kernels on the same numbers)."""
    _run('w1'); r = _run('w1n_1rank')
    (one,) = _load('w1', 1)
    (n,) = _load('w1n', 1)
    for k in ('J', 'disc', 'sdisc', 'x6d', 'betas'):
        if k == 'sdisc':      # float atomics in the shape discriminator's gradient: order differs run to run
            assert np.abs(n[k] - one[k]).max() < 5e-5, k
        else:
            assert np.array_equal(n[k], one[k]), k


def test_support_sized_all_reduce_equals_the_dense_one():
    """the in-loop J step exchanges the regressor's support (8 704 B, default) or the dense (17,6890) gradient
    (--j_allreduce dense): per rank the exchanged values are the same numbers, so with two ranks (a + b: one order) J and
    everything downstream is BIT-identical; with eight ranks only the all-reduce's own summation order may differ"""
    _run('w2'); _run('w2d'); _run('w8'); _run('w8d')
    s2, d2 = _load('w2', 2), _load('w2d', 2)
    for k in ('J', 'disc', 'x6d', 'betas'):      # (not 'sdisc': the shape discriminator's 171 weight gradients are float atomics)
        assert np.array_equal(s2[0][k], d2[0][k]) and np.array_equal(s2[1][k], d2[1][k]), k
    assert np.abs(s2[0]['sdisc'] - d2[0]['sdisc']).max() < 1e-6
    s8, d8 = _load('w8', 8), _load('w8d', 8)
    assert np.abs(s8[0]['J'] - d8[0]['J']).max() < 2e-6
    for r in d8[1:]:
        assert np.array_equal(d8[0]['J'], r['J'])
    moved = s8[0]['J'] != d8[0]['J']
    assert not moved[_default_J() <= 0].any()                      # outside the support both are exactly the untouched values


def _default_J():
    import importlib
    return importlib.import_module(conftest.PKG_NAME + '.smpl_model').default_h36m_regressor()


def test_eight_ranks_of_4096_poses_equal_one_rank_of_32768():
    """BASELINE configs[3] at its own shard size (8 x 4096 poses; one GPU stands in for the eight, gloo for RCCL): the real
    driver with a J step + all-reduce after EVERY inner iteration (/root/reference/scripts/optimize.py:220-265,300-312 under
    sharding).  The shards concatenate to the 1-rank run on the same 32 768 poses; J and the discriminators are bit-identical
    across the ranks."""
    _run('w1big'); _run('w8big')
    (one,) = _load('w1big', 1)
    eight = _load('w8big', 8)
    assert [(int(r['lo']), int(r['hi'])) for r in eight] == [(4096 * k, 4096 * k + 4096) for k in range(8)]
    x8 = np.concatenate([r['x6d'] for r in eight])
    b8 = np.concatenate([r['betas'] for r in eight])
    assert x8.shape == one['x6d'].shape == (32768, 24, 6)
    # two launch geometries (4096 vs 32 768 poses per engine): fp32 summation orders differ in the last bit and Adam's first steps
    # lr * g / (|g| + eps) amplify that wherever a gradient entry is ~ 0 -- the bound of every geometry comparison of the suite
    # (DESIGN.md section 6): max 6e-4 over the 4.7 M entries, and the MEAN pins the trajectory
    dx, db = np.abs(x8 - one['x6d']), np.abs(b8 - one['betas'])
    assert dx.max() < 6e-4 and dx.mean() < 1e-6 and db.max() < 6e-4 and db.mean() < 1e-6, (dx.max(), dx.mean(), db.max(), db.mean())
    for k in ('J', 'disc'):
        for r in eight[1:]:
            assert np.array_equal(eight[0][k], r[k]), k
        assert np.abs(eight[0][k] - one[k]).max() < 5e-5, k
    h1, h8 = json.loads(str(one['history']))[0], json.loads(str(eight[0]['history']))[0]
    for k in ('joint_loss', 'pose_discriminated_loss', 'pose_discriminator_loss', 'j_regressor_error', 'mpjpe', 'pampjpe'):
        np.testing.assert_allclose(h8[k], h1[k], rtol=2e-3, err_msg=k)
    assert (eight[0]['J'] != _default_J()).sum() == 62              # three J steps, exactly the positive support moved


def test_two_rank_driver_moved_the_regressor_only_on_its_support():
    _run('w2')
    two = _load('w2', 2)
    import importlib
    sm = importlib.import_module(conftest.PKG_NAME + '.smpl_model')
    J0 = sm.default_h36m_regressor()
    moved = two[0]['J'] != J0
    assert moved.sum() == (J0 > 0).sum() == 62          # golden G8: exactly the positive support receives gradient
    assert not moved[J0 <= 0].any()


@pytest.mark.parametrize('name,world,batch', [('bench2', 2, 256), ('bench2t', 2, 256), ('bench8', 8, 128),
                                              ('bench2s', 2, 256), ('bench8s', 8, 64)])
def test_bench_n_ranks(name, world, batch):
    """`python bench.py --gpus N` (bench.py launches its own torchrun child before touching the GPU) and the explicit
    torchrun line both give ONE JSON line of an N-rank run, with the evidence of the rank count in it"""
    r = _run(name)
    lines = [l for l in r['out'].splitlines() if l.startswith('{')]
    assert len(lines) == 1, r['out'][-2000:]
    j = json.loads(lines[0])
    strong = name.endswith('s')      # --scaling strong: `batch` here is the shard, the global batch was the command line's --batch
    assert j['n_gpus'] == world and j['scaling'] == ('strong' if strong else 'weak') and j['value'] > 0
    assert j['config']['global_batch'] == world * batch and j['config']['parallelism'] == f'dp{world}'
    assert j['config']['poses_per_gpu'] == batch
    # weak: N x (iterations of a per-GPU batch) per second; strong: iterations of the one global batch per second
    it_s = 1e3 / j['ms_per_step']
    assert abs(j['value'] - it_s * (1 if strong else world)) < 1e-2 * j['value']
    assert j['unit'] == f"it/s (x{world * batch if strong else batch} poses)"
    assert j['provenance']['in_tree_lib'] is True and len(j['provenance']['lib_sha16']) == 16
    assert j['config']['j_steps_in_timed_regions'] >= 1
    assert j['j_step']['allreduce_bytes'] == 17 * 128 * 4          # the regressor's support, not the dense (17,6890) gradient
    assert np.isfinite(j['config']['joint_loss_last'])
    c = j['collective']
    assert c['world'] == world and len(c['ranks_seen']) == world
    assert sorted(rk[0] for rk in c['ranks_seen']) == list(range(world))
    assert len({rk[3] for rk in c['ranks_seen']}) == world             # that many processes
    assert c['allreduce_check'] == c['allreduce_expected'] == world * (world - 1) / 2     # sum of ranks over the collective
    assert c['backend'] == 'gloo' and c['single_device_debug'] is True
    assert j['cadence1']['timed_regions'] >= 5 and j['cadence1']['value'] > 0
    # the first N > 1 run explains itself: per-rank spread, the collective on its own, the collective's share of a cadence-1 step
    pr = j['per_rank_ms_per_step']
    assert len(pr['each']) == world and 0 < pr['min'] <= pr['median'] <= pr['max'] <= j['ms_per_step'] * 1.05
    cc = j['collective_cost']
    sa = cc['standalone_allreduce']
    assert sa['j_step_support_17x128']['bytes'] == 8704 and sa['j_step_dense_17x6890']['bytes'] == 468520
    assert 7.8e6 < sa['outer_step_flat_bucket']['bytes'] < 7.9e6
    for rec in sa.values():
        assert rec['median_us_synchronised'] > 0 and rec['us_back_to_back'] > 0
    assert cc['cadence1_ms_per_step'] > 0 and cc['cadence1_noop_collective_ms_per_step'] > 0
    assert abs(cc['collective_us_per_step'] - (cc['cadence1_ms_per_step'] - cc['cadence1_noop_collective_ms_per_step']) * 1e3) < 0.2


def test_bench_rejects_a_world_that_contradicts_gpus():
    r = conftest.DP_RUNS.get('bench_mismatch')
    assert r is not None and r['rc'] != 0
    assert '--gpus 1 but the launcher started 2' in (r['out'] + r['err'])


def _multi_gpu():
    import torch
    return torch.cuda.device_count() >= 2          # (counting devices does not initialise the GPU)


@pytest.mark.skipif(not _multi_gpu(), reason='needs two GPUs: one rank per GPU over RCCL (the 1-GPU boxes run the gloo stand-in above)')
def test_two_rank_driver_over_rccl_equals_single_process():
    """one rank per GPU, backend nccl (= RCCL over xGMI): the flat all-reduce of the outer step, the in-loop J step"""
    _run('w1'); _run('w2n')
    (one,) = _load('w1', 1)
    two = _load('w2n', 2)
    x2 = np.concatenate([two[0]['x6d'], two[1]['x6d']])
    assert np.abs(x2 - one['x6d']).max() < 2e-4
    for k in ('J', 'disc', 'sdisc'):
        assert np.array_equal(two[0][k], two[1][k]), k
        assert np.abs(two[0][k] - one[k]).max() < 5e-5, k


@pytest.mark.skipif(not _multi_gpu(), reason='needs two GPUs')
def test_bench_two_ranks_over_rccl():
    r = _run('bench2n')
    j = json.loads([l for l in r['out'].splitlines() if l.startswith('{')][-1])
    c = j['collective']
    assert j['n_gpus'] == 2 and c['world'] == 2 and c['backend'].startswith('nccl')
    assert sorted(rk[1] for rk in c['ranks_seen']) == [0, 1]            # two different devices
    assert c['allreduce_check'] == 1.0 and c['single_device_debug'] is False
