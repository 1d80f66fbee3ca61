#!/usr/bin/env python3
"""Headline benchmark: pose-refinement inner-loop iterations/sec at batch 4096 per MI355X.

    python bench.py --gpus N --steps K --warmup W
    N > 1 without a torchrun environment: bench.py starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 ... bench.py <same flags>` itself, as a CHILD process and before anything touches the GPU,
    relays rank 0's JSON line and exits with the child's status (a launch under torchrun is used as it is); a world size
    that differs from --gpus is an error.  The line carries the evidence of the rank count (`collective`).

One step = one inner iteration of /root/reference/scripts/optimize.py:220-265 restricted to
BASELINE.json configs[2] ("batch=4096 full loop: 3D-joint loss + pose-discriminator adversarial
term"): rot6d->R, SMPL LBS over 6890 vertices, H36M joint regression, pelvis-centred MSE (x10000),
pose-discriminator forward + adversarial MSE (x10), analytic backward to (pose, orient, betas),
fused Adam -- all through the C ABI (include/jrr.h) on device-resident synthetic data.

N > 1 is weak scaling: every rank owns 4096 poses (global batch 4096*N), the MSE means are
normalised by the GLOBAL batch, and the shared J_regressor is stepped every --j_step_every inner
iterations (reference cadence 100, scripts/optimize.py:300-312) with ONE RCCL all-reduce on its
gradient.  A timed region is EXACTLY --steps iterations between barrier + synchronize pairs; it is
repeated until >= ~3 s of device time has been timed and the MEDIAN region is reported (every
region's time is listed under `repeat_ms_per_step`).  Reported separately, never part of `value`:
`cadence1` (J step + all-reduce after EVERY iteration: BASELINE configs[3] "each step"), the
pose-discriminator update, the folded-regressor mode, BASELINE configs[4] (`config5`), the CPU baseline.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import importlib
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = 'joint-regressor-refinement_amd'

# dense algorithmic FLOP per pose of ONE launch of each MFMA kernel.
# k_lbs_fwd = shape blend + pose blend + skinning blend (12 affine entries) + skinning apply + H36M regressor product;
# the rest-joint regression (992 160 FLOP in the survey's table) is folded into a (24x3)x10 table at model upload and
# is NOT counted.
FLOP_LBS_FWD_PER_POSE = 2 * 20670 * 10 + 2 * 207 * 20670 + 2 * 6890 * 24 * 12 + 2 * 6890 * 3 * 4 + 2 * 17 * 6890 * 3
FLOP_LBS_BWD_PER_POSE = 2 * 17 * 6890 * 3 + 2 * 6890 * 24 * 9 + 2 * 6890 * 9 + 2 * 6890 * 24 * 12   # dverts, T, dvp, dA
FLOP_BLEND_ADJ_PER_POSE = 2 * 217 * 20670                                                             # dF = D . dvp
FLOP_DISC_PER_POSE = 2 * 2 * (24 * (192 + 1024) + 786432 + 1048576 + 1024 + 768)                     # fwd + input-grad
FLOP_DISC_GEMMS_PER_POSE = 2 * 2 * (786432 + 1048576)      # what the four k_disc_gemm launches execute (fc0, fc2, their input adjoints); the
#                                                            per-joint MLP and the heads run in k_prep_fwd_dconv / k_dconv_bwd_reduce
# Joint-sparse skinning (engine info `joint_sparse`): every SMPL vertex has <= 4 skinning influences and a
# tile of 32 consecutive vertices few joints in total, so the skinning products run over the tile's own joint SLOTS instead of 24.
# ONE rule for every roofline figure of this file: the FLOP of the formulation that actually RUNS -- multiplications the kernels
# skip (structural zeros) are not counted, padded rows that they do multiply are not counted either (17 joint rows, not 32):
#   forward   skin blend over `slots` joint slots per tile (8 or 12 per pass; a WIDE tile runs two passes)
#   backward  T recompute over the tile's K steps of 4 slots; dA over the 16-row joint WINDOWS (not 24 joints) of both
#             joint-sparse backward kernels (k_lbs_bwd16 and the role kernel under JRR_BWD16=0)
# The dense-formulation rate is printed beside it (it may exceed the MFMA peak: the skipped multiplications are by zeros).
def skin_slots(model_info):
    """(mean forward slots, mean backward slots) per 32-vertex tile from the model's tile histogram (jrr_model_info)"""
    kjs, hist = model_info['joint_slots'], model_info['tile_joint_histogram']
    if not kjs:
        return 24.0, 24.0
    nt = sum(hist)
    fwd = sum(n * (kjs if k <= kjs else 2 * kjs) for k, n in enumerate(hist)) / nt
    bwd = sum(n * 4 * max(kjs // 4, (k + 3) // 4) for k, n in enumerate(hist)) / nt
    return fwd, bwd
def flop_lbs_fwd(model_info):
    return FLOP_LBS_FWD_PER_POSE - 2 * 6890 * 12 * (24 - skin_slots(model_info)[0])
def flop_lbs_bwd(model_info):
    if not model_info['joint_slots']:
        return FLOP_LBS_BWD_PER_POSE
    return 2 * 17 * 6890 * 3 + 2 * 6890 * skin_slots(model_info)[1] * 9 + 2 * 6890 * 9 + 2 * 6890 * 16 * 12   # dverts, T, dvp, dA (16-row windows)
# matrix instructions per (32-vertex tile x 32 poses) of the forward kernel, for the pipe-busy figure: 110 blend K steps x 3 planes
# (109 pairs of the 218 features + one step of the K-quad layout against zero rows) + 2 x slots x 3 rows on v_mfma_f32_32x32x2_f32
# (64 clocks) and 48 four-block regressor instructions (33 clocks measured)
def fwd_pipe_clocks_per_tile(model_info):
    slots = skin_slots(model_info)[0]
    return (330 + 6 * slots) * 64 + 48 * 33 if model_info['joint_slots'] else 522 * 64
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (= fp32 vector peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak (the side mode's own denominator)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=4096, help='poses per GPU under --scaling weak (BASELINE metric: 4096); the GLOBAL batch under --scaling strong')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak',
                    help='weak (default): every GPU refines --batch poses, the global batch grows with N.  strong: the GLOBAL batch is fixed '
                         'at --batch and rank r refines its contiguous shard of --batch / N poses (BASELINE.json\'s metric string "batch 4096, '
                         'at 1/2/4/8" read as one global batch); `value` then counts iterations of the global batch')
    ap.add_argument('--allow_experiment_lib', action='store_true',
                    help='accept a library named by JRR_LIB (tools/exp variants) instead of the in-tree libjrr_hip.so; the line then says so')
    ap.add_argument('--config', type=int, default=3, choices=[2, 3, 5],
                    help='BASELINE config: 2 = joint loss only, 3 = + pose discriminator (headline), 5 = + soft silhouette')
    ap.add_argument('--j_step_every', type=int, default=100,
                    help='inner iterations per J_regressor step (reference: 100); the timed region always contains at least '
                         'one J step with its all-reduce: the effective cadence is min(j_step_every, steps)')
    ap.add_argument('--min_timed_ms', type=float, default=3000.0, help='repeat the K-step timed region until this much is timed')
    ap.add_argument('--max_repeats', type=int, default=150)
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--cpu_batch', type=int, default=1024, help='cpu_baseline sample batch (scaled to batch-4096 units)')
    ap.add_argument('--cpu_seconds', type=float, default=8.0, help='time budget per cpu_baseline variant')
    ap.add_argument('--no_folded', action='store_true', help='skip the separately reported folded-regressor mode')
    ap.add_argument('--no_bf16x3', action='store_true', help='skip the separately reported split-bf16 side mode of the blend adjoint')
    ap.add_argument('--no_config5', action='store_true', help='skip the separately reported BASELINE configs[4] block')
    ap.add_argument('--no_config2', action='store_true', help='skip the separately reported BASELINE configs[1] block (batch 1024, joint loss only)')
    ap.add_argument('--no_support_tiles', action='store_true', help='skip the side run restricted to the vertex tiles of the regressor\'s support')
    ap.add_argument('--no_skin_variants', action='store_true',
                    help='skip the separately reported 12-joint / dense skinning runs (what a body model with a less coherent vertex order runs)')
    ap.add_argument('--no_driver_blocks', action='store_true',
                    help='skip `driver_outer_batch` / `reference_default` (the real optimize_pose_refiner() timed per outer batch)')
    ap.add_argument('--no_rccl_one_rank', action='store_true', help='skip the cadence-1 host-driven run whose all-reduce is executed by a one-rank RCCL group')
    ap.add_argument('--backend', type=str, default='nccl', help='torch.distributed backend (nccl = RCCL); gloo for debugging')
    ap.add_argument('--single_device', action='store_true',
                    help='debug: every rank uses cuda:0 (exercises the N > 1 code path on a 1-GPU box; use with --backend gloo)')
    return ap.parse_args()


def default_disc_flat(seed=0):
    """Pose-discriminator weights: torch default init under a fixed seed."""
    disc = importlib.import_module(PKG + '.discriminator')
    torch.manual_seed(seed)
    d = disc.Discriminator()
    return d.flat_parameters(), d.state_dict()


def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(model_np, J_np, batch_np, disc_sd, B, budget_s, use_disc, B_full):
    """The oracle (oracle/reference_port.py: torch-CPU restatement in the reference's op order, autograd backward,
    torch.optim.Adam) timed on this box's host cores on the SAME workload, BASELINE.md section 2:
      one_eval            batch B: 1 SMPL evaluation per iteration (the de-duplicated loop the HIP path computes), best thread count
      reference_3_evals   batch B: 3 SMPL evaluations per iteration as scripts/optimize.py:228,231,234 does, same thread count
      single_thread       batch B: one_eval on 1 thread
      full_batch          batch B_full (the metric's own batch): ONE iteration after one warm-up iteration, no scaling
      config1_forward_b4  BASELINE configs[0]: batch 4, SMPL forward + J_regressor evaluation only (scripts/test.py path)
    torch's intra-op pool is far from linear on these small ops, so a 1-iteration sweep picks the thread count."""
    import oracle
    T = torch.from_numpy
    smpl = oracle.OracleSMPL(model_np)
    sd = {k: v.clone() for k, v in disc_sd.items()} if use_disc else None

    def inputs(n):
        x6 = T(batch_np['pose6d'][:n])
        return x6, T(batch_np['betas'][:n]), oracle.move_pelvis(T(batch_np['gt_j3d'][:n]))

    def run(n_it, evals=1, n=B):
        x6, betas, gt_c = inputs(n)
        t0 = time.perf_counter()
        oracle.refine_poses(smpl, T(J_np), x6[:, :1], x6[:, 1:], betas, gt_c, n_it, disc_sd=sd, smpl_evals=evals)
        return time.perf_counter() - t0

    def timed(evals, threads):
        torch.set_num_threads(threads)
        dt1 = run(1, evals)                                  # warm-up + estimate
        n = max(1, min(40, int(budget_s / max(dt1, 1e-3))))
        dt = run(n, evals)
        return {'it_s_at_sample_batch': round(n / dt, 4), 'iterations': n, 'seconds': round(dt, 2), 'threads': threads,
                'smpl_evals_per_iter': evals, 'batch': B}

    ncpu = os.cpu_count() or 1
    phys = physical_cores() or ncpu
    best_t, best_n = None, 1
    sweep = sorted({min(ncpu, n) for n in (8, 16, 32, 64, 128, phys, ncpu) if n <= max(phys, 8) or n == phys})
    for nt in sweep:
        torch.set_num_threads(nt)
        run(1)
        dt = run(1)
        if best_t is None or dt < best_t:
            best_t, best_n = dt, nt
    out = {'one_eval': timed(1, best_n), 'reference_3_evals': timed(3, best_n), 'single_thread': timed(1, 1)}
    torch.set_num_threads(best_n)
    if B_full > B and B_full <= batch_np['pose6d'].shape[0]:
        run(1, 1, B_full)
        dt = run(1, 1, B_full)
        out['full_batch'] = {'it_s_at_sample_batch': round(1.0 / dt, 4), 'iterations': 1, 'seconds': round(dt, 2),
                             'threads': best_n, 'smpl_evals_per_iter': 1, 'batch': B_full}
    # BASELINE configs[0]: batch 4, forward + regressor only
    x6, betas, _ = inputs(4)
    Jt = T(J_np)
    def fwd4():
        Ro = oracle.rot6d_to_rotmat(x6[:, :1].reshape(-1, 6)).view(-1, 1, 3, 3)
        Rp = oracle.rot6d_to_rotmat(x6[:, 1:].reshape(-1, 6)).view(-1, 23, 3, 3)
        with torch.no_grad():
            return oracle.find_joints(smpl, betas, Ro, Rp, Jt)
    fwd4()
    t0 = time.perf_counter(); n4 = 0
    while time.perf_counter() - t0 < 1.0:
        fwd4(); n4 += 1
    dt4 = time.perf_counter() - t0
    out['config1_forward_b4'] = {'evals_per_s': round(n4 / dt4, 2), 'ms_per_eval': round(dt4 / n4 * 1e3, 3), 'batch': 4,
                                 'threads': best_n, 'workload': 'BASELINE configs[0]: SMPL forward + J_regressor eval, batch 4'}
    out['thread_sweep'] = {'tried': sweep, 'physical_cores': phys, 'logical_cpus': ncpu, 'picked': best_n}
    return out, best_n


def physical_cores():
    """distinct (physical id, core id) pairs of /proc/cpuinfo (SMT siblings counted once)"""
    try:
        cores, pid = set(), None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                pid = line.split(':', 1)[1].strip()
            elif line.startswith('core id'):
                cores.add((pid, line.split(':', 1)[1].strip()))
        return len(cores) or None
    except OSError:
        return None


def provenance(allow_experiment_lib):
    """which binary and which knobs this line measured: sha256 (first 16 hex digits) and path of the library the process loaded, every
    JRR_* variable of the environment (api.hip reads its experiment knobs from there).  JRR_LIB -- a variant library of tools/exp --
    is refused unless --allow_experiment_lib says the line is an experiment."""
    import hashlib
    lib_mod = importlib.import_module(PKG + '._lib')
    env = {k: v for k, v in sorted(os.environ.items()) if k.startswith('JRR_')}
    if 'JRR_LIB' in env and not allow_experiment_lib:
        raise SystemExit('bench.py: JRR_LIB is set (an experiment library of tools/exp): pass --allow_experiment_lib to measure it, '
                         'or unset it to measure the in-tree libjrr_hip.so')
    path = os.path.abspath(lib_mod.LIB_PATH)
    with open(path, 'rb') as f:
        sha = hashlib.sha256(f.read()).hexdigest()[:16]
    return {'lib_path': os.path.relpath(path, ROOT) if path.startswith(ROOT) else path, 'lib_sha16': sha, 'lib_bytes': os.path.getsize(path),
            'in_tree_lib': path == os.path.join(ROOT, PKG, 'libjrr_hip.so'), 'jrr_env': env,
            'note': 'sha256[:16] of the shared library this process loaded; jrr_env = every JRR_* variable set (none = the shipped kernel selection)'}


def self_launch(a):
    """--gpus N > 1 outside torchrun: start the N ranks as a CHILD `torch.distributed.run` before anything here has touched
    the GPU (a process that has initialised the GPU must never exec / be replaced), relay rank 0's JSON line, exit with the
    child's status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    json_lines = [l for l in lines if l.startswith('{')]
    for l in lines:
        if not l.startswith('{'):
            print(l, file=sys.stderr)
    if json_lines:
        print(json_lines[-1])
    sys.stdout.flush()
    sys.exit(r.returncode if r.returncode != 0 or json_lines else 1)


def block_folded(c):
    """folded-regressor mode (a different algorithm with its own denominator)"""
    # ---- folded-regressor mode (a different algorithm with its own denominator) ----
    folded = None
    if not c.a.no_folded and not c.use_sil:
        folded = c.side_run(c.eng_mod.FLAG_FOLDED | (c.eng_mod.FLAG_POSE_DISC if c.use_disc else 0))
        folded.update({'algorithmic_flop_per_pose_iter': 2 * (2 * 1224 * 218) + 4 * (17 * 3 * 24 * 4 * 2),
                       'note': 'joints = A.(H F) with H = sum_v Jn W D contracted once per J update (exact re-association, '
                               'same losses/updates; no vertices).  Separate mode, separate denominator: not the headline.'})
    return folded


def block_bf16x3(c):
    """SIDE MODE JRR_FLAG_BLEND_BF16X3: separately labelled, its own denominator, never `value`"""
    # ---- SIDE MODE, a different arithmetic type: the blend-basis adjoint as a split-bf16 product (JRR_FLAG_BLEND_BF16X3).  Reported
    #      like folded_mode: separately labelled, its own denominator, NEVER `value` (the reference computes in fp32; so does the
    #      headline).  Its parity bounds are its own: tests/test_gpu_bf16x3.py ----
    bf16x3 = None
    if not c.a.no_bf16x3 and not c.use_sil and c.use_disc and hasattr(c.eng_mod, 'FLAG_BLEND_BF16X3'):
        bf16x3 = c.side_run(c.eng_mod.FLAG_KEEP_VERTS | c.eng_mod.FLAG_POSE_DISC | c.eng_mod.FLAG_BLEND_BF16X3, force_profile=True)
        adj_ms = bf16x3.get('kernels_ms', {}).get('k_gemm_tn_blend_adjoint')
        # issued work of the kernel: three 32x32x16 bf16 matrix instructions per accumulator tile and 16 vertices
        bf_flop = 3 * 2.0 * 224 * (3 * 6912) * c.B
        bf16x3.update({
            'dtype': 'bf16x3, f32 accumulate (blend adjoint only; every other kernel exact f32)',
            'blend_adjoint': {'avg_launch_ms': adj_ms, 'exact_f32_kernel_ms': round(c.prof['k_gemm_tn_blend_adjoint'][0], 4),
                              'issued_bf16_mfma_flop_per_launch': bf_flop,
                              'achieved_tflops_bf16_issued': round(bf_flop / (adj_ms * 1e-3) / 1e12, 1) if adj_ms else None,
                              'peak_bf16_mfma_tflops': PEAK_BF16_MFMA_TFLOPS,
                              'frac_of_bf16_mfma_peak': round(bf_flop / (adj_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4) if adj_ms else None,
                              'operand_bytes_through_lds_per_launch': int((c.B // 128) * 1296 * (14336 + 8192)),
                              'bound': 'operand traffic (the split operands are as large as the f32 ones; the matrix part is 5.3 x shorter)'},
            'note': 'NOT the headline and not the reference\'s arithmetic: operands of the blend-basis adjoint taken as bf16 hi + lo, three bf16 '
                    'products per exact one, f32 accumulation (relative error of the product ~3e-5; joints after 100 iterations within 1e-4 m of '
                    'the exact engine and of the oracle: tests/test_gpu_bf16x3.py).  `value` stays the exact-f32 run above.'})
    return bf16x3


def block_support_tiles(c):
    """the iteration optimize.py runs by default: on the vertices / tiles of the regressor's support"""
    # ---- the same iteration on the vertex tiles of the regressor's SUPPORT only (JRR_FLAG_SUPPORT_TILES; what optimize.py runs by
    #      default).  Exact: the other tiles meet a zero block of the regressor and a zero vertex adjoint.  Reported beside the
    #      headline, which keeps running all 6890 vertices (north_star: "linear blend skinning over 6890 vertices") ----
    support_tiles = None
    if not c.a.no_support_tiles and not c.use_sil:
        # the body uploaded as optimize.py uploads it: with the regressor's positive columns as a hint for the library's internal vertex
        # order (jrr_model_create_hinted: the support stored first -- ceil(58 / 32) = 2 tiles instead of one per entry)
        hmodel = c.eng_mod.DeviceModel(c.model_np, c.dev, hint_vertices=np.nonzero((c.J_np > 0).any(0))[0])
        support_tiles = c.side_run(c.eng_mod.FLAG_KEEP_VERTS | (c.eng_mod.FLAG_POSE_DISC if c.use_disc else 0), tiles=True, model=hmodel)
        support_tiles['model'] = hmodel.info
        unhinted = c.side_run(c.eng_mod.FLAG_KEEP_VERTS | (c.eng_mod.FLAG_POSE_DISC if c.use_disc else 0), tiles=True)
        support_tiles['without_vertex_order_hint'] = {k: unhinted[k] for k in ('value', 'ms_per_step', 'vertex_tiles', 'kernels_ms')}
        nt = support_tiles['vertex_tiles']['run']
        km = support_tiles.get('kernels_ms', {})
        support_tiles['small_launches_ms'] = round(sum(km.get(k_, 0.0) for k_ in ('k_prep_fwd', 'k_joints_loss', 'k_prep_bwd', 'k_shape_disc')), 4)
        support_tiles['skinning_launches_ms'] = round(sum(km.get(k_, 0.0) for k_ in ('k_lbs_fwd', 'k_lbs_bwd', 'k_gemm_tn_blend_adjoint', 'k_sup_step')), 4)
        per_vertex = support_tiles['support_vertices']['per_vertex_iteration']
        # FLOP it runs: the listed tiles at the 8-slot rate + a second pass for each of them that is wide (all, with the hint); per
        # vertex (round 6): the two 192 x 224 products of the support's coordinate rows, the rest is vector work
        fl = (flop_lbs_fwd(c.dmodel.info) + flop_lbs_bwd(c.dmodel.info) + FLOP_BLEND_ADJ_PER_POSE) * nt / 216 + (FLOP_DISC_PER_POSE if c.use_disc else 0)
        if per_vertex:
            fl = 2 * (2 * 192 * 224) + (FLOP_DISC_PER_POSE if c.use_disc else 0)
        support_tiles.update({
            'flop_per_pose_iter_it_runs': round(fl), 'achieved_tflops': round(fl * c.B / (support_tiles['ms_per_step'] * 1e-3) / 1e12, 2),
            'frac_of_f32_mfma_peak': round(fl * c.B / (support_tiles['ms_per_step'] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            'note': 'joint-loss iterations read the vertices through the regressor only: a vertex without a positive regressor entry '
                    'adds exact zeros to the joints and receives a zero adjoint.  Round 6: when the support has <= 64 vertices (58 here) the '
                    'iteration runs per VERTEX in ONE launch per 32-pose group (prep.hip k_sup_step: chain forward, support-vertex SMPL '
                    'forward, loss, backward, per-joint MLP adjoint, chain adjoint + Adam, per-joint MLP forward of the next iteration) behind '
                    'the four discriminator GEMM launches -- 5 launches where the tile lists of round 4 took 13; results equal the all-tiles '
                    'run up to the order of the sums (tests/test_gpu_round4.py, tests/test_gpu_trajectory.py).  NOT the headline: `value` '
                    'runs all 216 tiles.'})
        if c.B >= 1024 and c.a.config == 3 and not c.a.no_config2:
            c2t = c.side_run(c.eng_mod.FLAG_KEEP_VERTS, Bs=1024, disc=False, tiles=True, model=hmodel)
            support_tiles['config2_batch1024_joint_loss_only'] = c2t
    return support_tiles


def block_config5(c):
    """BASELINE configs[4]: + soft-silhouette loss inside the inner loop"""
    # ---- BASELINE configs[4]: + soft-silhouette loss inside the inner loop, separately timed ----
    config5 = None
    if not c.a.no_config5 and not c.use_sil and c.use_disc:
        config5 = c.side_run(c.eng_mod.FLAG_KEEP_VERTS | c.eng_mod.FLAG_POSE_DISC | c.eng_mod.FLAG_SILHOUETTE, c.silhouette_setup)
        config5['workload'] = 'BASELINE configs[4]: configs[2] + soft-silhouette loss (224x224 rasteriser as HIP kernel) in the inner loop'
        # the rasteriser is integer / vector-ALU work, not a GEMM: its line is the VALU issue rate (wave-instructions per launch
        # from the PMC pass of tools/prof_c5.sh, static) over the live duration, against 1 wave-instruction per SIMD per 2 clocks
        try:
            pm = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'))).get('k_sil_raster_adj_pmc_per_launch_b4096') if c.B == 4096 else None
        except Exception:
            pm = None
        ras_ms = config5.get('kernels_ms', {}).get('silhouette_fwd_bwd')
        if pm and ras_ms:
            peak = 256 * 4 * 2.4e9 / 2
            config5['rasteriser_issue'] = {'bound': 'valu issue', 'valu_wave_insts_per_launch': pm['SQ_INSTS_VALU'],
                                           'valu_wave_insts_per_pose': round(pm['SQ_INSTS_VALU'] / c.B), 'avg_launch_ms': ras_ms,
                                           'achieved_ginst_s': round(pm['SQ_INSTS_VALU'] / (ras_ms * 1e-3) / 1e9, 1), 'peak_ginst_s': round(peak / 1e9, 1),
                                           'frac': round(pm['SQ_INSTS_VALU'] / (ras_ms * 1e-3) / peak, 4),
                                           'parked_wave_cycle_frac': round(pm['SQ_WAIT_ANY'] / pm['SQ_WAVE_CYCLES'], 3),
                                           'source': 'profiles/pmc_traffic.json (static PMC counts, live duration)'}
    return config5


def block_config2(c):
    """BASELINE configs[1]: batch 1024, 3D-joint loss only, all tiles"""
    # ---- BASELINE configs[1]: batch 1024, 3D-joint loss only, separately timed with its own roofline fraction ----
    config2 = None
    if not c.a.no_config2 and c.a.config == 3 and c.B >= 1024:
        config2 = c.side_run(c.eng_mod.FLAG_KEEP_VERTS, Bs=1024, disc=False)
        fl2 = flop_lbs_fwd(c.dmodel.info) + flop_lbs_bwd(c.dmodel.info) + FLOP_BLEND_ADJ_PER_POSE
        config2.update({'workload': 'BASELINE configs[1]: batch=1024 pose optimisation, 3D-joint L2 loss only',
                        'whole_step': {'flop_per_pose_iter': fl2, 'achieved': round(fl2 * 1024 / (config2['ms_per_step'] * 1e-3) / 1e12, 2),
                                       'frac': round(fl2 * 1024 / (config2['ms_per_step'] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                       'dense_formulation_flop_per_pose_iter': FLOP_LBS_FWD_PER_POSE + FLOP_LBS_BWD_PER_POSE + FLOP_BLEND_ADJ_PER_POSE}})
    return config2


def block_skin_variants(c):
    """the 12-slot / dense / wide-tile / random-file-order bodies"""
    # ---- what a body model with a less coherent vertex order runs: the 12-joint-per-tile kernels and the dense kernels
    #      (the synthetic body's ring-major vertex order is what lets the headline use the 8-joint kernels) ----
    skin_variants = None
    if not c.a.no_skin_variants and not c.use_sil:
        skin_variants = {}
        variants = [('skin12', {'JRR_SKIN_JOINTS': '12'}, c.model_np), ('dense', {'JRR_DENSE_SKINNING': '1'}, c.model_np),
                    # a body whose FILE order means nothing to the tiles (seeded random): the library stores it in its own order along the
                    # kinematic chains (invisible at the API); tiles that are still wide would pay a second pass themselves
                    ('capsules_random_file_order', {}, c.sm.synthetic_smpl(1234, kind='capsules')),
                    # the benchmarked body with ONE tile skinned by 13 joints: it pays a second pass itself, the model stays in its class
                    ('one_13_joint_tile', {}, c.sm.with_wide_tile(c.model_np, 100, 13)),
                    # ... and both together: the 12-slot kernels WITH a wide tile (the instantiation that used to spill registers)
                    ('one_13_joint_tile_skin12', {'JRR_SKIN_JOINTS': '12'}, c.sm.with_wide_tile(c.model_np, 100, 13))]
        for name, env, mnp in variants:
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)                       # read by jrr_model_create
            try:
                mdl = c.eng_mod.DeviceModel(mnp, c.dev)
            finally:
                for k, vv in old.items():
                    if vv is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = vv
            r = c.side_run(c.eng_mod.FLAG_KEEP_VERTS | (c.eng_mod.FLAG_POSE_DISC if c.use_disc else 0), model=mdl)
            fl = flop_lbs_fwd(mdl.info) + flop_lbs_bwd(mdl.info) + FLOP_BLEND_ADJ_PER_POSE + (FLOP_DISC_PER_POSE if c.use_disc else 0)
            r.update({'flop_per_pose_iter': fl, 'achieved_tflops': round(fl * c.B / (r['ms_per_step'] * 1e-3) / 1e12, 2),
                      'frac_of_f32_mfma_peak': round(fl * c.B / (r['ms_per_step'] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                      'forced_by': env, 'model': mdl.info})
            skin_variants[name] = r
            del mdl
    return skin_variants


def block_driver(c, support_tiles):
    """the ENTRY POINT per outer batch: optimize_pose_refiner() on synthetic batches (driver_outer_batch, reference_default)"""
    # ---- the ENTRY POINT, per outer batch (/root/reference/scripts/optimize.py:144-337 is one call of optimize_pose_refiner() per run:
    #      H->D copy, [camera pre-fit], 100 inner iterations, D updates, J step, two evaluations, a log record).  The real driver of the
    #      package on synthetic batches; the first batch (engine set-up, code-object loads) is discarded ----
    def driver_run(batch, extra, n_batches=7, inner=100):
        argsmod = importlib.import_module(PKG + '.args')
        argsmod._LazyArgs._ns = argsmod.get_args(['--batch_size', str(batch), '--synthetic_batches', str(n_batches), '--inner_iters', str(inner),
                                                   '--synthetic', '--device', str(c.dev), '--smpl_dir', '/nonexistent', '--j_regressor_init',
                                                   '/nonexistent'] + list(extra))
        optm = importlib.import_module(PKG + '.optimize')
        recs = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        optm.optimize_pose_refiner(log=recs.append)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        kept = recs[1:] or recs
        sec = statistics.median([r['seconds_batch'] for r in kept])
        return {'flags': ' '.join(extra) or '(defaults)', 'batch': batch, 'inner_iters': inner, 'outer_batches_run': len(recs), 'first_batch_discarded': len(recs) > 1,
                'seconds_per_outer_batch': round(sec, 5), 'seconds_per_outer_batch_each': [round(r['seconds_batch'], 5) for r in recs],
                'loop_and_outer_step_seconds': round(statistics.median([r['seconds'] for r in kept]), 5),
                'it_s_through_the_entry_point': round(inner / sec, 2), 'vertex_tiles_run': kept[-1]['vertex_tiles_run'],
                'support_vertices_run': kept[-1].get('support_vertices_run'),
                'whole_call_seconds_incl_synthetic_batch_generation_and_setup': round(wall, 3),
                'joint_loss_last': kept[-1]['joint_loss'], 'mpjpe_last': kept[-1].get('mpjpe')}

    driver_outer, ref_default = None, None
    if c.dist is None and not c.a.no_driver_blocks and c.a.config == 3 and not c.use_sil:
        outer_ms = (c.d_ms or 0.0) + c.j_ms                  # pose-D update + J step as timed above (stand-alone, this engine)
        driver_outer = {}
        for name, extra, step_ms in (('all_vertex_tiles', ['--all_vertex_tiles'], c.inner_ms),
                                     ('support_tiles_default', [], support_tiles['ms_per_step'] if support_tiles else None)):
            r = driver_run(c.B, extra)
            if step_ms:
                kern = (r['inner_iters'] * step_ms + outer_ms) * 1e-3
                r.update({'inner_ms_per_step_of_this_mode': round(step_ms, 4), 'outer_step_ms': round(outer_ms, 3),
                          'kernel_seconds_expected': round(kern, 5),
                          'share_not_inner_loop_or_outer_step': round(1.0 - kern / r['seconds_per_outer_batch'], 4)})
            driver_outer[name] = r
        driver_outer['note'] = ('optimize_pose_refiner() of this package (the reference entry point restated) on 7 synthetic outer batches, first one '
                                'discarded, median of the rest; seconds_per_outer_batch = wall time from the batch arriving to its record (H->D copies, fresh Adam '
                                'state, 100 inner iterations in ONE C call, D update, J step, evaluations, the one read-back); '
                                'share_not_inner_loop_or_outer_step = 1 - (100 x ms_per_step + pose-D update + J step) / that')
        # the reference's OWN default run (scripts/args.py:8 batch 256; all five terms of scripts/optimize.py:252-253 with the 1000-step
        # camera pre-fit of :187-199)
        ref_default = driver_run(256, ['--shape_disc', '--reprojection', '--silhouette'], n_batches=6)
        ref_default['workload'] = ('scripts/args.py:8 default batch 256, all five loss terms (2-D joints, silhouette, 3-D joints, pose-D, shape-D), '
                                   '1000-step camera pre-fit, 100 inner iterations, D updates + J step per outer batch')
    return driver_outer, ref_default



def build_line(c):
    """the JSON line of rank 0 from the measurements of main() (c: its locals): headline, roofline of the dominant kernel and of the whole
    step, per-kernel times, cadence-1 blocks, outer-step rates, and the separately reported blocks"""
    ms_per_step = c.elapsed / c.a.steps * 1e3
    it_s = c.a.steps / c.elapsed
    dom_ms, dom_n = c.prof['k_lbs_fwd']
    kjs = int(c.eng.info.get('joint_sparse') or 0)      # joint slots per vertex tile and pass the LBS kernels multiply by (0: all 24)
    sparse = kjs > 0
    flop_fwd = flop_lbs_fwd(c.dmodel.info)
    flop_bwd = flop_lbs_bwd(c.dmodel.info)
    achieved = flop_fwd * c.B / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    traffic, pmc = None, {}
    tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(tpath):
        try:
            pmc = json.load(open(tpath)) if c.B == 4096 else {}
            traffic = pmc.get('k_lbs_fwd_hbm_bytes_per_launch')
        except Exception:
            traffic, pmc = None, {}

    def issued(name, ms):
        """the ISSUED matrix work of a kernel: SQ_INSTS_MFMA of one launch (static PMC pass, profiles/pmc_traffic.json:
        `mfma_issued_b4096`) x FLOP per instruction over the LIVE duration -- beside the algorithmic figure so that a reader
        sees the two agree; only for the benchmarked body at batch 4096"""
        rec = (pmc.get('mfma_issued_b4096') or {}).get(name)
        if not rec or not ms or c.dmodel.info['wide_tiles'] or not sparse:
            return None
        fl = rec['flop_per_launch']
        return {'sq_insts_mfma': rec['sq_insts_mfma'], 'flop_per_launch': fl, 'tflops': round(fl / (ms * 1e-3) / 1e12, 2),
                'frac': round(fl / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4), 'source': 'profiles/pmc_traffic.json (static count, live duration)'}
    step_flop = flop_fwd + flop_bwd + FLOP_BLEND_ADJ_PER_POSE + (FLOP_DISC_PER_POSE if c.use_disc else 0)
    step_flop_dense = FLOP_LBS_FWD_PER_POSE + FLOP_LBS_BWD_PER_POSE + FLOP_BLEND_ADJ_PER_POSE + (FLOP_DISC_PER_POSE if c.use_disc else 0)
    c1_ms = c.c1_el / c.a.steps * 1e3
    out = {
        # BASELINE.json's metric string.  --scaling weak (default): batch = poses per GPU, `value` = N x iterations/s of 4096-pose batches;
        # --scaling strong: batch = the global batch, sharded N ways, `value` = iterations/s of that one batch
        'metric': 'pose-refinement iters/sec, batch 4096, at 1/2/4/8 MI355X',
        'value': round(it_s * c.agg, 3), 'unit': f'it/s (x{c.UB} poses)', 'n_gpus': c.world, 'steps': c.a.steps, 'warmup': c.a.warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': c.a.scaling, 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'BASELINE configs[{c.a.config - 1}]: batch={c.B}/GPU inner loop, 3D-joint loss'
                               + (' + pose-discriminator adversarial term' if c.use_disc else '')
                               + (' + soft-silhouette loss (224x224 rasteriser)' if c.use_sil else ''),
                   'global_batch': c.B * c.world, 'poses_per_gpu': c.B, 'poses_per_sec': round(it_s * c.world * c.B, 1),
                   'j_step_every': c.cadence, 'j_steps_in_timed_regions': c.nj_region, 'parallelism': f'dp{c.world}', 'joint_loss_last': c.loss_joint,
                   'timed_regions': len(c.regions), 'timed_steps': len(c.regions) * c.a.steps,
                   'value_is': 'all timed regions: timed steps / timed seconds (the J step follows every j_step_every-th iteration, counted '
                               'across regions: the reference cadence, not one J step per region)',
                   'median_region_ms_per_step': round(statistics.median(c.regions) / c.a.steps * 1e3, 4),
                   'j_steps_per_region': c.region_nj if len(set(c.region_nj)) > 1 else c.region_nj[0],
                   'repeat_ms_per_step': [round(r / c.a.steps * 1e3, 4) for r in c.regions],
                   'host_calls_per_region': 'one C call (jrr_refine_run_j_steps)' if c.dist is None else 'refine_run + j_regressor_grad + all_reduce + j_step_apply per J step',
                   'forward_reuse_after_j_step': False,
                   'vertex_tiles': 'all 216 (every iteration skins all 6890 vertices; the iteration restricted to the tiles of the '
                                   'regressor\'s support -- what optimize.py runs by default -- is the separate block `support_tiles`)',
                   'geometry': dict(c.eng.info, **c.dmodel.info)},
        'collective': c.collective,
        'provenance': c.prov,
        'per_rank_ms_per_step': {'min': round(min(c.per_rank_ms), 4), 'median': round(statistics.median(c.per_rank_ms), 4), 'max': round(max(c.per_rank_ms), 4),
                                 'each': [round(x, 4) for x in c.per_rank_ms],
                                 'note': 'every rank\'s own time per step over the headline regions (before the closing barrier); `ms_per_step` is the max over ranks incl. barriers'},
        'roofline': {'bound': 'mfma', 'kernel': 'k_lbs_fwd<true,false>', 'achieved': round(achieved, 2),
                     'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                     'traffic': traffic,
                     'traffic_source': 'profiles/pmc_traffic.json (static: rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes of an earlier '
                                       'run of this command, gfx950-corrected; NOT measured in this run)' if traffic else None,
                     'avg_launch_ms': round(dom_ms, 4), 'launches_timed': dom_n,
                     # HBM side of the same kernel (PMC bytes per launch / live duration) against the 8 TB/s spec
                     'hbm_gb_s': round(traffic / (dom_ms * 1e-3) / 1e9, 1) if traffic and dom_ms > 0 else None,
                     'hbm_frac_of_8tb_s': round(traffic / (dom_ms * 1e-3) / 8e12, 4) if traffic and dom_ms > 0 else None,
                     'algorithmic_flop_per_launch': flop_fwd * c.B,
                     'issued_work': issued('k_lbs_fwd', dom_ms),
                     'formulation': (f'joint-sparse skinning: each 32-vertex tile multiplies by its own joints only, {kjs} slots per pass '
                                     f"({c.dmodel.info['wide_tiles']} wide tiles run a second pass; exact: the skipped terms are zeros); "
                                     'FLOP counted for THIS formulation') if sparse
                                    else 'dense skinning (SURVEY.md section 8d counts)',
                     'dense_formulation': {'flop_per_launch': FLOP_LBS_FWD_PER_POSE * c.B,
                                           'rate_tflops': round(FLOP_LBS_FWD_PER_POSE * c.B / (dom_ms * 1e-3) / 1e12, 2) if dom_ms > 0 else None,
                                           'note': 'the reference formulation\'s FLOP over the same time; not a roofline figure'},
                     # the WHOLE inner iteration against the same peak: algorithmic FLOP of its four MFMA stages
                     # (k_lbs_fwd + k_lbs_bwd + blend adjoint + discriminator fwd/input-grad) over the time of a region of
                     # inner iterations ONLY (no J step, no forward reuse; median of 3 regions)
                     'whole_step': {'flop_per_pose_iter': step_flop,
                                    'achieved': round(step_flop * c.B / (c.inner_ms * 1e-3) / 1e12, 2),
                                    'frac': round(step_flop * c.B / (c.inner_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                    'inner_only_ms_per_step': round(c.inner_ms, 4), 'timed': 'its own regions: no J step, no forward reuse',
                                    'dense_formulation_flop_per_pose_iter': step_flop_dense,
                                    'dense_formulation_rate_tflops': round(step_flop_dense * c.B / (c.inner_ms * 1e-3) / 1e12, 2)},
                     # in-kernel probe (s_memtime / s_memrealtime, workgroup 0 / wave 0): the clock the chip holds on
                     # this kernel (`peak` assumes 2.4 GHz), hence the MFMA-pipe utilisation of the whole launch, and
                     # how much of the launch the FIRST-dispatched workgroup is resident (the two workgroups of a CU
                     # do not progress evenly: the older one finishes early)
                     'sustained_clock_ghz': round(c.probe[0] / c.probe[4], 3) if c.probe[4] else None,
                     # matrix-pipe busy fraction of the launch: issue clocks of every matrix instruction (32x32x2: 64 clocks, the 48
                     # four-block regressor instructions per tile: 33) per SIMD over the launch duration at the sustained clock
                     'mfma_pipe_utilisation': round(216 * (c.eng.info['BP'] / 32) * fwd_pipe_clocks_per_tile(c.dmodel.info) / 1024
                                                    / (dom_ms * 1e-3 * c.probe[0] / c.probe[4] * 1e9), 4) if c.probe[4] and dom_ms > 0 else None,
                     'mfma_pipe_utilisation_note': 'instruction-priced (64 / 33 issue clocks); the algorithmic-FLOP ratio achieved / (peak x clock / 2.4) reads ~1 % higher',
                     'first_workgroup_resident_frac': round(c.probe[4] * 1e-6 / dom_ms, 3) if c.probe[4] and dom_ms > 0 else None},
        'kernels_ms': {k: round(t, 4) for k, (t, n) in c.prof.items() if n},
        # the other matrix-core launches of the iteration against the same peak, each on the FLOP of the formulation it runs
        'kernels_roofline': {name: {'flop_per_launch': fl * c.B, 'achieved_tflops': round(fl * c.B / (c.prof[cls][0] * 1e-3) / 1e12, 2),
                                    'frac': round(fl * c.B / (c.prof[cls][0] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                    'issued_work': issued(pm_name, c.prof[cls][0])}
                             for name, cls, fl, pm_name in (('k_lbs_bwd', 'k_lbs_bwd', flop_bwd, 'k_lbs_bwd'),
                                                            ('k_blend_adjoint', 'k_gemm_tn_blend_adjoint', FLOP_BLEND_ADJ_PER_POSE, 'k_blend_adjoint'),
                                                            ('pose_disc_gemms (4 launches)', 'pose_disc_gemms', FLOP_DISC_GEMMS_PER_POSE, 'pose_disc_gemms'))
                             if c.prof.get(cls, (0, 0))[1] and c.prof[cls][0] > 0},
        'j_step': {'ms': round(c.j_ms, 3), 'includes': 'SMPL fwd + dJ product + all-reduce + Adam(J) + renormalise',
                   'allreduce_bytes': c.xch.nbytes, 'allreduce_payload': 'regressor support [17][128]' if c.xch.compact else 'dense (17,6890)',
                   'in_timed_regions': c.nj_region, 'host_calls': 2 + (1 if c.dist is not None else 0)},
        'cadence1': {'value': round(c.a.steps / c.c1_el * c.agg, 3), 'unit': f'it/s (x{c.UB} poses)', 'ms_per_step': round(c1_ms, 4),
                     'j_step_every': 1, 'timed_regions': len(c.c1_regions), 'value_is': 'median region',
                     'repeat_ms_per_step': [round(r / c.a.steps * 1e3, 4) for r in c.c1_regions],
                     'spread_frac': round((max(c.c1_regions) - min(c.c1_regions)) / c.c1_el, 4),
                     'forward_reuse_after_j_step': True,
                     'note': 'BASELINE configs[3] / north_star "all-reduce on the J_regressor gradient each step": '
                             'the J step (+ its all-reduce) after EVERY inner iteration, timed like `value`; the iteration after a J '
                             'step re-regresses its joints from the J step\'s stored vertices (same poses: explicit reuse)'},
    }
    if c.c1h_el is not None:
        out['cadence1_host_driven'] = {
            'value': round(c.a.steps / c.c1h_el, 3), 'unit': f'it/s (x{c.B} poses)', 'ms_per_step': round(c.c1h_el / c.a.steps * 1e3, 4),
            'timed_regions': len(c.c1h_regions), 'value_is': 'median region',
            'repeat_ms_per_step': [round(r / c.a.steps * 1e3, 4) for r in c.c1h_regions],
            'spread_frac': round((max(c.c1h_regions) - min(c.c1h_regions)) / c.c1h_el, 4),
            'allreduce_bytes_it_would_send': c.xch.nbytes,
            'with_rccl_one_rank_allreduce': None,
            'note': 'cadence 1 through the call sequence N > 1 ranks execute (refine_run_after_j_step -> j_regressor_grad_support -> '
                    '[all-reduce: a no-op at world size 1] -> j_step_apply_support per iteration, host-driven): bounds the multi-GPU '
                    'cadence-1 cost of everything but the collective itself'}
    # outer-step work.  `value` already contains the J step (+ all-reduce) at cadence
    # `j_step_every`; the pose-D update is timed separately.  Two derived rates: everything at the measured
    # cadence, and everything after EVERY inner iteration (cadence 1).
    out['outer_step'] = {'j_step_ms': round(c.j_ms, 3), 'pose_d_update_ms': None if c.d_ms is None else round(c.d_ms, 3),
                         'inner_only_ms_per_step': round(c.inner_ms, 4),
                         'it_s_incl_pose_d_update_at_cadence': round(c.agg / ((ms_per_step + (c.d_ms or 0.0) / c.cadence) * 1e-3), 3),
                         'it_s_all_outer_work_every_iteration': round(c.agg / ((c1_ms + (c.d_ms or 0.0)) * 1e-3), 3),
                         'j_allreduce_bytes': c.xch.nbytes, 'pose_d_allreduce_bytes': 1840153 * 4 if c.use_disc else 0}
    if c.collective_cost is not None:
        out['collective_cost'] = c.collective_cost
    if c.driver_outer is not None:
        out['driver_outer_batch'] = c.driver_outer
    if c.ref_default is not None:
        out['reference_default'] = c.ref_default
    if c.folded is not None:
        out['folded_mode'] = c.folded
    if c.bf16x3 is not None:
        out['bf16x3_mode'] = c.bf16x3
    if c.support_tiles is not None:
        out['support_tiles'] = c.support_tiles
    if c.config2 is not None:
        out['config2'] = c.config2
    if c.config5 is not None:
        out['config5'] = c.config5
    if c.skin_variants is not None:
        out['skin_variants'] = c.skin_variants
    return out


def main():
    a = parse()
    prov = provenance(a.allow_experiment_lib)      # before anything touches the GPU: a refused JRR_LIB costs nothing
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(a)                 # does not return; nothing above has touched the GPU
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != a.gpus:
        raise SystemExit(f'bench.py: --gpus {a.gpus} but the launcher started {world} rank(s)')
    if not torch.cuda.is_available():
        raise SystemExit(f'bench.py needs an MI355X (the HIP path has no CPU fallback) [rank {rank} of {world}]')
    if a.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if a.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(a.backend)
    # evidence of what the collective spans: every rank's (rank, device index, device name, pid) and a sum of ranks
    me = [rank, torch.cuda.current_device(), torch.cuda.get_device_name(dev), os.getpid()]
    ranks_seen, ar_check = [me], 0.0
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, me)
        ranks_seen = gathered
        t = torch.tensor([float(rank)], device=dev)
        dist.all_reduce(t)
        ar_check = float(t.item())
    collective = {'backend': (a.backend + (' (RCCL)' if a.backend == 'nccl' else '')) if dist is not None else None, 'world': world,
                  'ranks_seen': ranks_seen, 'allreduce_check': ar_check, 'allreduce_expected': world * (world - 1) / 2,
                  'single_device_debug': bool(a.single_device)}

    sm = importlib.import_module(PKG + '.smpl_model')
    eng_mod = importlib.import_module(PKG + '.engine')
    strong = a.scaling == 'strong'
    if strong and a.batch % world:
        raise SystemExit(f'bench.py --scaling strong: the global batch {a.batch} does not divide into {world} equal shards')
    B = a.batch // world if strong else a.batch      # poses on THIS GPU
    UB = a.batch                                      # the batch an "iteration" of `value` refers to: global (strong) / per GPU (weak)
    agg = 1 if strong else world                      # `value` = agg x this job's iterations per second
    use_disc = a.config in (3, 5)
    use_sil = a.config == 5
    model_np = sm.synthetic_smpl(1234)
    J_np = sm.default_h36m_regressor()
    batch_np = sm.synthetic_batch(model_np, J_np, B, seed=1000 + rank)
    dmodel = eng_mod.DeviceModel(model_np, dev)
    jd = importlib.import_module(PKG + '.dist')
    flags = eng_mod.FLAG_KEEP_VERTS | (eng_mod.FLAG_POSE_DISC if use_disc else 0) | (eng_mod.FLAG_SILHOUETTE if use_sil else 0)
    eng = eng_mod.RefineEngine(dmodel, B, batch_norm=B * world, flags=flags)
    J = torch.from_numpy(J_np).to(dev).contiguous()
    eng.set_j_regressor(J)
    disc_flat, disc_sd = default_disc_flat(0)
    if use_disc:
        eng.set_pose_disc(disc_flat.to(dev))

    x6d = torch.from_numpy(batch_np['pose6d']).to(dev).contiguous()
    betas = torch.from_numpy(batch_np['betas']).to(dev).contiguous()
    gt = torch.from_numpy(batch_np['gt_j3d'])
    gt_c = (gt - gt[:, :1]).to(dev).contiguous()
    m = torch.zeros(B, 154, device=dev)
    v = torch.zeros(B, 154, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    sq = torch.zeros(B, device=dev)

    def silhouette_setup(e, xs, bs):
        """BASELINE configs[4]: silhouette target = the initial mesh through a perturbed camera, binarised"""
        cam = torch.from_numpy(batch_np['cam']).to(dev).contiguous()
        cam_m, cam_v = torch.zeros_like(cam), torch.zeros_like(cam)
        _, verts0 = e.find_joints_forward(bs, x6d=xs, return_verts=True)
        shift = torch.tensor([0.15, -0.1, 1.0], device=dev)
        sil_mask = (e.silhouette_forward(verts0, (cam + shift).contiguous()) > 0).float().contiguous()
        del verts0
        e.set_silhouette(sil_mask, cam, cam_m, cam_v)
        return sil_mask, cam, cam_m, cam_v

    if use_sil:
        sil_refs = silhouette_setup(eng, x6d, betas)   # noqa: F841  (the engine keeps raw pointers)
    Jm, Jv = torch.zeros_like(J), torch.zeros_like(J)
    Jstep = torch.zeros(1, dtype=torch.int32, device=dev)
    dJ = torch.zeros_like(J)          # the all-reduce bucket of the J step: allocated once
    after_j = [False]                 # the previous engine call was a J step on (x6d, betas): the next iteration reuses its forward

    # The J step between two segments of the loop when the host drives it (N > 1): local gradient -> ONE all-reduce -> replicated
    # Adam + re-normalisation.  The payload is the regressor's support (17 x 128 floats = 8 704 B) when it fits the engine's lists,
    # the dense (17,6890) gradient otherwise (dist.JStepExchange; asked once, here).
    xch = jd.JStepExchange(eng, dJ, compact=True, reduce=(lambda t: dist.all_reduce(t)) if dist is not None else (lambda t: t))

    def j_step():
        """scripts/optimize.py:300-312 data-parallel; no allocation, nothing read back"""
        xch.step(J, Jm, Jv, Jstep, 1e-2, x6d, betas, gt_c)
        after_j[0] = not use_sil

    it_count = [0]     # inner iterations run so far in the current measurement: the J step follows every cadence-th one, counted
                       # ACROSS timed regions (a region of --steps 20 at the reference cadence of 100 holds a J step every 5th time)

    def run_host_driven(n, cadence, reuse=True):
        """what N > 1 ranks execute: the host issues refine_run / j_regressor_grad / all_reduce / j_step_apply per J step"""
        left, nj = n, 0
        while left > 0:
            seg = min(left, cadence - it_count[0] % cadence)
            eng.refine_run(x6d, betas, gt_c, m, v, step, 1e-2, seg, sqerr=sq, after_j_step=after_j[0] and reuse)
            after_j[0] = False
            left -= seg
            it_count[0] += seg
            if it_count[0] % cadence == 0:
                j_step()
                nj += 1
        return nj

    def run(n, cadence, reuse=False):
        """n inner iterations with a J step after every `cadence`-th one (counted across calls: it_count).
        One process: ONE C call per aligned stretch (jrr_refine_run_j_steps).  N > 1: the all-reduce sits between the two
        halves of each J step, so the host drives the segments (run_host_driven).
        reuse: the iteration after a J step re-regresses its joints from the J step's stored vertices instead of repeating the
        SMPL forward.  The reference draws a NEW batch after every J step (scripts/optimize.py:144-148, 300-312), so the headline
        runs with reuse=False; only `cadence1` (a J step after every iteration on the same poses) uses it."""
        if dist is not None:
            return run_host_driven(n, cadence, reuse)
        left, nj = n, 0

        def with_j(k, every):
            eng.refine_run_j_steps(x6d, betas, gt_c, m, v, step, 1e-2, k, every, J, Jm, Jv, Jstep, 1e-2, sqerr=sq,
                                   after_j_step=after_j[0] and reuse, reuse_forward=reuse)
            after_j[0] = (not use_sil) and reuse
            it_count[0] += k
        off = it_count[0] % cadence
        if off and left >= cadence - off:          # finish the cadence a previous region left open: its J step closes this stretch
            with_j(cadence - off, cadence - off)
            left -= cadence - off; nj += 1
        if it_count[0] % cadence == 0 and left >= cadence:
            k = left // cadence
            with_j(k * cadence, cadence)
            left -= k * cadence; nj += k
        if left:
            eng.refine_run(x6d, betas, gt_c, m, v, step, 1e-2, left, sqerr=sq, after_j_step=after_j[0] and reuse)
            after_j[0] = False
            it_count[0] += left
        return nj

    def barrier():
        if dist is not None:
            dist.barrier()

    own_times = []      # this rank's OWN time to finish each timed region (before the closing barrier): the per-rank spread of the line

    def timed_region(n, cadence, fn=None):
        """EXACTLY n steps between barrier + synchronize pairs; max over ranks"""
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        nj = (fn or run)(n, cadence)
        torch.cuda.synchronize()
        own_times.append(time.perf_counter() - t0)
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, nj

    def repeats_for(el, lo=1):
        reps = int(min(a.max_repeats, max(lo, np.ceil(a.min_timed_ms * 1e-3 / max(el, 1e-6)))))
        if dist is not None:           # every rank must run the same number of regions
            t = torch.tensor([reps], device=dev, dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            reps = int(t.item())
        return reps

    cadence = max(1, a.j_step_every)   # the reference's: one J step (SMPL fwd, dJ, RCCL all-reduce, Adam) per 100 inner iterations
    run(a.warmup, cadence)
    j_step()                       # untimed: the first J step zero-fills the padded vertex buffer
    it_count[0] = 0
    timed_region(max(a.steps, cadence), cadence)  # untimed: loads the code objects of the J step path (holds >= 1 J step)
    it_count[0] = 0
    regions, region_nj = [], []
    del own_times[:]
    el, nj0 = timed_region(a.steps, cadence)
    regions.append(el); region_nj.append(nj0)
    # whole cadences: the timed regions together hold the reference's share of J steps exactly (1 per `cadence` iterations)
    reps = repeats_for(el)
    per_cad = max(1, cadence // np.gcd(cadence, a.steps))          # regions per whole number of cadences
    reps = min(a.max_repeats, (reps + per_cad - 1) // per_cad * per_cad)
    for _ in range(reps - 1):
        el, njr = timed_region(a.steps, cadence)
        regions.append(el); region_nj.append(njr)
    # `value`: every timed step over every timed second (the J steps amortised at the reference cadence)
    elapsed = sum(regions) / len(regions)
    nj_region = sum(region_nj)
    own_ms = sum(own_times) / len(own_times) / a.steps * 1e3     # this rank alone, mean over the headline's regions
    per_rank_ms = [own_ms]
    if dist is not None:
        per_rank_ms = [None] * world
        dist.all_gather_object(per_rank_ms, own_ms)
    loss_joint = float(sq.sum().item()) / (B * 51)

    # ---- BASELINE configs[3] "all-reduce on the J_regressor gradient EACH step": J step after every iteration,
    #      timed like `value` (median of >= 5 regions) ----
    run_c1 = lambda n, c: run(n, c, reuse=True)      # noqa: E731
    it_count[0] = 0
    timed_region(a.steps, 1, run_c1)       # untimed warm-up region
    c1_regions = [timed_region(a.steps, 1, run_c1)[0]]
    for _ in range(max(4, min(repeats_for(c1_regions[0]), 10) - 1)):
        c1_regions.append(timed_region(a.steps, 1, run_c1)[0])
    c1_el = statistics.median(c1_regions)
    # ---- N > 1: what the collective itself costs.  (a) stand-alone all-reduces of the three payloads of this path (median of 50 after
    #      a warm-up; barrier + synchronize around each one, max over ranks is what the slowest rank sees); (b) the cadence-1 regions
    #      again with a no-op in place of the all-reduce (each rank steps J with its local gradient: timing only, J restored after) ----
    collective_cost = None
    if dist is not None:
        def ar_time(nfloats):
            buf = torch.zeros(nfloats, device=dev)
            for _ in range(5):
                dist.all_reduce(buf)
            ts = []
            for _ in range(50):
                torch.cuda.synchronize(); dist.barrier()
                t0 = time.perf_counter()
                dist.all_reduce(buf)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            t = torch.tensor([statistics.median(ts)], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            # ... and 20 back to back in the stream (what a loop that does not synchronise per step pays)
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            for _ in range(20):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            t2 = torch.tensor([(time.perf_counter() - t0) / 20], device=dev, dtype=torch.float64)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            return {'bytes': nfloats * 4, 'median_us_synchronised': round(float(t.item()) * 1e6, 1), 'us_back_to_back': round(float(t2.item()) * 1e6, 1)}
        opt_mod = importlib.import_module(PKG + '.optimize')
        n_bucket = opt_mod.SharedBucket(dev, True, False, 10).flat.numel()
        collective_cost = {'standalone_allreduce': {'j_step_support_17x128': ar_time(17 * 128), 'j_step_dense_17x6890': ar_time(17 * 6890),
                                                    'outer_step_flat_bucket': ar_time(n_bucket)}}
        keepJ = [t.clone() for t in (J, Jm, Jv, Jstep)]
        keep_reduce = xch.reduce
        xch.reduce = lambda t: t
        try:
            it_count[0] = 0
            timed_region(a.steps, 1, run_c1)
            noop_regions = [timed_region(a.steps, 1, run_c1)[0] for _ in range(5)]
        finally:
            xch.reduce = keep_reduce
            for dst, src in zip((J, Jm, Jv, Jstep), keepJ):
                dst.copy_(src)
            eng.set_j_regressor(J)
            eng.j_support_info()
            after_j[0] = False
        noop_ms = statistics.median(noop_regions) / a.steps * 1e3
        collective_cost.update({'cadence1_ms_per_step': round(c1_el / a.steps * 1e3, 4), 'cadence1_noop_collective_ms_per_step': round(noop_ms, 4),
                                'collective_us_per_step': round((c1_el / a.steps * 1e3 - noop_ms) * 1e3, 1),
                                'note': 'cadence-1 iteration (J step + all-reduce after every iteration) minus its twin with a no-op in place of '
                                        'the all-reduce = what the collective costs in the loop, launch + wait for the slowest rank included'})
    # ---- the same cadence through the call sequence N > 1 ranks execute (host-driven segments, the support-sized payload, a
    #      no-op in place of the collective at world size 1): bounds the multi-GPU cadence-1 cost before any 8-GPU box sees it ----
    c1h_el, c1h_regions, c1r = None, [], None
    rccl_hung = [False]
    if dist is None:
        run_c1h = lambda n, c: run_host_driven(n, c, reuse=True)      # noqa: E731
        it_count[0] = 0
        timed_region(a.steps, 1, run_c1h)
        c1h_regions = [timed_region(a.steps, 1, run_c1h)[0] for _ in range(5)]
        c1h_el = statistics.median(c1h_regions)
        after_j[0] = False

    def rccl_one_rank_block():
        """cadence-1 host-driven regions with the collective EXECUTED: a one-rank RCCL group on this GPU (a sum over one rank is the
        identity; what is timed is RCCL's own enqueue + kernel on the 8 704-byte device buffer between the two halves of every J step).
        Runs LAST, after every other measurement and the CPU baseline: if the bring-up hangs, nothing else shares the GPU with a
        half-initialised communicator, the line is printed with the error and the process exits non-zero (status 3)."""
        c1r = None
        if dist is None and not a.no_rccl_one_rank and a.backend == 'nccl':
            # (RCCL prints a version banner through C stdio on stdout: this program's stdout is ONE JSON line, so fd 1 points at
            # stderr while the group lives, and C stdio is flushed before it is restored)
            import ctypes
            sys.stdout.flush()
            fd1 = os.dup(1)
            os.dup2(2, 1)
            try:
                import socket
                import threading
                import torch.distributed as tdist
                with socket.socket() as so:
                    so.bind(('127.0.0.1', 0))
                    port = so.getsockname()[1]
                up = {}

                def bring_up():      # the part that talks to the outside (bootstrap sockets): under a watchdog, so that a box whose
                    try:             # RCCL cannot start costs this block, not the bench line
                        torch.cuda.set_device(dev)
                        tdist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=dev)
                        t0_ = torch.zeros(8, device=dev)
                        tdist.all_reduce(t0_)
                        torch.cuda.synchronize()
                        up['ok'] = True
                    except Exception as ex_:      # noqa: BLE001
                        up['err'] = repr(ex_)[:200]
                th = threading.Thread(target=bring_up, daemon=True)
                th.start()
                th.join(90.0)
                if th.is_alive():
                    rccl_hung[0] = True
                    c1r = {'ms_per_step': None, 'error': 'the one-rank RCCL group did not come up within 90 s'}
                elif 'err' in up:
                    c1r = {'ms_per_step': None, 'error': up['err']}
                else:
                    keep = xch.reduce
                    xch.reduce = lambda t: tdist.all_reduce(t)
                    try:
                        it_count[0] = 0
                        timed_region(a.steps, 1, run_c1h)
                        c1r_regions = [timed_region(a.steps, 1, run_c1h)[0] for _ in range(5)]
                    finally:
                        xch.reduce = keep
                    torch.cuda.synchronize()
                    tdist.destroy_process_group()
                    c1r = {'ms_per_step': round(statistics.median(c1r_regions) / a.steps * 1e3, 4),
                           'repeat_ms_per_step': [round(r / a.steps * 1e3, 4) for r in c1r_regions], 'backend': 'nccl (RCCL), one rank'}
            except Exception as ex:      # the bench line survives a box whose RCCL cannot start
                c1r = {'ms_per_step': None, 'error': repr(ex)[:200]}
            finally:
                try:
                    ctypes.CDLL(None).fflush(None)
                except Exception:
                    pass
                os.dup2(fd1, 1)
                os.close(fd1)
            after_j[0] = False

        return c1r

    # ---- the inner iteration ALONE (no J step in the region, no forward reuse): the denominator of roofline.whole_step ----
    def run_inner(n, _cadence):
        eng.refine_run(x6d, betas, gt_c, m, v, step, 1e-2, n, sqerr=sq)
        after_j[0] = False
        return 0
    in_regions = [timed_region(a.steps, 0, run_inner)[0] for _ in range(3)]
    inner_ms = statistics.median(in_regions) / a.steps * 1e3

    # ---- per-kernel timing (HIP events on the launch stream) over the same number of steps ----
    eng.set_profiling(True)
    eng.refine_run(x6d, betas, gt_c, m, v, step, 1e-2, max(2, min(a.steps, 20)), sqerr=sq)
    prof = eng.profile_read()
    probe = eng.probe_read()
    eng.set_profiling(False)

    # ---- J step alone ----
    j_step()
    torch.cuda.synchronize(); barrier()
    tj = time.perf_counter()
    nj = 10
    for _ in range(nj):
        j_step()
    torch.cuda.synchronize(); barrier()
    j_ms = (time.perf_counter() - tj) / nj * 1e3
    after_j[0] = False

    # ---- pose-discriminator update (scripts/optimize.py:276-284; one per outer batch in the reference): two D
    #      forward + weight-gradient passes, all-reduce of the 1.84 M-float gradient, Adam(lr 1e-3), re-upload.
    #      Timed separately, never part of `value`. ----
    d_ms = None
    if use_disc:
        dflat = disc_flat.to(dev).clone()
        dm_, dv_ = torch.zeros_like(dflat), torch.zeros_like(dflat)
        dstep = torch.zeros(1, dtype=torch.int32, device=dev)
        spin_pose = torch.from_numpy(batch_np['pose6d']).to(dev).contiguous()
        g = torch.zeros_like(dflat)

        def d_step():
            g.zero_()
            eng.pose_disc_backward_params(x6d, 0.0, g)
            eng.pose_disc_backward_params(spin_pose, 1.0, g)
            if dist is not None:
                dist.all_reduce(g)
            dstep.add_(1)
            eng_mod.adam_step(dflat, g, dm_, dv_, dstep, 1e-3)
            eng.set_pose_disc(dflat)

        d_step()
        torch.cuda.synchronize(); barrier()
        td = time.perf_counter()
        for _ in range(nj):
            d_step()
        torch.cuda.synchronize(); barrier()
        d_ms = (time.perf_counter() - td) / nj * 1e3
        eng.set_pose_disc(disc_flat.to(dev))

    def side_run(flags_, setup=None, model=None, Bs=None, disc=None, tiles=False, force_profile=False):
        """a separately reported mode on a fresh copy of the same batch (its first Bs poses): warm-up, then --steps timed
        iterations (median of 3)"""
        Bs = Bs or B
        disc = use_disc if disc is None else disc
        e2 = eng_mod.RefineEngine(model or dmodel, Bs, batch_norm=Bs * world, flags=flags_ | (eng_mod.FLAG_SUPPORT_TILES if tiles else 0))
        if flags_ & eng_mod.FLAG_FOLDED:
            e2.set_folded(True)
        e2.set_j_regressor(J)
        if tiles:
            e2.j_support_info()      # engages FLAG_SUPPORT_TILES when the support fits the device lists
        if disc:
            e2.set_pose_disc(disc_flat.to(dev))
        fx = torch.from_numpy(batch_np['pose6d'][:Bs]).to(dev).contiguous()
        fb = torch.from_numpy(batch_np['betas'][:Bs]).to(dev).contiguous()
        fgt = gt_c[:Bs].contiguous()
        refs = setup(e2, fx, fb) if setup else None      # noqa: F841
        fm, fv = torch.zeros(Bs, 154, device=dev), torch.zeros(Bs, 154, device=dev)
        fstep = torch.zeros(1, dtype=torch.int32, device=dev)
        e2.refine_run(fx, fb, fgt, fm, fv, fstep, 1e-2, a.warmup)

        def go(n, _c):
            e2.refine_run(fx, fb, fgt, fm, fv, fstep, 1e-2, n)
            return 0
        fel = statistics.median([timed_region(a.steps, 0, go)[0] for _ in range(3)])
        out_ = {'value': round(a.steps / fel * agg, 3), 'unit': f'it/s (x{Bs * (world if strong else 1)} poses)', 'ms_per_step': round(fel / a.steps * 1e3, 4),
                'joint_sparse': int(e2.info.get('joint_sparse') or 0)}
        if tiles:
            on, nt = e2.support_tiles()
            out_['vertex_tiles'] = {'run': nt, 'of': 216, 'restricted_to_regressor_support': on}
            sv_on, n_sv = e2.support_vertices()
            out_['support_vertices'] = {'per_vertex_iteration': sv_on, 'vertices': n_sv,
                                        'launches_per_iteration': (5 if disc else 1) if sv_on else None}
            # ... and with a J step (+ forward reuse) after EVERY iteration, one C call (BASELINE configs[3]'s pattern on this engine)
            Jc = J.clone()
            Jm2, Jv2, Js2 = torch.zeros_like(Jc), torch.zeros_like(Jc), torch.zeros(1, dtype=torch.int32, device=dev)
            e2.set_j_regressor(Jc)
            e2.j_support_info()

            def go1(n, _c):
                e2.refine_run_j_steps(fx, fb, fgt, fm, fv, fstep, 1e-2, n, 1, Jc, Jm2, Jv2, Js2, 1e-2, reuse_forward=True)
                return n
            timed_region(a.steps, 0, go1)
            c1t = statistics.median([timed_region(a.steps, 0, go1)[0] for _ in range(3)])
            out_['cadence1_ms_per_step'] = round(c1t / a.steps * 1e3, 4)
            e2.set_j_regressor(J)
            e2.j_support_info()
        if setup is silhouette_setup or Bs != B or tiles or force_profile:
            e2.set_profiling(True)
            e2.refine_run(fx, fb, fgt, fm, fv, fstep, 1e-2, max(2, min(a.steps, 10)))
            pr = e2.profile_read()
            e2.set_profiling(False)
            out_['kernels_ms'] = {k: round(t, 4) for k, (t, n) in pr.items() if n}
            if tiles and out_['support_vertices']['per_vertex_iteration']:
                # the composed kernel is timed under the forward kernel's class; the chain-forward class holds the per-joint MLP forward
                # that precedes the FIRST iteration of a call only
                km_ = out_['kernels_ms']
                if 'k_lbs_fwd' in km_:
                    km_['k_sup_step'] = km_.pop('k_lbs_fwd')
                if 'k_prep_fwd' in km_:
                    km_['k_dconv_fwd_first_iteration_only'] = km_.pop('k_prep_fwd')
        return out_

    # ---- the separately reported blocks (one function each, above main()): never part of `value` ----
    import types
    c = types.SimpleNamespace(a=a, dev=dev, world=world, dist=dist, B=B, use_disc=use_disc, use_sil=use_sil, eng_mod=eng_mod, sm=sm, dmodel=dmodel,
                              model_np=model_np, J_np=J_np, batch_np=batch_np, side_run=side_run, silhouette_setup=silhouette_setup, prof=prof,
                              inner_ms=inner_ms, d_ms=d_ms, j_ms=j_ms, strong=strong, agg=agg)
    folded = block_folded(c)
    bf16x3 = block_bf16x3(c)
    support_tiles = block_support_tiles(c)
    config5 = block_config5(c)
    config2 = block_config2(c)
    skin_variants = block_skin_variants(c)
    driver_outer, ref_default = block_driver(c, support_tiles)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    import types as _types
    out = build_line(_types.SimpleNamespace(**{k_: v_ for k_, v_ in locals().items() if not k_.startswith('_')}))
    if not a.no_cpu_baseline and world == 1:   # the CPU baseline leg runs at N = 1 only
        cb = min(a.cpu_batch or B, B)
        variants, nthreads = cpu_baseline(model_np, J_np, batch_np, disc_sd, cb, a.cpu_seconds, use_disc, B)
        sweep_info = variants.pop('thread_sweep', None)
        for vv in variants.values():
            if 'it_s_at_sample_batch' in vv:
                vv['value_batch4096_units'] = round(vv['it_s_at_sample_batch'] * vv['batch'] / B, 5)
        out['cpu_baseline'] = {'value': variants['one_eval']['value_batch4096_units'], 'unit': f'it/s (x{B} poses)', 'cores': nthreads,
                               'kind': 'port', 'cpu_model': cpu_model_name(), 'host_threads': os.cpu_count(), 'thread_sweep': sweep_info,
                               'sample': f"{variants['one_eval']['iterations']} inner iterations at batch {cb} of the same workload "
                                         f'(oracle/reference_port.py: torch-CPU ops in the reference order, autograd, torch.optim.Adam; '
                                         f"1 SMPL eval/iter), {variants['one_eval']['seconds']} s on {nthreads} threads picked from a "
                                         f'1-iteration sweep up to all physical cores (`thread_sweep`), scaled x{cb}/{B} to batch-{B} units; `full_batch` = one '
                                         f'iteration at batch {B} itself, `config1_forward_b4` = BASELINE configs[0]',
                               'variants': variants}
    if c1h_el is not None:       # LAST: the one measurement that brings up a communicator (see rccl_one_rank_block)
        out['cadence1_host_driven']['with_rccl_one_rank_allreduce'] = rccl_one_rank_block()
    print(json.dumps(out))
    if rccl_hung[0]:      # a bootstrap thread is still stuck inside RCCL: the line is out; leave without waiting for it, and say so
        sys.stdout.flush()
        os._exit(3)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
