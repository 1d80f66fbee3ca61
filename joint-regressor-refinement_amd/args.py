"""Flag system of the reference (/root/reference/scripts/args.py:1-103) for this path.

The reference evaluates `parser.parse_args()` at import time and exposes a module-level singleton
`args`.  The same 15 flags are kept with identical names, types and defaults; flags added by this
build (synthetic data, SMPL directory, J-step cadence, checkpoint output) never rename existing
ones.  `args` is parsed lazily from sys.argv on first attribute access (unknown flags are
tolerated so the module can be imported under pytest / torchrun).
"""
from __future__ import annotations

import argparse
import sys


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser()
    # ---- the reference's flags (scripts/args.py:5-21), unchanged ----
    parser.add_argument('--name', type=str)
    parser.add_argument('--train_epochs', type=int, default=1)
    parser.add_argument('--opt_steps', type=int, default=40)
    parser.add_argument('--batch_size', type=int, default=256)
    parser.add_argument('--optimization_batch_size', type=int, default=1)
    parser.add_argument('--learning_rate', type=float, default=1e-4)
    parser.add_argument('--opt_lr', type=float, default=1e-2)
    parser.add_argument('--disc_learning_rate', type=float, default=1e-4)
    parser.add_argument('--opt_disc_learning_rate', type=float, default=1e-3)
    parser.add_argument('--translation_lr', type=float, default=1e-6)
    parser.add_argument('--j_reg_lr', type=float, default=1e-2)
    parser.add_argument('--optimization_rate', type=float, default=1e4)
    parser.add_argument('--wandb_log', action='store_true')
    parser.add_argument('--compute_canada', action='store_true')
    parser.add_argument('--device', type=str, default='cuda:0')
    # ---- additions of this build ----
    parser.add_argument('--smpl_dir', type=str, default='SPIN/data/smpl',
                        help='directory with SMPL_NEUTRAL.{pkl,npz} (optimize.py:96-99); synthetic model if absent')
    parser.add_argument('--j_regressor_init', type=str, default='SPIN/data/J_regressor_h36m.npy',
                        help='initial H36M regressor (optimize.py:105-107); shipped checkpoint support if absent')
    parser.add_argument('--synthetic_batches', type=int, default=2,
                        help='number of synthetic outer batches (the Human3.6M loader of scripts/data.py is out of scope)')
    parser.add_argument('--inner_iters', type=int, default=100, help='pose-refinement iterations per batch (optimize.py:220)')
    parser.add_argument('--j_step_every', type=int, default=100,
                        help='inner iterations per J_regressor step (100 = reference cadence: once per outer batch)')
    parser.add_argument('--j_allreduce', type=str, default='support', choices=['support', 'dense'],
                        help='payload of the in-loop J step all-reduce under data parallelism: the regressor\'s support (8 704 B) or the dense (17,6890) gradient')
    parser.add_argument('--all_vertex_tiles', action='store_true',
                        help='run every joint-loss iteration on all 216 vertex tiles (default: only the tiles that hold an entry of the J_regressor\'s support; same results up to summation order)')
    parser.add_argument('--no_pose_disc', action='store_true', help='drop the pose-discriminator term (BASELINE config 2)')
    parser.add_argument('--shape_disc', action='store_true', help='add the shape-discriminator term (optimize.py:244,249-250)')
    parser.add_argument('--reprojection', action='store_true',
                        help='camera pre-fit (optimize.py:187-199) + 2-D joint term (optimize.py:231-233) on synthetic gt_j2d')
    parser.add_argument('--silhouette', action='store_true',
                        help='soft-silhouette term (optimize.py:234-237, x100) against synthetic masks (BASELINE configs[4])')
    parser.add_argument('--camera_iters', type=int, default=1000, help='camera pre-fit Adam steps (optimize.py:190)')
    parser.add_argument('--save_j_regressor', type=str, default=None,
                        help='write the trained regressor in the models/retrained_J_Regressor.pt format')
    parser.add_argument('--seed', type=int, default=0)
    parser.add_argument('--eval_j_regressor', type=str, default=None,
                        help='retrained regressor read by the evaluation report (default: --save_j_regressor, else '
                             'models/retrained_J_Regressor.pt as scripts/test.py:46-47)')
    parser.add_argument('--skip_eval', action='store_true', help='main.py: do not run test_pose_refiner_model() after the optimiser')
    parser.add_argument('--dist_backend', type=str, default=None,
                        help='torch.distributed backend under torchrun (default: nccl = RCCL on GPUs); gloo for debugging')
    parser.add_argument('--single_device', action='store_true',
                        help='debug: every rank uses --device (exercises the N > 1 path on a 1-GPU box; use with --dist_backend gloo)')
    parser.add_argument('--data_root', type=str, default=None,
                        help='directory holding precomputed_{train,val}/ in the reference layout (scripts/data.py:49-86); '
                             'synthetic batches if not given')
    parser.add_argument('--synthetic', action='store_true',
                        help='accept the synthetic body model / regressor when --smpl_dir / --j_regressor_init do not exist '
                             '(otherwise a missing explicit path is an error)')
    return parser


REFERENCE_FLAGS = {
    'name': None, 'train_epochs': 1, 'opt_steps': 40, 'batch_size': 256, 'optimization_batch_size': 1,
    'learning_rate': 1e-4, 'opt_lr': 1e-2, 'disc_learning_rate': 1e-4, 'opt_disc_learning_rate': 1e-3,
    'translation_lr': 1e-6, 'j_reg_lr': 1e-2, 'optimization_rate': 1e4, 'wandb_log': False, 'compute_canada': False,
    'device': 'cuda:0'}


def get_args(argv=None) -> argparse.Namespace:
    ns, _ = build_parser().parse_known_args(sys.argv[1:] if argv is None else argv)
    return ns


class _LazyArgs:
    """Module-level singleton with the reference's `from scripts.args import args` ergonomics."""
    _ns = None

    def _get(self):
        if _LazyArgs._ns is None:
            _LazyArgs._ns = get_args()
        return _LazyArgs._ns

    def __getattr__(self, k):
        return getattr(self._get(), k)

    def __setattr__(self, k, v):
        setattr(self._get(), k, v)

    def __repr__(self):
        return repr(self._get())


args = _LazyArgs()
