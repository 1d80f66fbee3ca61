"""Entry point with the reference's shape (/root/reference/main.py:1-29):
    python main.py [flags]            (flags: scripts/args.py names + this build's additions)
    torchrun --nproc-per-node N main.py [flags]   (data parallel, one process per GPU)
Runs set_seed(0) then optimize_pose_refiner(); the reference's evaluation scripts that follow
(test_pose_refiner_model*, scripts/test.py) need Human3.6M / VIBE / MEVA and are out of scope."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('joint-regressor-refinement_amd')
args = importlib.import_module('joint-regressor-refinement_amd.args').args
optimize = importlib.import_module('joint-regressor-refinement_amd.optimize')
utils = importlib.import_module('joint-regressor-refinement_amd.utils')

if __name__ == '__main__':
    if args.wandb_log:
        try:
            import wandb
            wandb.init(project='human_body_pose_optimization', name=args.name)
        except ImportError:
            print('wandb is not installed; logging to stdout')
    utils.set_seed(0)
    optimize.optimize_pose_refiner()
