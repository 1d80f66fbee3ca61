"""Entry point with the reference's shape (/root/reference/main.py:1-29):
    python main.py [flags]            (flags: scripts/args.py names + this build's additions)
    torchrun --nproc-per-node N main.py [flags]   (data parallel, one process per GPU over RCCL)
Runs set_seed(0), optimize_pose_refiner() and then the evaluation report test_pose_refiner_model()
(/root/reference/main.py:21-25).  The reference's two further evaluations (test_pose_refiner_model_VIBE_MEVA,
main.py:26-27) need the external VIBE / MEVA checkouts and are out of scope."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = 'joint-regressor-refinement_amd'
pkg = importlib.import_module(PKG)
args = importlib.import_module(PKG + '.args').args
optimize = importlib.import_module(PKG + '.optimize')
evaluation = importlib.import_module(PKG + '.test')
utils = importlib.import_module(PKG + '.utils')

if __name__ == '__main__':
    if args.wandb_log:
        try:
            import wandb
            wandb.init(project='human_body_pose_optimization', name=args.name)
        except ImportError:
            print('wandb is not installed; logging to stdout')
    utils.set_seed(0)
    res = optimize.optimize_pose_refiner()                                   # main.py:23
    if not args.skip_eval and int(os.environ.get('RANK', '0')) == 0:
        # main.py:25.  The reference reads models/retrained_J_Regressor.pt; when this run did not write a checkpoint
        # (--save_j_regressor unset) and that file is absent, the regressor just trained is evaluated through a
        # temporary checkpoint in the same format.
        path = args.eval_j_regressor or args.save_j_regressor
        if path is None and not os.path.exists('models/retrained_J_Regressor.pt'):
            import tempfile
            path = os.path.join(tempfile.mkdtemp(prefix='jrr_'), 'retrained_J_Regressor.pt')
            importlib.import_module(PKG + '.checkpoint').save_j_regressor(res['J_regressor'], path)
        evaluation.test_pose_refiner_model(path)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
