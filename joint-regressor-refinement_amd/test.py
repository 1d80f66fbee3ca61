"""Evaluation report `test_pose_refiner_model()` restated from the reference
(/root/reference/scripts/test.py:33-138, called at /root/reference/main.py:25) on the HIP path.

Per validation batch, under no_grad (reference line numbers):
  :46-53   J_regressor = torch.load('models/retrained_J_Regressor.pt'); J_regressor_initial = J_regressor_h36m.npy;
           mask from the initial regressor
  :59-63   DataLoader(data_set("validation"), batch_size=args.batch_size, shuffle=True, drop_last=True)
  :89-105  SPIN prediction -> rot6d_to_rotmat -> (B,24,3,3)       (here: the batch's own 6-D pose, see optimize.py)
  :107-123 find_joints with the INITIAL regressor -> evaluate -> "before";  with the RETRAINED one -> "after"
  :125-138 print the means of the per-batch MPJPE / PA-MPJPE, 4 decimals, in mm
The SMPL forward runs ONCE per batch for both regressors' joints where the reference runs it twice (the vertices do
not depend on the regressor); `evaluate` is the on-device Procrustes kernel (k_evaluate).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import checkpoint, engine as _engine, smpl_model, utils
from .args import args
from .smpl import SMPL


def test_pose_refiner_model(retrained_path: Optional[str] = None, log=print) -> Dict[str, float]:
    device = torch.device(args.device)
    torch.cuda.set_device(device)
    smpl = SMPL(args.smpl_dir, batch_size=1, allow_synthetic=args.synthetic or args.smpl_dir == 'SPIN/data/smpl').to(device)   # :40-43
    path = retrained_path or args.eval_j_regressor or args.save_j_regressor or 'models/retrained_J_Regressor.pt'
    J_regressor = checkpoint.load_j_regressor(path).to(device)                                          # :46-47
    J_np = smpl_model.default_h36m_regressor(args.j_regressor_init,
                                             allow_default=args.synthetic or args.j_regressor_init == 'SPIN/data/J_regressor_h36m.npy')
    J_regressor_initial = torch.from_numpy(J_np).float().to(device)                                      # :48-49
    j_reg_mask = utils.find_j_reg_mask(J_regressor_initial)                                              # :51-53

    from . import optimize as _opt
    if args.data_root:
        source = _opt._dataset_batches(args.data_root, args.batch_size, args.seed, device, drop_last=True)       # :59-63
    else:        # synthetic validation batches: seeds disjoint from the optimiser's
        source = _opt._synthetic_batches(smpl.model_np, J_np, args.batch_size, args.synthetic_batches, args.seed + 7919)

    mpjpe_before, pampjpe_before, mpjpe_after, pampjpe_after = [], [], [], []
    engines: Dict[int, _engine.RefineEngine] = {}
    with torch.no_grad():
        for batch in source:
            B = int(batch['pose6d'].shape[0])
            if B not in engines:
                engines[B] = _engine.RefineEngine(smpl.device_model, B)
            eng = engines[B]
            x6d = batch['pose6d'].to(device).float().contiguous()
            betas = batch['betas'].to(device).float().contiguous()
            gt = utils.move_pelvis(batch['gt_j3d'].to(device).float())                                   # :87
            eng.set_j_regressor(J_regressor_initial, j_reg_mask)
            joints = eng.find_joints_forward(betas, x6d=x6d)                                             # :107-108
            mb, pb = utils.evaluate(joints, gt)                                                         # :110-111
            eng.set_j_regressor(J_regressor, j_reg_mask)
            joints = eng.find_joints_forward(betas, x6d=x6d)                                             # :116-117
            ma, pa = utils.evaluate(joints, gt)                                                         # :119-120
            mpjpe_before.append(mb); pampjpe_before.append(pb); mpjpe_after.append(ma); pampjpe_after.append(pa)
    if not mpjpe_before:
        raise RuntimeError('no validation batch (drop_last=True needs at least --batch_size samples)')
    mean = lambda xs: float(torch.tensor(xs, dtype=torch.float64).mean())
    rep = {'mpjpe_before': mean(mpjpe_before), 'pampjpe_before': mean(pampjpe_before), 'mpjpe_after': mean(mpjpe_after),
           'pampjpe_after': mean(pampjpe_after), 'batches': len(mpjpe_before), 'retrained_j_regressor': path,
           'body_model': smpl.provenance}
    log('MPJPE')                                                                                         # :125-138
    log(f"{rep['mpjpe_before']:.4f}")
    log('PAMPJPE')
    log(f"{rep['pampjpe_before']:.4f}")
    log('')
    log('after')
    log('MPJPE')
    log(f"{rep['mpjpe_after']:.4f}")
    log('PAMPJPE')
    log(f"{rep['pampjpe_after']:.4f}")
    return rep


test_pose_refiner_model.__test__ = False      # not a pytest test: the reference's function name
