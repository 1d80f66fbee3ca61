#!/usr/bin/env python3
"""K7: pin the SMPL LBS restatement on a machine that HAS the licence-gated SMPL model
and `smplx==0.1.26` (neither is available in the build image, which is why LBS parity is "unpinned").

    python tools/dump_smplx_vertices.py /path/to/smpl_models out.npz      # on the machine with smplx
    python tools/dump_smplx_vertices.py --check out.npz /path/to/smpl_models   # anywhere: oracle (and HIP if a GPU
                                                                                # is present) vs the dumped vertices
The dump holds 4 seeded poses (rotation matrices), betas, and smplx's vertices / joints."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dump(model_dir, out):
    import smplx
    smpl = smplx.SMPL(model_dir, batch_size=4)
    g = torch.Generator().manual_seed(0)
    import oracle
    R = oracle.rodrigues(torch.randn(4 * 24, 3, generator=g) * 0.4).view(4, 24, 3, 3)
    betas = torch.randn(4, 10, generator=g)
    o = smpl(global_orient=R[:, :1], body_pose=R[:, 1:], betas=betas, pose2rot=False)
    np.savez_compressed(out, R=R.numpy(), betas=betas.numpy(), vertices=o.vertices.detach().numpy(),
                        joints=o.joints.detach().numpy()[:, :24])
    print('wrote', out)


def check(npz, model_dir):
    import oracle
    d = np.load(npz)
    sm = importlib.import_module('joint-regressor-refinement_amd.smpl_model')
    model = sm.load_smpl_model(model_dir)
    R, betas = torch.from_numpy(d['R']), torch.from_numpy(d['betas'])
    verts = oracle.OracleSMPL(model, dtype=torch.float64)(R[:, :1].double(), R[:, 1:].double(), betas.double()).vertices
    print('oracle vs smplx: max |dv| =', float((verts - torch.from_numpy(d['vertices']).double()).abs().max()), 'm')
    if torch.cuda.is_available():
        smpl = importlib.import_module('joint-regressor-refinement_amd.smpl').SMPL(model=model).to('cuda:0')
        v = smpl(global_orient=R[:, :1].cuda(), body_pose=R[:, 1:].cuda(), betas=betas.cuda()).vertices
        print('HIP vs smplx:    max |dv| =', float((v.cpu().double() - torch.from_numpy(d['vertices']).double()).abs().max()), 'm')


if __name__ == '__main__':
    if sys.argv[1] == '--check':
        check(sys.argv[2], sys.argv[3])
    else:
        dump(sys.argv[1], sys.argv[2])
