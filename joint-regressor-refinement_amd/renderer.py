"""2-D projection of the regressed joints: `return_2d_joints` of the reference
(/root/reference/scripts/renderer.py:10-51) -- SURVEY.md section 8 row f1.

pytorch3d 0.3.0 `PerspectiveCameras(T=batch['cam'], focal_length=5000/224, principal_point=0)` +
`transform_points_screen(points, 224x224)` restated per SURVEY.md Appendix B (pytorch3d is absent from the
image: parity unpinned).  The operator-level function below is differentiable through torch ops on the
HIP `find_joints`; the fused inner loop and the camera pre-fit use the kernels
(jrr_engine_set_reprojection / jrr_camera_prefit)."""
from __future__ import annotations

import torch

from . import utils

FOCAL = 5000.0 / 224.0
IMAGE = 224.0


def project_points(point_cloud: torch.Tensor, cam: torch.Tensor) -> torch.Tensor:
    """renderer.py:35-49: flip x,y and scale by 2, translate by cam, perspective divide, NDC -> screen.
    (B,N,3), (B,3) -> (B,N,3) with [..., :2] the screen coordinates (the only part the reference reads)."""
    X = -2 * point_cloud[..., 0] + cam[:, None, 0]
    Y = -2 * point_cloud[..., 1] + cam[:, None, 1]
    Z = 2 * point_cloud[..., 2] + cam[:, None, 2]
    xs = (IMAGE - 1) / 2 * (1 - FOCAL * X / Z)
    ys = (IMAGE - 1) / 2 * (1 - FOCAL * Y / Z)
    return torch.stack([xs, ys, Z], dim=-1)


def return_2d_joints(batch, smpl, J_regressor=None, mask=None):
    """batch: dict with 'pose' (B,23,6), 'orient' (B,1,6), 'betas' (B,10), 'cam' (B,3)."""
    pose = utils.rot6d_to_rotmat(batch['pose'].reshape(-1, 6)).reshape(-1, 23, 3, 3)
    orient = utils.rot6d_to_rotmat(batch['orient'].reshape(-1, 6)).reshape(-1, 1, 3, 3)
    if J_regressor is not None:
        point_cloud = utils.find_joints(smpl, batch['betas'], orient, pose, J_regressor, mask=mask)
    else:
        point_cloud = smpl(betas=batch['betas'], body_pose=pose, global_orient=orient, pose2rot=False).vertices
    return project_points(point_cloud, batch['cam'])
