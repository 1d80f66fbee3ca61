"""Precomputed Human3.6M tensors in the reference's on-disk layout.

The reference's `data_set(set)` (/root/reference/scripts/data.py:28-163) reads, per split directory
`data/human3.6m/precomputed_{train,val}/`:
    bboxes.pt  betas.pt  estimated_translation.pt  gt_j2d.pt  gt_j3d.pt  intrinsics.pt  orient.pt  pose.pt
    images.pkl  pixel_annotations.pkl
and, per sample, JPEG frames + Mask-RCNN masks that it crops with a differentiable image sampler
(data.py:110-128, 220-271).  Human3.6M is licensed and absent; the image pipeline (imageio, the
similarity-warp sampler of scripts/linearized.py / sampling_helper.py) is out of scope.  This module keeps
the TENSOR part of the contract so that a user with the data can feed real batches to the HIP path:
  * the same file names and split directories,
  * the crop parameters of find_crop (data.py:220-247) computed from the bounding boxes alone,
  * gt_j2d repositioned into the 224-crop pixel frame exactly as data.py:134-138,
  * per-sample dict keys as data.py:140-158 minus the image-valued ones ('image', 'spin_image', 'mask_rcnn',
    'valid').
"""
from __future__ import annotations

import os
from typing import Dict

import torch
from torch.utils.data import Dataset

TENSOR_FILES = ['bboxes', 'betas', 'estimated_translation', 'gt_j2d', 'gt_j3d', 'intrinsics', 'orient', 'pose']


def crop_params(bboxes: torch.Tensor):
    """find_crop's geometry (data.py:222-247): bboxes (N,4) = (min_y, min_x, max_y, max_x) in the 1000-px frame
    -> (min_x, min_y, scale) of the square crop; scale = half side in units of 500 px."""
    min_x, max_x = (bboxes[:, 1] - 500) / 500, (bboxes[:, 3] - 500) / 500
    min_y, max_y = (bboxes[:, 0] - 500) / 500, (bboxes[:, 2] - 500) / 500
    average_x, average_y = (min_x + max_x) / 2, (min_y + max_y) / 2
    scale = torch.maximum(max_x - min_x, max_y - min_y) / 2
    return (average_x - scale) * 500 + 500, (average_y - scale) * 500 + 500, scale


def reposition_j2d(gt_j2d: torch.Tensor, bboxes: torch.Tensor) -> torch.Tensor:
    """data.py:134-138: 2-D joints (N,17,2) in the 1000-px frame -> the 224-crop pixel frame."""
    min_x, min_y, scale = crop_params(bboxes)
    out = gt_j2d.clone()
    out[..., 0] -= min_x[:, None]
    out[..., 1] -= min_y[:, None]
    out /= scale[:, None, None]
    out /= 1000 / 224
    return out


class data_set(Dataset):
    """data_set("train" | "validation"): tensor-valued samples of the reference's dataset."""

    def __init__(self, set: str, root: str = 'data/human3.6m'):
        location = os.path.join(root, 'precomputed_train' if set == 'train' else 'precomputed_val')
        missing = [f for f in TENSOR_FILES if not os.path.exists(os.path.join(location, f + '.pt'))]
        if missing:
            raise FileNotFoundError(f'{location}: missing {missing} (Human3.6M precomputed tensors are not shipped; '
                                    f'use the synthetic batches of smpl_model.synthetic_batch instead)')
        for f in TENSOR_FILES:
            setattr(self, f, torch.load(os.path.join(location, f + '.pt'), map_location='cpu').float())
        n = self.gt_j3d.shape[0]
        for f in TENSOR_FILES:
            if getattr(self, f).shape[0] != n:
                raise ValueError(f'{f}.pt has {getattr(self, f).shape[0]} rows, gt_j3d.pt has {n}')
        self.inc_gt = torch.ones(n, dtype=torch.bool)
        self.gt_j2d_crop = reposition_j2d(self.gt_j2d, self.bboxes)

    def __len__(self):
        return self.gt_j3d.shape[0]

    def __getitem__(self, index) -> Dict[str, torch.Tensor]:
        return {'bboxes': self.bboxes[index], 'betas': self.betas[index], 'cam': self.estimated_translation[index],
                'gt_j2d': self.gt_j2d_crop[index], 'gt_j3d': self.gt_j3d[index], 'intrinsics': self.intrinsics[index],
                'orient': self.orient[index], 'pose': self.pose[index], 'inc_gt': self.inc_gt[index]}
