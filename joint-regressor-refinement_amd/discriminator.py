"""Pose / shape discriminators with the reference's module structure and state_dict keys
(/root/reference/scripts/discriminator.py:7-74), evaluated by the HIP kernels.

`Discriminator()` and `Shape_Discriminator()` are nn.Modules whose parameters use torch's default
initialisation (same constructors, same order => same values under the same seed as the
reference).  forward() runs the C-ABI path (jrr_pose_disc_forward / _backward_input) through an
autograd.Function; gradients w.r.t. the INPUT are provided (that is what the inner loop uses,
scripts/optimize.py:241-253).
"""
from __future__ import annotations

from typing import Dict

import torch
from torch import nn

from . import engine as _engine


class Discriminator(nn.Module):
    def __init__(self):
        super().__init__()
        self.num_inputs = 24
        self.conv_operations = nn.Sequential(nn.Conv2d(6, 32, 1), nn.ReLU(), nn.Conv2d(32, 32, 1), nn.ReLU())
        self.linears = nn.ModuleList([nn.Linear(32, 1) for _ in range(self.num_inputs)])
        self.linear_operations = nn.Sequential(nn.Linear(32 * self.num_inputs, 1024), nn.ReLU(), nn.Linear(1024, 1024),
                                               nn.ReLU(), nn.Linear(1024, 1))
        self._engines: Dict = {}

    def flat_parameters(self) -> torch.Tensor:
        return _engine.flatten_state_dict(self.state_dict(), _engine.DISC_KEYS)

    def _engine_for(self, batch: int, device, model):
        key = (batch, str(device))
        if key not in self._engines:
            self._engines[key] = _engine.RefineEngine(model, batch, flags=_engine.FLAG_POSE_DISC)
        eng = self._engines[key]
        eng.set_pose_disc(self.flat_parameters().to(device))
        return eng

    def forward(self, rot6d: torch.Tensor, model=None) -> torch.Tensor:
        """rot6d (B,24,6) -> (B,25,1) sigmoid scores (output 0 global, 1..24 per joint)."""
        if model is None:
            model = getattr(self, 'device_model', None)
        if model is None:
            raise RuntimeError('Discriminator.forward needs a DeviceModel (set .device_model or pass model=)')
        eng = self._engine_for(rot6d.shape[0], rot6d.device, model)
        return _PoseDiscFn.apply(rot6d.contiguous(), eng).unsqueeze(-1)


class _PoseDiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eng):
        x = x.detach().contiguous().float()
        ctx.eng = eng
        ctx.save_for_backward(x)
        return eng.pose_disc_forward(x)

    @staticmethod
    def backward(ctx, gout):
        (x,) = ctx.saved_tensors
        return ctx.eng.pose_disc_vjp_input(x, gout.contiguous()), None


class Shape_Discriminator(nn.Module):
    def __init__(self):
        super().__init__()
        self.shape_operations = nn.Sequential(nn.Linear(10, 10), nn.ReLU(), nn.Linear(10, 5), nn.ReLU(), nn.Linear(5, 1))

    def flat_parameters(self) -> torch.Tensor:
        return _engine.flatten_state_dict(self.state_dict(), _engine.SHAPE_DISC_KEYS)
