"""Pose / shape discriminators with the reference's module structure and state_dict keys
(/root/reference/scripts/discriminator.py:7-74), evaluated by the HIP kernels.

`Discriminator()` and `Shape_Discriminator()` are nn.Modules whose parameters use torch's default
initialisation (same constructors, same order => same values under the same seed as the
reference).  `forward()` runs the C-ABI path through an autograd.Function that returns gradients
w.r.t. the INPUT (what the inner loop uses, scripts/optimize.py:241-253) AND w.r.t. the module's
parameters (what `loss.backward(); disc_optimizer.step()` needs, scripts/optimize.py:276-293), so the
modules train exactly like the reference's.  They own a model-less engine per (batch, device): no SMPL
model is needed to evaluate a discriminator.  The fused driver (optimize.py) does not go through the
modules; it uses the flat-parameter entry points (jrr_pose_disc_backward_params, jrr_adam_step).
"""
from __future__ import annotations

from typing import Dict

import torch
from torch import nn

from . import engine as _engine


class _ModuleEngines:
    """(batch, device) -> model-less RefineEngine holding this module's parameters; the 7.4 MB upload (and the two
    weight transposes behind it) is repeated only when a parameter has changed (torch's per-tensor version counter).
    The engine itself records which parameter version it holds (`param_version`), so that anything else that uploads
    weights into it (the restore path of _PoseDiscFn.backward) invalidates the record.  At most MAX_ENGINES engines are
    kept per module (least recently used first out): a workspace is ~60 KB per pose for the pose discriminator."""
    MAX_ENGINES = 4

    def __init__(self, flag: int):
        self.flag = flag
        self.engines: Dict = {}

    def get(self, module: nn.Module, batch: int, device, flat: torch.Tensor, setter: str):
        key = (batch, str(device))
        eng = self.engines.pop(key, None)
        if eng is None:
            eng = _engine.RefineEngine(None, batch, flags=self.flag, device=device)
            eng.param_version = None
            while len(self.engines) >= self.MAX_ENGINES:
                self.engines.pop(next(iter(self.engines)))
        self.engines[key] = eng                      # most recently used last
        version = tuple((p.data_ptr(), p._version) for p in module.parameters())
        if eng.param_version != version:
            getattr(eng, setter)(flat.detach())
            eng.param_version = version
        return eng


class _PoseDiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flat, eng):
        x = x.detach().contiguous().float()
        out = eng.pose_disc_forward(x)
        ctx.eng, ctx.gen = eng, eng.generation
        ctx.save_for_backward(x, flat.detach())
        return out

    @staticmethod
    def backward(ctx, gout):
        x, flat = ctx.saved_tensors
        eng = ctx.eng
        gout = gout.contiguous()
        if eng.generation != ctx.gen:          # another call has used the engine since: restore this forward's state
            eng.set_pose_disc(flat)            # (possibly older weights than the module's current ones:
            eng.param_version = None           #  the module must upload again before its next forward)
            eng.pose_disc_forward(x)
        dx = eng.pose_disc_vjp_input(x, gout) if ctx.needs_input_grad[0] else None
        dflat = None
        if ctx.needs_input_grad[1]:
            dflat = torch.zeros_like(flat)
            eng.pose_disc_vjp_params(x, gout, dflat)   # re-runs the forward with ROW-MAJOR activations: the quad
        ctx.gen = -1                                   # state pose_disc_vjp_input reads is gone -> a second backward
        return dx, dflat, None                         # through this node (retain_graph) restores it first


class Discriminator(nn.Module):
    def __init__(self):
        super().__init__()
        self.num_inputs = 24
        self.conv_operations = nn.Sequential(nn.Conv2d(6, 32, 1), nn.ReLU(), nn.Conv2d(32, 32, 1), nn.ReLU())
        self.linears = nn.ModuleList([nn.Linear(32, 1) for _ in range(self.num_inputs)])
        self.linear_operations = nn.Sequential(nn.Linear(32 * self.num_inputs, 1024), nn.ReLU(), nn.Linear(1024, 1024),
                                               nn.ReLU(), nn.Linear(1024, 1))
        self._jrr = _ModuleEngines(_engine.FLAG_POSE_DISC)

    def flat_parameters(self) -> torch.Tensor:
        return _engine.flatten_state_dict(self.state_dict(), _engine.DISC_KEYS)

    def _flat_with_grad(self) -> torch.Tensor:
        """the flat parameter vector as a differentiable function of the nn.Parameters (state_dict order)"""
        named = dict(self.named_parameters())
        return torch.cat([named[k].reshape(-1) for k in _engine.DISC_KEYS])

    def forward(self, rot6d: torch.Tensor, model=None) -> torch.Tensor:
        """rot6d (B,24,6) -> (B,25,1) sigmoid scores (output 0 global, 1..24 per joint).  `model` is accepted for
        backward compatibility and ignored."""
        flat = self._flat_with_grad().to(rot6d.device)
        eng = self._jrr.get(self, rot6d.shape[0], rot6d.device, flat, 'set_pose_disc')
        return _PoseDiscFn.apply(rot6d.contiguous(), flat, eng).unsqueeze(-1)


class _ShapeDiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, betas, flat, eng):
        betas = betas.detach().contiguous().float()
        ctx.eng = eng
        ctx.save_for_backward(betas, flat.detach())
        return eng.shape_disc_forward(betas)

    @staticmethod
    def backward(ctx, gout):
        betas, flat = ctx.saved_tensors
        eng = ctx.eng
        eng.set_shape_disc(flat)               # 684 bytes: the vjp kernels recompute the forward, so this is all the state
        gout = gout.contiguous()
        db = eng.shape_disc_vjp_input(betas, gout) if ctx.needs_input_grad[0] else None
        dflat = None
        if ctx.needs_input_grad[1]:
            dflat = torch.zeros_like(flat)
            eng.shape_disc_vjp_params(betas, gout, dflat)
        return db, dflat, None


class Shape_Discriminator(nn.Module):
    def __init__(self):
        super().__init__()
        self.shape_operations = nn.Sequential(nn.Linear(10, 10), nn.ReLU(), nn.Linear(10, 5), nn.ReLU(), nn.Linear(5, 1))
        self._jrr = _ModuleEngines(_engine.FLAG_SHAPE_DISC)

    def flat_parameters(self) -> torch.Tensor:
        return _engine.flatten_state_dict(self.state_dict(), _engine.SHAPE_DISC_KEYS)

    def forward(self, betas: torch.Tensor) -> torch.Tensor:
        """scripts/discriminator.py:70-74: betas (B,10) -> (B,1) sigmoid score."""
        named = dict(self.named_parameters())
        flat = torch.cat([named[k].reshape(-1) for k in _engine.SHAPE_DISC_KEYS]).to(betas.device)
        eng = self._jrr.get(self, betas.shape[0], betas.device, flat, 'set_shape_disc')
        return _ShapeDiscFn.apply(betas.contiguous(), flat, eng).unsqueeze(-1)
