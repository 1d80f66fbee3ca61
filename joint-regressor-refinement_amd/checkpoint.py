"""The checkpoint artefact models/retrained_J_Regressor.pt.

Format (inspected on the shipped file): a torch zip archive holding ONE bare float32 tensor of shape
(17, 6890) -- the RAW (un-normalised, possibly negative) parameter; the shipped file stores it with
stride (1, 17), storage tagged 'cuda:0' and requires_grad=True.  Readers call
`torch.load('models/retrained_J_Regressor.pt').to(device)` and apply ReLU + row-normalisation
themselves (/root/reference/scripts/test.py:46-47,206-208; or inside find_joints).
"""
from __future__ import annotations

import os

import torch

SHAPE = (17, 6890)


def load_j_regressor(path: str, device='cpu') -> torch.Tensor:
    """Read a checkpoint with arbitrary strides / device tag / requires_grad; returns a contiguous
    float32 (17, 6890) tensor on `device` (detached)."""
    t = torch.load(path, map_location='cpu', weights_only=True)
    if not torch.is_tensor(t):
        raise ValueError(f'{path}: expected a bare tensor, got {type(t).__name__}')
    if tuple(t.shape) != SHAPE:
        raise ValueError(f'{path}: expected shape {SHAPE}, got {tuple(t.shape)}')
    return t.detach().to(torch.float32).contiguous().to(device)


def save_j_regressor(J: torch.Tensor, path: str, reference_layout: bool = True) -> None:
    """Write the raw parameter so that `torch.load(path)` returns a float32 (17, 6890) tensor.
    With reference_layout=True the tensor is stored column-major (stride (1, 17)) and with
    requires_grad=True like the shipped artefact; it is saved from CPU memory so that it loads on
    machines without a GPU (the reference's readers move it with .to(device) anyway)."""
    if tuple(J.shape) != SHAPE:
        raise ValueError(f'expected shape {SHAPE}, got {tuple(J.shape)}')
    t = J.detach().to('cpu', torch.float32)
    if reference_layout:
        t = t.t().contiguous().t()          # shape (17, 6890), stride (1, 17)
        t.requires_grad_(True)
    os.makedirs(os.path.dirname(os.path.abspath(path)) or '.', exist_ok=True)
    torch.save(t, path)
