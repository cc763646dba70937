"""The device side of one train step (reference `_pjit_train_step`, trainer.py:322-392), driven stage by stage so the
gradient all-reduce of finished buckets overlaps the rest of the backward."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .gaussian_diffusion import split_key, vdx_loss_sum, vdx_q_sample

_vp = C.c_void_p
vdx_loss_grad = L._sig('vdx_loss_grad', C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_long, C.c_int, _vp])
vdx_adam_ema_step = L._sig('vdx_adam_ema_step', C.c_int, [_vp] * 5 + [C.c_long] + [C.c_float] * 4 + [C.c_long, C.c_float, C.c_int, C.c_float, _vp])


def stage_of_param(name: str, n_levels: int) -> int:
    """Backward stage (== forward position) of a parameter: 0 = stem + time MLP, 1..n = downs, n+1 = mid,
    n+2..2n+1 = ups, 2n+2 = head.  stage_of_param('__count__', n) returns the number of stages."""
    if name == '__count__':
        return 2 * n_levels + 3
    head = name.split('.')[0]
    if head in ('time_rel_pos_bias', 'init_conv', 'init_temporal_attn', 'time_mlp', 'null_cond_emb'):
        return 0
    if head == 'downs':
        return 1 + int(name.split('.')[1])
    if head.startswith('mid_'):
        return n_levels + 1
    if head == 'ups':
        return n_levels + 2 + int(name.split('.')[1])
    if head == 'final_conv':
        return 2 * n_levels + 2
    raise KeyError(name)


def run_train_step(tr, batch: torch.Tensor, step: int) -> torch.Tensor:
    raise NotImplementedError('the HIP backward (vdx_unet_backward) lands next; see DESIGN.md §6')
