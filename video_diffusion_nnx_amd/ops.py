"""Thin operator wrappers over the C ABI (used by the per-block parity tests and by debugging tools).
The network-level entry points live in unet3d.py / gaussian_diffusion.py."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def _mode(mode) -> int:
    return L.MODES[mode] if isinstance(mode, str) else int(mode)


def pack_conv_weights(kernel: torch.Tensor, mode) -> torch.Tensor:
    """kernel: Flax layout (..., kh, kw, Cin, Cout) / (1, Cin, Cout) / (Cin, Cout) -> packed byte tensor."""
    m = _mode(mode)
    cin, cout = kernel.shape[-2], kernel.shape[-1]
    taps = kernel.numel() // (cin * cout)
    k = kernel.contiguous().float()
    n = L.vdx_packed_conv_bytes(m, taps, cin, cout)
    out = torch.empty(n, dtype=torch.uint8, device=k.device)
    L.check(L.vdx_pack_conv_weights(m, L.ptr(k), L.ptr(out), taps, cin, cout, L.stream_ptr()))
    return out


def gn_stats_zeros(batch: int, groups: int, device) -> torch.Tensor:
    return torch.zeros(batch * L.GN_SLOTS * groups * 2, dtype=torch.float64, device=device)


def gn_stats_reduce(stats: torch.Tensor, batch: int, groups: int) -> torch.Tensor:
    """-> [batch, groups, 2] (sum, sumsq)."""
    return stats.view(batch, L.GN_SLOTS, groups, 2).sum(1)


def conv_forward(x0, packed_w, cout, *, mode, bias=None, x1=None, kind=0, k=3, stride=1,
                 in_stats=None, gamma=None, beta=None, groups=8, scale_shift=None,
                 out_stats=None, out_groups=8) -> torch.Tensor:
    """x0: [B,F,H,W,C0] channel-last fp32 (x1 likewise, concatenated on channels)."""
    B, Fr, H, W, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[-1]
    if kind == 1:
        Ho, Wo = 2 * H, 2 * W
    else:
        Ho, Wo = -(-H // stride), -(-W // stride)
    y = torch.empty(B, Fr, Ho, Wo, cout, dtype=torch.float32, device=x0.device)
    d = L.ConvDesc()
    d.x0, d.x1, d.c0, d.c1 = L.ptr(x0), L.ptr(x1), c0, c1
    d.packed_w, d.bias, d.y, d.cout = L.ptr(packed_w), L.ptr(bias), L.ptr(y), cout
    d.batch, d.frames, d.h, d.w = B, Fr, H, W
    d.kind, d.kh, d.kw, d.stride = kind, k, k, stride
    d.in_stats, d.gamma, d.beta, d.groups = L.ptr(in_stats), L.ptr(gamma), L.ptr(beta), groups
    d.scale_shift = L.ptr(scale_shift)
    d.scale_shift_stride = 0 if scale_shift is None else scale_shift.shape[-1]
    d.out_stats, d.out_groups = L.ptr(out_stats), out_groups
    L.check(L.vdx_conv_forward(_mode(mode), C.byref(d), L.stream_ptr()))
    return y
