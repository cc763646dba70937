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
                 out_stats=None, out_groups=8, y_bf16=False, res=None) -> torch.Tensor:
    """x0: [B,F,H,W,C0] channel-last fp32 -- or bf16 (bf16 mode: bf16 activation storage) -- (x1 likewise, concatenated on
    channels); y_bf16 selects a bf16 output tensor; res: optional residual [B,F,H,W,cout] (fp32 or bf16) added to the output."""
    B, Fr, H, W, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[-1]
    if kind == 1:
        Ho, Wo = 2 * H, 2 * W
    else:
        Ho, Wo = -(-H // stride), -(-W // stride)
    x_bf16 = x0.dtype == torch.bfloat16
    assert x0.dtype in (torch.float32, torch.bfloat16) and (x1 is None or x1.dtype == x0.dtype)
    y = torch.empty(B, Fr, Ho, Wo, cout, dtype=torch.bfloat16 if y_bf16 else torch.float32, device=x0.device)
    d = L.ConvDesc()
    d.x_bf16, d.y_bf16 = int(x_bf16), int(bool(y_bf16))
    d.x0, d.x1, d.c0, d.c1 = L.ptr(x0), L.ptr(x1), c0, c1
    d.packed_w, d.bias, d.y, d.cout = L.ptr(packed_w), L.ptr(bias), L.ptr(y), cout
    d.batch, d.frames, d.h, d.w = B, Fr, H, W
    d.kind, d.kh, d.kw, d.stride = kind, k, k, stride
    d.in_stats, d.gamma, d.beta, d.groups = L.ptr(in_stats), L.ptr(gamma), L.ptr(beta), groups
    d.scale_shift = L.ptr(scale_shift)
    d.scale_shift_stride = 0 if scale_shift is None else scale_shift.shape[-1]
    d.out_stats, d.out_groups = L.ptr(out_stats), out_groups
    d.res, d.res_bf16 = L.ptr(res), int(res is not None and res.dtype == torch.bfloat16)
    L.check(L.vdx_conv_forward(_mode(mode), C.byref(d), L.stream_ptr()))
    return y


def resblock_tail(y2, r, stats, gn_gamma, gn_beta, ln_gamma, ln_beta, groups=8):
    B, C = y2.shape[0], y2.shape[-1]
    pix = y2.numel() // (B * C)
    out = torch.empty_like(y2)
    L.check(L.vdx_resblock_tail(L.ptr(y2), L.ptr(r), L.ptr(out), L.ptr(stats), L.ptr(gn_gamma), L.ptr(gn_beta), groups,
                                L.ptr(ln_gamma), L.ptr(ln_beta), C, B, pix, L.stream_ptr()))
    return out


def resblock_tail_rc_bf16(y2, x0, x1, rc_kernel, rc_bias, stats, gn_gamma, gn_beta, ln_gamma, ln_beta, groups=8):
    """Tail with the 1x1 res_conv inside (bf16 tensors): y2 [B,...,C], x0 [B,...,C0], x1 [B,...,C1] or None (all torch.bfloat16),
    rc_kernel Flax [C0+C1, C] fp32 -> out bf16."""
    B, C = y2.shape[0], y2.shape[-1]
    pix = y2.numel() // (B * C)
    c0, c1 = x0.shape[-1], (0 if x1 is None else x1.shape[-1])
    wp = rc_kernel.t().contiguous().to(torch.bfloat16)           # [C][Cin], K-contiguous: the packed operand layout
    if (c0 + c1) % 64:                                           # (rows padded to 64 input channels, as vdx_pack_conv_weights lays them out)
        wp = torch.nn.functional.pad(wp, (0, 64 - (c0 + c1) % 64)).contiguous()
    out = torch.empty_like(y2)
    L.check(L.vdx_resblock_tail_rc_bf16(L.ptr(y2), L.ptr(x0), None if x1 is None else L.ptr(x1), c0, c1, L.ptr(wp), L.ptr(rc_bias),
                                        L.ptr(out), L.ptr(stats), L.ptr(gn_gamma), L.ptr(gn_beta), groups, L.ptr(ln_gamma),
                                        L.ptr(ln_beta), C, B, pix, L.stream_ptr()))
    return out


def gn_silu_apply_bf16(y, stats, gn_gamma, gn_beta, scale_shift=None, groups=8):
    """Block prologue in place on a bf16 tensor y [B,...,C]: SiLU(GroupNorm(y) * (scale + 1) + shift); scale_shift fp32 [B, 2C] or None."""
    assert y.dtype == torch.bfloat16 and y.is_contiguous()
    B, C = y.shape[0], y.shape[-1]
    pix = y.numel() // (B * C)
    L.check(L.vdx_gn_silu_apply_bf16(L.ptr(y), L.ptr(stats), L.ptr(gn_gamma), L.ptr(gn_beta), None if scale_shift is None else L.ptr(scale_shift),
                                     0 if scale_shift is None else scale_shift.shape[-1], groups, C, B, pix, L.stream_ptr()))
    return y


def init_conv(x, kernel, bias):
    """x [B,C,F,H,W] (external layout); kernel Flax (1,k,k,C,D) -> [B,F,H,W,D]."""
    B, Cin, Fr, H, W = x.shape
    k, cout = kernel.shape[1], kernel.shape[-1]
    y = torch.empty(B, Fr, H, W, cout, dtype=torch.float32, device=x.device)
    L.check(L.vdx_init_conv(L.ptr(x), L.ptr(kernel.contiguous()), L.ptr(bias), L.ptr(y), B, Cin, Fr, H, W, cout, k, L.stream_ptr()))
    return y


def final_conv(x, kernel, bias):
    d, cout = kernel.shape[-2], kernel.shape[-1]
    npix = x.numel() // d
    y = torch.empty(*x.shape[:-1], cout, dtype=torch.float32, device=x.device)
    L.check(L.vdx_final_conv(L.ptr(x), L.ptr(kernel.contiguous()), L.ptr(bias), L.ptr(y), npix, d, cout, L.stream_ptr()))
    return y


def time_mlp(time, w1, b1, w2, b2, cond=None, null_cond_emb=None, cond_mask=None, null_all=False):
    B, dim = time.shape[0], w1.shape[0]
    cond_dim = 0 if cond is None else cond.shape[-1]
    temb = torch.empty(B, 4 * dim + cond_dim, dtype=torch.float32, device=w1.device)
    t32 = time.to(torch.int32).contiguous()
    cm = None if cond_mask is None else cond_mask.to(torch.uint8).contiguous()
    L.check(L.vdx_time_mlp(L.ptr(t32), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), dim, L.ptr(cond), L.ptr(null_cond_emb),
                           L.ptr(cm), int(null_all), cond_dim, L.ptr(temb), B, L.stream_ptr()))
    return temb


def pack_mha(pq, pk, pv, po, mode):
    """pq/pk/pv: (kernel (C,H,D), bias (H,D)); po: (kernel (H,D,C), bias (C))."""
    C_ = pq[0].shape[0]
    wqkv = torch.cat([p[0].reshape(C_, -1) for p in (pq, pk, pv)], dim=1).contiguous()
    bqkv = torch.cat([p[1].reshape(-1) for p in (pq, pk, pv)]).contiguous()
    wo = po[0].reshape(-1, C_).contiguous()
    return pack_conv_weights(wqkv, mode), bqkv, pack_conv_weights(wo, mode), po[1].contiguous()


def attention_forward(x, packed, heads, temporal, mode, fp8_core=False):
    B, Fr, H, W, C_ = x.shape
    y = torch.empty_like(x)
    wqkv, bqkv, wo, bo = packed
    L.check(L.vdx_attention_forward_ex(_mode(mode), L.ptr(x), L.ptr(y), L.ptr(wqkv), L.ptr(bqkv), L.ptr(wo), L.ptr(bo),
                                       B, Fr, H, W, C_, heads, int(temporal), int(bool(fp8_core)), L.stream_ptr()))
    return y


def attention_forward_bf16(x, packed, heads, temporal, fp8_core=False):
    """x: bf16 channel-last [B, F, H, W, C]; packed = pack_mha(..., 'bf16')."""
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    B, Fr, H, W, C_ = x.shape
    y = torch.empty_like(x)
    wqkv, bqkv, wo, bo = packed
    L.check(L.vdx_attention_forward_bf16(L.ptr(x), L.ptr(y), L.ptr(wqkv), L.ptr(bqkv), L.ptr(wo), L.ptr(bo),
                                         B, Fr, H, W, C_, heads, int(temporal), int(bool(fp8_core)), L.stream_ptr()))
    return y


def sla_forward(x, wq, wk, wv, wo, heads, mode):
    """wq/wk/wv: Flax (1, C, 256); wo: (1, 256, C)."""
    B, Fr, H, W, C_ = x.shape
    m = _mode(mode)
    y = torch.empty_like(x)
    ws = torch.empty(L.vdx_sla_workspace_bytes(m, B * Fr, H * W, heads), dtype=torch.uint8, device=x.device)
    pk = [pack_conv_weights(t, mode) for t in (wq, wk, wv, wo)]
    L.check(L.vdx_sla_forward(m, L.ptr(x), L.ptr(y), L.ptr(pk[0]), L.ptr(pk[1]), L.ptr(pk[2]), L.ptr(pk[3]), L.ptr(ws),
                              B, Fr, H, W, C_, heads, L.stream_ptr()))
    return y


def sla_forward_bf16(x, wq, wk, wv, wo, heads=8):
    """x: bf16 channel-last [B, F, H, W, C]; wq/wk/wv: Flax (1, C, 256); wo: (1, 256, C)."""
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    B, Fr, H, W, C_ = x.shape
    m = _mode('bf16')
    y = torch.empty_like(x)
    ws = torch.empty(L.vdx_sla_workspace_bytes(m, B * Fr, H * W, heads), dtype=torch.uint8, device=x.device)
    pk = [pack_conv_weights(t, 'bf16') for t in (wq, wk, wv, wo)]
    L.check(L.vdx_sla_forward_bf16(L.ptr(x), L.ptr(y), L.ptr(pk[0]), L.ptr(pk[1]), L.ptr(pk[2]), L.ptr(pk[3]), L.ptr(ws),
                                   B, Fr, H, W, C_, heads, L.stream_ptr()))
    return y


# ---- backward building blocks ---------------------------------------------------------------------------


class WgradDesc(C.Structure):
    _fields_ = [('x0', C.c_void_p), ('x1', C.c_void_p), ('c0', C.c_int), ('c1', C.c_int), ('dy', C.c_void_p), ('cout', C.c_int),
                ('dw', C.c_void_p), ('batch', C.c_int), ('frames', C.c_int), ('h', C.c_int), ('w', C.c_int),
                ('kind', C.c_int), ('kh', C.c_int), ('kw', C.c_int), ('stride', C.c_int),
                ('in_stats', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p), ('groups', C.c_int),
                ('scale_shift', C.c_void_p), ('scale_shift_stride', C.c_int), ('bf16_operands', C.c_int)]


_pack_t = L._sig('vdx_pack_conv_weights_t', C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p])
_wgrad = L._sig('vdx_conv_backward_weights', C.c_int, [C.POINTER(WgradDesc), C.c_void_p])
_colsum = L._sig('vdx_colsum', C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p])


def pack_conv_weights_t(kernel: torch.Tensor, mode) -> torch.Tensor:
    m = _mode(mode)
    cin, cout = kernel.shape[-2], kernel.shape[-1]
    taps = kernel.numel() // (cin * cout)
    k = kernel.contiguous().float()
    out = torch.empty(L.vdx_packed_conv_bytes(m, taps, cout, cin), dtype=torch.uint8, device=k.device)
    L.check(_pack_t(m, L.ptr(k), L.ptr(out), taps, cin, cout, L.stream_ptr()))
    return out


def conv_backward_weights(x0, dy, kshape, *, x1=None, kind=0, k=3, stride=1, in_stats=None, gamma=None, beta=None, groups=8,
                          scale_shift=None, dw=None, bf16_operands=False) -> torch.Tensor:
    B, Fr, H, W, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[-1]
    cout = dy.shape[-1]
    if dw is None:
        dw = torch.zeros(kshape, dtype=torch.float32, device=x0.device)
    d = WgradDesc()
    d.x0, d.x1, d.c0, d.c1, d.dy, d.cout, d.dw = L.ptr(x0), L.ptr(x1), c0, c1, L.ptr(dy), cout, L.ptr(dw)
    d.batch, d.frames, d.h, d.w = B, Fr, H, W
    d.kind, d.kh, d.kw, d.stride = kind, k, k, stride
    d.in_stats, d.gamma, d.beta, d.groups = L.ptr(in_stats), L.ptr(gamma), L.ptr(beta), groups
    d.scale_shift = L.ptr(scale_shift)
    d.scale_shift_stride = 0 if scale_shift is None else scale_shift.shape[-1]
    d.bf16_operands = int(bool(bf16_operands))
    L.check(_wgrad(C.byref(d), L.stream_ptr()))
    return dw


def colsum(x: torch.Tensor) -> torch.Tensor:
    c = x.shape[-1]
    out = torch.zeros(c, dtype=torch.float32, device=x.device)
    L.check(_colsum(L.ptr(x), L.ptr(out), x.numel() // c, c, L.stream_ptr()))
    return out


_norm_bwd = L._sig('vdx_norm_act_backward', C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_long, C.c_void_p])


_norm_bwd_scr = L._sig('vdx_norm_act_backward_scratch_floats', C.c_size_t, [C.c_int, C.c_int, C.c_long])


def norm_act_backward(dact, y, stats, gamma, beta, groups=8, scale_shift=None, r=None, ln_gamma=None):
    """-> dict(dy, d_gamma, d_beta, dss, dr, d_ln_gamma, d_ln_beta)"""
    B, Cc = y.shape[0], y.shape[-1]
    pix = y.numel() // (B * Cc)
    dev = y.device
    z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
    out = dict(dy=torch.empty_like(y), d_gamma=z(Cc), d_beta=z(Cc), dss=z(B, 2 * Cc) if scale_shift is not None else None,
               dr=torch.empty_like(y) if r is not None else None, d_ln_gamma=z(Cc) if r is not None else None,
               d_ln_beta=z(Cc) if r is not None else None)
    scratch = torch.empty(_norm_bwd_scr(Cc, B, pix), dtype=torch.float32, device=dev)       # uninitialised on purpose: nothing accumulates into it
    L.check(_norm_bwd(L.ptr(dact), L.ptr(y), L.ptr(out['dy']), L.ptr(stats), L.ptr(gamma), L.ptr(beta), groups, L.ptr(scale_shift),
                      0 if scale_shift is None else scale_shift.shape[-1], L.ptr(out['d_gamma']), L.ptr(out['d_beta']), L.ptr(out['dss']),
                      L.ptr(r), L.ptr(ln_gamma), L.ptr(out['dr']), L.ptr(out['d_ln_gamma']), L.ptr(out['d_ln_beta']), L.ptr(scratch),
                      Cc, B, pix, L.stream_ptr()))
    return out


_attn_core_bwd = L._sig('vdx_attention_core_backward_ex', C.c_int, [C.c_void_p] * 6 + [C.c_int] * 7 + [C.c_void_p])
_sla_scr = L._sig('vdx_sla_backward_scratch_floats', C.c_size_t, [C.c_int, C.c_int])
_sla_core_bwd = L._sig('vdx_sla_core_backward_ex', C.c_int, [C.c_void_p] * 9 + [C.c_int] * 4 + [C.c_void_p])


def attention_core_backward(qkv, d_o, B, Fr, H, W, heads, temporal, bf16_operands=False):
    outs = [torch.empty_like(d_o) for _ in range(4)]
    L.check(_attn_core_bwd(L.ptr(qkv), L.ptr(d_o), *[L.ptr(t) for t in outs], B, Fr, H, W, heads, int(temporal), int(bool(bf16_operands)),
                           L.stream_ptr()))
    return outs      # o, dq, dk, dv


_attn_bwd_fused = L._sig('vdx_temporal_attention_backward_fused', C.c_int, [C.c_void_p] * 8 + [C.c_int] * 4 + [C.c_void_p])


def temporal_attention_backward_fused(x, dy, wqkv, bqkv, wo):
    """x, dy: fp32 (B, F, H, W, 64); wqkv (64, 768) = [Wq | Wk | Wv], bqkv (768,), wo (256, 64) Flax kernels.
    Returns dx (fp32), o (rows, 256) and dqkv (rows, 768) as bfloat16 (the bf16-mode backward of the widest level's temporal attention)."""
    B, Fr, H, W, C_ = x.shape
    assert C_ == 64 and wqkv.shape == (64, 768) and wo.shape == (256, 64)
    pw = pack_conv_weights(wqkv, 'bf16')
    pwo_t = pack_conv_weights_t(wo, 'bf16')
    rows = B * Fr * H * W
    o = torch.empty(rows, 256, dtype=torch.bfloat16, device=x.device)
    dqkv = torch.empty(rows, 768, dtype=torch.bfloat16, device=x.device)
    dx = torch.empty_like(x)
    bq = bqkv.contiguous().float()
    L.check(_attn_bwd_fused(L.ptr(x), L.ptr(dy), L.ptr(pw), L.ptr(bq), L.ptr(pwo_t), L.ptr(o), L.ptr(dqkv), L.ptr(dx), B, Fr, H, W, L.stream_ptr()))
    return dx, o, dqkv


def sla_core_backward(q, k, v, d_out, nframes, npix, heads=8, bf16_operands=False):
    outs = [torch.empty_like(q) for _ in range(4)]
    scr = torch.empty(_sla_scr(nframes, heads), dtype=torch.float32, device=q.device)
    L.check(_sla_core_bwd(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(d_out), *[L.ptr(t) for t in outs], L.ptr(scr), nframes, npix, heads,
                          int(bool(bf16_operands)), L.stream_ptr()))
    return outs      # o, dq, dk, dv
