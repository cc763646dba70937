"""ORACLE (test infrastructure only) -- CPU restatement of the reference Unet3D forward.

This file is NOT part of the product path.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it.  It restates, in plain PyTorch-CPU ops
(fp32 or fp64), WHAT the reference computes, written from the reference's source text:

  * network graph ............ /root/reference/unet3d.py:58-387
  * building blocks .......... /root/reference/modules.py:21-396
  * Up/Downsample ............ /root/reference/utils.py:103-125
  * prob_mask_like ........... /root/reference/utils.py:85-101

The arithmetic itself lives in un-vendored third-party libraries (jax / flax.nnx, unpinned in
/root/reference/requirements.txt:1-4, not installable here), so the Flax layer semantics are
restated from their published behaviour:
  Conv: channel-last, SAME padding, cross-correlation, kernel (*k, Cin, Cout);
  ConvTranspose(transpose_kernel=False): lhs-dilate by stride, pad (2,2) for k=4/s=2, correlate
  with the UNFLIPPED kernel; LayerNorm/GroupNorm: eps 1e-6, fast variance E[x^2]-E[x]^2 clamped
  at 0; gelu = tanh approximation; softmax = max-subtracted.

PARITY STATUS: the reference's own tests pin only SHAPES and DTYPES for the network
(/root/reference/test_unet3d.py:12-60, /root/reference/test_modules.py:13-293), so the numeric
values of this UNet restatement are "parity unpinned" against the real JAX reference; the
reference quirks that define the computed function (PreNorm no-op, unused SLA scale, double
LayerNorm in ResnetBlock, ...) are reproduced and are each covered by an
invariant test in tests/test_oracle_unet.py.

Parameters are a flat dict  {nnx-state-path: torch tensor}  in Flax layouts, e.g.
  'downs.0.0.block_1.proj.kernel' : (1, 3, 3, Cin, Cout)
  'downs.0.3.fn.fn.fn.q.kernel'   : (C, heads, dim_head)
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

BERT_MODEL_DIM = 768  # constant of the external `video_diffusion_pytorch.text` (unet3d.py:10)
NORM_EPS = 1e-6       # Flax LayerNorm / GroupNorm default epsilon


@dataclass
class UnetConfig:
    """Mirror of the Unet3D constructor arguments (unet3d.py:58-75)."""
    dim: int
    dim_mults: Tuple[int, ...] = (1, 2, 4, 8)
    cond_dim: Optional[int] = None
    out_dim: Optional[int] = None
    channels: int = 3
    attn_heads: int = 8
    attn_dim_head: int = 32
    use_bert_text_cond: bool = False
    init_dim: Optional[int] = None
    init_kernel_size: int = 7
    use_sparse_linear_attn: bool = True
    resnet_groups: int = 8
    sla_heads: int = 8       # unet3d.py:174 passes heads=attn_heads, D=32
    sla_dim_head: int = 32

    def __post_init__(self):
        self.dim_mults = tuple(self.dim_mults)
        self.sla_heads = self.attn_heads
        self.has_cond = (self.cond_dim is not None) or self.use_bert_text_cond  # unet3d.py:136
        self.cond_in = BERT_MODEL_DIM if self.use_bert_text_cond else self.cond_dim  # :137
        self.init_dim_ = self.init_dim if self.init_dim is not None else self.dim   # :103
        self.time_dim = self.dim * 4                                                 # :127
        self.temb_dim = self.time_dim + int(self.cond_in or 0)                       # :150
        self.out_dim_ = self.out_dim if self.out_dim is not None else self.channels  # :242
        dims = [self.init_dim_] + [self.dim * m for m in self.dim_mults]             # :123
        self.in_out = list(zip(dims[:-1], dims[1:]))                                 # :124
        self.mid_dim = dims[-1]


# ----------------------------------------------------------------------------------------------
# parameter inventory (names follow the nnx state tree)
# ----------------------------------------------------------------------------------------------

def _resnet_spec(prefix: str, cin: int, cout: int, temb: Optional[int]) -> List[Tuple[str, tuple]]:
    s = []
    if temb is not None:                                   # modules.py:202-207
        s += [(f'{prefix}.mlp.layers.1.kernel', (temb, 2 * cout)),
              (f'{prefix}.mlp.layers.1.bias', (2 * cout,))]
    s += [(f'{prefix}.norm_1.scale', (2 * cout,)), (f'{prefix}.norm_1.bias', (2 * cout,))]  # :208
    s += [(f'{prefix}.block_1.proj.kernel', (1, 3, 3, cin, cout)), (f'{prefix}.block_1.proj.bias', (cout,)),
          (f'{prefix}.block_1.norm.scale', (cout,)), (f'{prefix}.block_1.norm.bias', (cout,)),
          (f'{prefix}.block_2.proj.kernel', (1, 3, 3, cout, cout)), (f'{prefix}.block_2.proj.bias', (cout,)),
          (f'{prefix}.block_2.norm.scale', (cout,)), (f'{prefix}.block_2.norm.bias', (cout,))]
    if cin != cout:                                        # modules.py:219-222
        s += [(f'{prefix}.res_conv.kernel', (1, cin, cout)), (f'{prefix}.res_conv.bias', (cout,))]
    s += [(f'{prefix}.norm_2.scale', (cout,)), (f'{prefix}.norm_2.bias', (cout,))]          # :223
    return s


def _mha_spec(prefix: str, c: int, heads: int, d: int) -> List[Tuple[str, tuple]]:
    # Residual(PreNorm(dim, EinopsToAndFrom(..., MultiheadAttention))) : unet3d.py:86-96,118-120
    s = [(f'{prefix}.fn.norm.scale', (c,)), (f'{prefix}.fn.norm.bias', (c,))]
    for n in ('q', 'k', 'v'):
        s += [(f'{prefix}.fn.fn.fn.{n}.kernel', (c, heads, d)), (f'{prefix}.fn.fn.fn.{n}.bias', (heads, d))]
    s += [(f'{prefix}.fn.fn.fn.out.kernel', (heads, d, c)), (f'{prefix}.fn.fn.fn.out.bias', (c,))]
    return s


def _sla_spec(prefix: str, c: int, heads: int, d: int) -> List[Tuple[str, tuple]]:
    # Residual(PreNorm(dim, SpatialLinearAttention)) : unet3d.py:170-178, modules.py:64-91
    s = [(f'{prefix}.fn.norm.scale', (c,)), (f'{prefix}.fn.norm.bias', (c,))]
    hd = heads * d
    for n in ('q', 'k', 'v'):
        s += [(f'{prefix}.fn.fn.{n}.kernel', (1, c, hd))]
    s += [(f'{prefix}.fn.fn.to_out.kernel', (1, hd, c))]
    return s


def param_spec(cfg: UnetConfig) -> List[Tuple[str, tuple]]:
    """Ordered (name, shape) list of every Unet3D parameter, in construction order."""
    s: List[Tuple[str, tuple]] = []
    H, Dh = cfg.attn_heads, cfg.attn_dim_head
    s += [('time_rel_pos_bias.relative_attention_bias.embedding', (32, H))]            # unet3d.py:98
    k = cfg.init_kernel_size
    s += [('init_conv.kernel', (1, k, k, cfg.channels, cfg.init_dim_)), ('init_conv.bias', (cfg.init_dim_,))]
    s += _mha_spec('init_temporal_attn', cfg.init_dim_, H, Dh)
    s += [('time_mlp.layers.1.kernel', (cfg.dim, cfg.time_dim)), ('time_mlp.layers.1.bias', (cfg.time_dim,)),
          ('time_mlp.layers.3.kernel', (cfg.time_dim, cfg.time_dim)), ('time_mlp.layers.3.bias', (cfg.time_dim,))]
    if cfg.has_cond:
        s += [('null_cond_emb', (1, cfg.cond_in))]
    n_res = len(cfg.in_out)
    for i, (din, dout) in enumerate(cfg.in_out):                                       # unet3d.py:163-189
        s += _resnet_spec(f'downs.{i}.0', din, dout, cfg.temb_dim)
        s += _resnet_spec(f'downs.{i}.1', dout, dout, cfg.temb_dim)
        if cfg.use_sparse_linear_attn:
            s += _sla_spec(f'downs.{i}.2', dout, cfg.sla_heads, cfg.sla_dim_head)
        s += _mha_spec(f'downs.{i}.3', dout, H, Dh)
        if i < n_res - 1:
            s += [(f'downs.{i}.4.kernel', (1, 4, 4, dout, dout)), (f'downs.{i}.4.bias', (dout,))]
    m = cfg.mid_dim
    s += _resnet_spec('mid_block1', m, m, cfg.temb_dim)
    s += _mha_spec('mid_spatial_attn', m, H, Dh)
    s += _mha_spec('mid_temporal_attn', m, H, Dh)
    s += _resnet_spec('mid_block2', m, m, cfg.temb_dim)
    for i, (din, dout) in enumerate(reversed(cfg.in_out)):                             # unet3d.py:213-240
        s += _resnet_spec(f'ups.{i}.0', dout * 2, din, cfg.temb_dim)
        s += _resnet_spec(f'ups.{i}.1', din, din, cfg.temb_dim)
        if cfg.use_sparse_linear_attn:
            s += _sla_spec(f'ups.{i}.2', din, cfg.sla_heads, cfg.sla_dim_head)
        s += _mha_spec(f'ups.{i}.3', din, H, Dh)
        if i < n_res - 1:
            s += [(f'ups.{i}.4.kernel', (1, 4, 4, din, din)), (f'ups.{i}.4.bias', (din,))]
    s += _resnet_spec('final_conv.layers.0', cfg.dim * 2, cfg.dim, None)               # unet3d.py:249-252
    s += [('final_conv.layers.1.kernel', (1, cfg.dim, cfg.out_dim_)), ('final_conv.layers.1.bias', (cfg.out_dim_,))]
    return s


# ----------------------------------------------------------------------------------------------
# Flax layer semantics
# ----------------------------------------------------------------------------------------------

def _same_pad(n: int, k: int, s: int) -> Tuple[int, int]:
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def conv_1kk(x: torch.Tensor, kernel: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 1) -> torch.Tensor:
    """nnx.Conv with kernel (1, kh, kw, Cin, Cout), strides (1, s, s), SAME, on [B,F,H,W,Cin]."""
    B, Fr, H, W, Cin = x.shape
    _, kh, kw, ci, co = kernel.shape
    assert ci == Cin
    w = kernel[0].permute(3, 2, 0, 1)                       # (Cout, Cin, kh, kw); cross-correlation both sides
    xi = x.reshape(B * Fr, H, W, Cin).permute(0, 3, 1, 2)
    pt, pb = _same_pad(H, kh, stride)
    pl, pr = _same_pad(W, kw, stride)
    xi = F.pad(xi, (pl, pr, pt, pb))
    y = F.conv2d(xi, w, bias, stride=stride)
    return y.permute(0, 2, 3, 1).reshape(B, Fr, y.shape[2], y.shape[3], co)


def conv_pointwise(x: torch.Tensor, kernel: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """nnx.Conv(kernel_size=1): 1-D kernel (1, Cin, Cout); leading dims are batch."""
    y = x @ kernel[0]
    return y if bias is None else y + bias


def conv_transpose_144(x: torch.Tensor, kernel: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """nnx.ConvTranspose((1,4,4), strides (1,2,2), SAME, transpose_kernel=False) (utils.py:103-113).

    lhs-dilate by 2 (zeros between samples), pad 2/2, correlate with the unflipped kernel -> 2n.
    """
    B, Fr, H, W, C = x.shape
    _, kh, kw, ci, co = kernel.shape
    assert (kh, kw) == (4, 4) and ci == C
    xi = x.reshape(B * Fr, H, W, C).permute(0, 3, 1, 2)
    xd = xi.new_zeros(B * Fr, C, 2 * H - 1, 2 * W - 1)
    xd[:, :, ::2, ::2] = xi
    xd = F.pad(xd, (2, 2, 2, 2))
    w = kernel[0].permute(3, 2, 0, 1)
    y = F.conv2d(xd, w, bias)
    assert y.shape[2] == 2 * H and y.shape[3] == 2 * W
    return y.permute(0, 2, 3, 1).reshape(B, Fr, 2 * H, 2 * W, co)


def layer_norm(x: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    mean = x.mean(-1, keepdim=True)
    var = ((x * x).mean(-1, keepdim=True) - mean * mean).clamp_min(0.0)   # use_fast_variance
    return (x - mean) * torch.rsqrt(var + NORM_EPS) * scale + bias


def group_norm(x: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, groups: int) -> torch.Tensor:
    """nnx.GroupNorm on [B, ..., C]: statistics over every non-batch axis x channels-in-group."""
    B, C = x.shape[0], x.shape[-1]
    xg = x.reshape(B, -1, groups, C // groups)
    mean = xg.mean(dim=(1, 3), keepdim=True)
    var = ((xg * xg).mean(dim=(1, 3), keepdim=True) - mean * mean).clamp_min(0.0)
    y = ((xg - mean) * torch.rsqrt(var + NORM_EPS)).reshape(x.shape)
    return y * scale + bias


def silu(x):
    return x * torch.sigmoid(x)


def gelu_tanh(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


# ----------------------------------------------------------------------------------------------
# blocks (modules.py)
# ----------------------------------------------------------------------------------------------

def sinusoidal_pos_emb(t: torch.Tensor, dim: int, dtype) -> torch.Tensor:
    """modules.py:30-45."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half, dtype=dtype) * -e)
    arg = t.to(dtype)[..., None] * freqs[None, :]
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)


def block(p: Dict[str, torch.Tensor], prefix: str, x, scale_shift, groups: int):
    """modules.py:150-179: conv(1,3,3) -> GroupNorm -> x*(scale+1)+shift -> SiLU."""
    x = conv_1kk(x, p[f'{prefix}.proj.kernel'], p[f'{prefix}.proj.bias'])
    x = group_norm(x, p[f'{prefix}.norm.scale'], p[f'{prefix}.norm.bias'], groups)
    if scale_shift is not None:
        scale, shift = scale_shift
        x = x * (scale + 1) + shift
    return silu(x)


def resnet_block(p, prefix: str, x, temb, groups: int):
    """modules.py:182-243 (incl. quirk Q4: LayerNorm on the time-MLP output and on the residual)."""
    scale_shift = None
    if f'{prefix}.mlp.layers.1.kernel' in p:
        assert temb is not None, 'time emb must be passed in'
        te = silu(temb) @ p[f'{prefix}.mlp.layers.1.kernel'] + p[f'{prefix}.mlp.layers.1.bias']
        te = layer_norm(te, p[f'{prefix}.norm_1.scale'], p[f'{prefix}.norm_1.bias'])
        te = te[:, None, None, None, :]
        scale_shift = te.chunk(2, dim=-1)
    h = block(p, f'{prefix}.block_1', x, scale_shift, groups)
    h = block(p, f'{prefix}.block_2', h, None, groups)
    if f'{prefix}.res_conv.kernel' in p:
        r = conv_pointwise(x, p[f'{prefix}.res_conv.kernel'], p[f'{prefix}.res_conv.bias'])
    else:
        r = x
    return h + layer_norm(r, p[f'{prefix}.norm_2.scale'], p[f'{prefix}.norm_2.bias'])


def spatial_linear_attention(p, prefix: str, x, heads: int):
    """modules.py:94-129.  Quirks Q2/Q3: `q * scale` is dead; q softmax over D, k softmax over N."""
    B, Fr, H, W, C = x.shape
    xf = x.reshape(B * Fr, H * W, C)
    def proj(n):
        y = xf @ p[f'{prefix}.{n}.kernel'][0]                  # (BF, N, heads*D)
        return y.reshape(B * Fr, H * W, heads, -1).permute(0, 2, 3, 1)   # b h c (x y)
    q = torch.softmax(proj('q'), dim=-2)
    k = torch.softmax(proj('k'), dim=-1)
    v = proj('v')
    ctx = torch.einsum('bhdn,bhen->bhde', k, v)
    out = torch.einsum('bhde,bhdn->bhen', ctx, q)
    out = out.permute(0, 3, 1, 2).reshape(B * Fr, H * W, -1)   # b (x y) (h c)
    out = out @ p[f'{prefix}.to_out.kernel'][0]
    return out.reshape(B, Fr, H, W, C)


def multihead_attention(p, prefix: str, x, dim_head: int, focus_present_mask=None, pos_bias=None):
    """modules.py:280-326 stand-alone semantics (x is [..., L, C]); Q5 ordering kept.

    Inside Unet3D the PreNorm wrapper drops both kwargs (Q1), so they are None on the hot path.
    """
    def lin(n):
        return torch.einsum('...c,chd->...hd', x, p[f'{prefix}.{n}.kernel']) + p[f'{prefix}.{n}.bias']
    q, k, v = lin('q'), lin('k'), lin('v')
    L = x.shape[-2]
    def out_proj(a):
        return torch.einsum('...hd,hdc->...c', a, p[f'{prefix}.out.kernel']) + p[f'{prefix}.out.bias']
    if focus_present_mask is not None and bool(torch.all(focus_present_mask)):
        return out_proj(v)
    q = q / dim_head ** 0.5
    sim = torch.einsum('...ihd,...jhd->...hij', q, k)
    attn = torch.softmax(sim, dim=-1)
    if focus_present_mask is not None and bool(torch.any(focus_present_mask)):
        eye = torch.eye(L, dtype=torch.bool)
        m = torch.where(focus_present_mask.reshape(-1, 1, 1, 1, 1, 1), eye.reshape(1, 1, 1, 1, L, L),
                        torch.ones(1, 1, 1, 1, L, L, dtype=torch.bool))
        attn = torch.where(m, attn, torch.full_like(attn, torch.finfo(torch.float32).min))
    if pos_bias is not None:
        attn = attn + pos_bias
    o = torch.einsum('...hij,...jhd->...ihd', attn, v)
    return out_proj(o)


def temporal_attention(p, prefix: str, x, dim_head: int):
    """Residual(PreNorm(EinopsToAndFrom('b f h w c' -> 'b (h w) f c', MHA))) with PreNorm a no-op.

    unet3d.py:86-96,118-120; modules.py:146-148 (Q1): fn(x) + x, un-normalised, no bias, no mask.
    """
    B, Fr, H, W, C = x.shape
    xt = x.permute(0, 2, 3, 1, 4).reshape(B, H * W, Fr, C)
    o = multihead_attention(p, f'{prefix}.fn.fn.fn', xt, dim_head)
    o = o.reshape(B, H, W, Fr, C).permute(0, 3, 1, 2, 4)
    return o + x


def mid_spatial_attention(p, prefix: str, x, dim_head: int):
    """unet3d.py:196-205: MHA over (h w) tokens per frame, Residual + no-op PreNorm."""
    B, Fr, H, W, C = x.shape
    xs = x.reshape(B, Fr, H * W, C)
    o = multihead_attention(p, f'{prefix}.fn.fn.fn', xs, dim_head)
    return o.reshape(B, Fr, H, W, C) + x


def sla_residual(p, prefix: str, x, heads: int):
    return spatial_linear_attention(p, f'{prefix}.fn.fn', x, heads) + x


def relative_position_bias(p, n: int, heads: int) -> torch.Tensor:
    """modules.py:350-390 (dead on the Unet3D path; kept for API completeness, Q9).

    `_relative_position_bucket` is always called with its defaults (32 buckets, max_distance 128).
    """
    emb = p['time_rel_pos_bias.relative_attention_bias.embedding']
    qpos = torch.arange(n)[:, None]
    kpos = torch.arange(n)[None, :]
    rel = qpos - kpos
    num_buckets, max_distance = 32, 128
    nn_ = -rel
    num_buckets //= 2
    ret = (nn_ < 0).to(torch.int64) * num_buckets
    nn_ = nn_.abs()
    max_exact = num_buckets // 2
    is_small = nn_ < max_exact
    safe = nn_.clamp_min(1).to(torch.float32)
    val_if_large = max_exact + (torch.log(safe / max_exact) / math.log(max_distance / max_exact)
                                * (num_buckets - max_exact)).to(torch.int64)
    val_if_large = torch.minimum(val_if_large, torch.full_like(val_if_large, num_buckets - 1))
    ret = ret + torch.where(is_small, nn_, val_if_large)
    return emb[ret].permute(2, 0, 1)


# ----------------------------------------------------------------------------------------------
# the network (unet3d.py:262-387)
# ----------------------------------------------------------------------------------------------

def unet_forward(p: Dict[str, torch.Tensor], cfg: UnetConfig, x: torch.Tensor, time: torch.Tensor,
                 cond: Optional[torch.Tensor] = None, null_cond_prob: float = 0.0,
                 cond_mask: Optional[torch.Tensor] = None,
                 taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """x: [B,C,F,H,W]; time: [B] int; returns channel-LAST [B,F,H,W,out_dim] (unet3d.py:387).

    `cond_mask` ([B] bool) overrides the Bernoulli(null_cond_prob) mask for 0<p<1 (Q14).
    `taps`, if given, receives named intermediates for block-level parity tests.
    """
    assert not (cfg.has_cond and cond is None), 'cond must be passed in if cond_dim specified'
    dt = x.dtype
    H, Dh, G = cfg.attn_heads, cfg.attn_dim_head, cfg.resnet_groups
    def tap(n, v):
        if taps is not None:
            taps[n] = v
    x = x.permute(0, 2, 3, 4, 1)                                            # b c f h w -> b f h w c
    x = conv_1kk(x, p['init_conv.kernel'], p['init_conv.bias'])
    tap('init_conv', x)
    x = temporal_attention(p, 'init_temporal_attn', x, Dh)
    tap('init_temporal_attn', x)
    r = x
    t = sinusoidal_pos_emb(time, cfg.dim, dt)
    t = t @ p['time_mlp.layers.1.kernel'] + p['time_mlp.layers.1.bias']
    t = gelu_tanh(t)
    t = t @ p['time_mlp.layers.3.kernel'] + p['time_mlp.layers.3.bias']
    if cfg.has_cond:
        B = x.shape[0]
        if cond_mask is None:
            if null_cond_prob == 1:
                cond_mask = torch.ones(B, dtype=torch.bool)
            elif null_cond_prob == 0:
                cond_mask = torch.zeros(B, dtype=torch.bool)
            else:
                raise ValueError('pass cond_mask explicitly for 0 < null_cond_prob < 1')
        c = torch.where(cond_mask[:, None], p['null_cond_emb'].to(dt), cond.to(dt))
        t = torch.cat((t, c), dim=-1)
    tap('temb', t)
    hs = []
    n_res = len(cfg.in_out)
    for i in range(n_res):
        x = resnet_block(p, f'downs.{i}.0', x, t, G); tap(f'downs.{i}.0', x)
        x = resnet_block(p, f'downs.{i}.1', x, t, G); tap(f'downs.{i}.1', x)
        if cfg.use_sparse_linear_attn:
            x = sla_residual(p, f'downs.{i}.2', x, cfg.sla_heads); tap(f'downs.{i}.2', x)
        x = temporal_attention(p, f'downs.{i}.3', x, Dh); tap(f'downs.{i}.3', x)
        hs.append(x)
        if i < n_res - 1:
            x = conv_1kk(x, p[f'downs.{i}.4.kernel'], p[f'downs.{i}.4.bias'], stride=2); tap(f'downs.{i}.4', x)
    x = resnet_block(p, 'mid_block1', x, t, G); tap('mid_block1', x)
    x = mid_spatial_attention(p, 'mid_spatial_attn', x, Dh); tap('mid_spatial_attn', x)
    x = temporal_attention(p, 'mid_temporal_attn', x, Dh); tap('mid_temporal_attn', x)
    x = resnet_block(p, 'mid_block2', x, t, G); tap('mid_block2', x)
    for i in range(n_res):
        x = torch.cat((x, hs.pop()), dim=-1)
        x = resnet_block(p, f'ups.{i}.0', x, t, G); tap(f'ups.{i}.0', x)
        x = resnet_block(p, f'ups.{i}.1', x, t, G); tap(f'ups.{i}.1', x)
        if cfg.use_sparse_linear_attn:
            x = sla_residual(p, f'ups.{i}.2', x, cfg.sla_heads); tap(f'ups.{i}.2', x)
        x = temporal_attention(p, f'ups.{i}.3', x, Dh); tap(f'ups.{i}.3', x)
        if i < n_res - 1:
            x = conv_transpose_144(x, p[f'ups.{i}.4.kernel'], p[f'ups.{i}.4.bias']); tap(f'ups.{i}.4', x)
    x = torch.cat((x, r), dim=-1)
    x = resnet_block(p, 'final_conv.layers.0', x, None, G); tap('final_conv.layers.0', x)
    out = conv_pointwise(x, p['final_conv.layers.1.kernel'], p['final_conv.layers.1.bias'])
    return out


def forward_with_cond_scale(p, cfg: UnetConfig, x, time, cond=None, cond_scale: float = 2.0):
    """unet3d.py:254-260."""
    logits = unet_forward(p, cfg, x, time, cond=cond, null_cond_prob=0.0)
    if cond_scale == 1 or not cfg.has_cond:
        return logits
    null_logits = unet_forward(p, cfg, x, time, cond=cond, null_cond_prob=1.0)
    return null_logits + (logits - null_logits) * cond_scale


def random_params(cfg: UnetConfig, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Random test parameters that exercise EVERY term (non-zero biases, non-unit norm scales).

    Not the product initialiser: kernels ~ N(0, 1/fan_in), biases ~ 0.1 N(0,1), scales ~ 1 + 0.1 N.
    """
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, shape in param_spec(cfg):
        leaf = name.rsplit('.', 1)[-1]
        if leaf == 'kernel':
            if name.endswith('out.kernel') and len(shape) == 3 and 'fn.fn.fn' in name:
                fan_in = shape[0] * shape[1]
            elif len(shape) == 3 and 'fn.fn.fn' in name:
                fan_in = shape[0]
            else:
                fan_in = 1
                for d in shape[:-1]:
                    fan_in *= d
            v = torch.randn(shape, generator=g, dtype=torch.float64) / math.sqrt(fan_in)
        elif leaf == 'scale':
            v = 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float64)
        elif leaf == 'embedding':
            v = torch.randn(shape, generator=g, dtype=torch.float64) / math.sqrt(shape[-1])
        elif name == 'null_cond_emb':
            v = torch.randn(shape, generator=g, dtype=torch.float64)
        else:
            v = 0.1 * torch.randn(shape, generator=g, dtype=torch.float64)
        p[name] = v.to(dtype)
    return p
