"""GPU parity of the conv backward building blocks vs torch autograd through oracle/unet3d_ref.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet3d_ref as R

DEV = 'cuda:0'


def _rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


CASES = [
    # B, F, H, W, Cin, Cout, k, stride, kind
    (1, 4, 16, 16, 16, 32, 3, 1, 0),
    (2, 2, 8, 8, 64, 64, 3, 1, 0),
    (1, 3, 6, 10, 8, 24, 3, 1, 0),
    (1, 2, 4, 4, 128, 64, 3, 1, 0),
    (1, 4, 16, 16, 32, 32, 4, 2, 0),      # Downsample
    (1, 4, 8, 8, 32, 32, 4, 1, 1),        # Upsample
    (1, 2, 2, 2, 64, 64, 4, 1, 1),
    (1, 4, 8, 8, 48, 40, 1, 1, 0),        # pointwise
    (1, 3, 5, 7, 72, 256, 1, 1, 0),       # pointwise, Cout % 256 == 0: split-K GEMM form of bf16 mode (ragged rows and channels)
    (2, 4, 8, 8, 64, 768, 1, 1, 0),       # the q|k|v projection shape
]


def _fwd(x, kern, bias, k, stride, kind):
    if kind == 1:
        return R.conv_transpose_144(x, kern, bias)
    if k == 1:
        return R.conv_pointwise(x, kern[0], bias)
    return R.conv_1kk(x, kern, bias, stride=stride)


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('case', CASES)
def test_conv_dgrad_wgrad_bias(mode, case):
    from video_diffusion_nnx_amd import ops
    B, Fr, H, W, Cin, Cout, k, stride, kind = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Fr, H, W, Cin, generator=g, dtype=torch.float64, requires_grad=True)
    kern = (torch.randn(1, k, k, Cin, Cout, generator=g, dtype=torch.float64) / (k * k * Cin) ** 0.5).requires_grad_(True)
    bias = torch.randn(Cout, generator=g, dtype=torch.float64, requires_grad=True)
    y = _fwd(x, kern, bias, k, stride, kind)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    gx, gk, gb = torch.autograd.grad(y, (x, kern, bias), dy)
    dyd = dy.float().to(DEV)
    # data gradient = forward kernel with the transposed/flipped packing (and Down <-> Up swapped)
    pwt = ops.pack_conv_weights_t(kern.detach().float().to(DEV), mode)
    if kind == 1:
        dx = ops.conv_forward(dyd, pwt, Cin, mode=mode, k=4, stride=2)
    elif stride == 2:
        dx = ops.conv_forward(dyd, pwt, Cin, mode=mode, kind=1)
    else:
        dx = ops.conv_forward(dyd, pwt, Cin, mode=mode, k=k)
    tol = 2e-6 if mode == 'f32' else 8e-3
    assert dx.shape == gx.shape
    assert _rel(dx.cpu().double(), gx) < tol, (mode, case, _rel(dx.cpu().double(), gx))
    dw = ops.conv_backward_weights(x.detach().float().to(DEV), dyd, kern.shape, kind=kind, k=k, stride=stride)
    assert _rel(dw.cpu().double(), gk) < 5e-6, (case, _rel(dw.cpu().double(), gk))      # exact-f32 MFMA form
    if mode == 'bf16':     # the form a bf16-mode handle's backward uses: operands rounded to bf16 (transposing LDS reads), fp32 accumulate
        dw16 = ops.conv_backward_weights(x.detach().float().to(DEV), dyd, kern.shape, kind=kind, k=k, stride=stride, bf16_operands=True)
        assert _rel(dw16.cpu().double(), gk) < 8e-3, (case, _rel(dw16.cpu().double(), gk))
    db = ops.colsum(dyd)
    assert _rel(db.cpu().double(), gb) < 5e-6


def test_wgrad_concat_and_prologue():
    from video_diffusion_nnx_amd import ops
    g = torch.Generator().manual_seed(3)
    B, Fr, H, W, C0, C1, Cout = 2, 2, 8, 8, 32, 16, 32
    xa = torch.randn(B, Fr, H, W, C0, generator=g, dtype=torch.float64)
    xb = torch.randn(B, Fr, H, W, C1, generator=g, dtype=torch.float64)
    kern = (torch.randn(1, 3, 3, C0 + C1, Cout, generator=g, dtype=torch.float64) / 20).requires_grad_(True)
    y = R.conv_1kk(torch.cat((xa, xb), -1), kern, None)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (gk,) = torch.autograd.grad(y, kern, dy)
    dw = ops.conv_backward_weights(xa.float().to(DEV), dy.float().to(DEV), kern.shape, x1=xb.float().to(DEV))
    assert _rel(dw.cpu().double(), gk) < 5e-6
    # prologue: Xhat = SiLU(GN(y1) * (s+1) + sh) recomputed inside the wgrad kernel
    y1 = torch.randn(B, Fr, H, W, Cout, generator=g, dtype=torch.float64)
    gamma, beta = 1 + 0.1 * torch.randn(Cout, generator=g, dtype=torch.float64), 0.1 * torch.randn(Cout, generator=g, dtype=torch.float64)
    ss = 0.3 * torch.randn(B, 2 * Cout, generator=g, dtype=torch.float64)
    k2 = (torch.randn(1, 3, 3, Cout, Cout, generator=g, dtype=torch.float64) / 17).requires_grad_(True)
    h = R.group_norm(y1, gamma, beta, 8) * (ss[:, None, None, None, :Cout] + 1) + ss[:, None, None, None, Cout:]
    y2 = R.conv_1kk(R.silu(h), k2, None)
    dy2 = torch.randn(y2.shape, generator=g, dtype=torch.float64)
    (gk2,) = torch.autograd.grad(y2, k2, dy2)
    yg = y1.reshape(B, -1, 8, Cout // 8)
    stats = torch.zeros(B, 32, 8, 2, dtype=torch.float64)
    stats[:, 0, :, 0] = yg.sum(dim=(1, 3)); stats[:, 0, :, 1] = (yg * yg).sum(dim=(1, 3))
    dw2 = ops.conv_backward_weights(y1.float().to(DEV), dy2.float().to(DEV), k2.shape, in_stats=stats.reshape(-1).to(DEV),
                                    gamma=gamma.float().to(DEV), beta=beta.float().to(DEV), scale_shift=ss.float().to(DEV))
    assert _rel(dw2.cpu().double(), gk2) < 2e-5
    # bf16-operand form of both
    dw16 = ops.conv_backward_weights(xa.float().to(DEV), dy.float().to(DEV), kern.shape, x1=xb.float().to(DEV), bf16_operands=True)
    assert _rel(dw16.cpu().double(), gk) < 8e-3
    dw216 = ops.conv_backward_weights(y1.float().to(DEV), dy2.float().to(DEV), k2.shape, in_stats=stats.reshape(-1).to(DEV),
                                      gamma=gamma.float().to(DEV), beta=beta.float().to(DEV), scale_shift=ss.float().to(DEV), bf16_operands=True)
    assert _rel(dw216.cpu().double(), gk2) < 8e-3
    # pointwise conv over a concat input, wide output (the split-K GEMM form of bf16 mode)
    kp = (torch.randn(1, 1, 1, C0 + C1, 256, generator=g, dtype=torch.float64) / 7).requires_grad_(True)
    yp = R.conv_pointwise(torch.cat((xa, xb), -1), kp[0], None)
    dyp = torch.randn(yp.shape, generator=g, dtype=torch.float64)
    (gkp,) = torch.autograd.grad(yp, kp, dyp)
    dwp = ops.conv_backward_weights(xa.float().to(DEV), dyp.float().to(DEV), kp.shape, x1=xb.float().to(DEV), k=1, bf16_operands=True)
    assert _rel(dwp.cpu().double(), gkp) < 8e-3


@pytest.mark.parametrize('C,B,shape,use_ss,tail', [(16, 2, (3, 5, 5), True, False), (64, 2, (4, 8, 8), True, False), (64, 1, (2, 8, 8), False, True),
                                                   (256, 2, (2, 4, 4), False, True), (512, 1, (2, 2, 2), True, False), (1024, 1, (2, 2, 2), False, True),
                                                   (24, 1, (2, 3, 3), False, True)])
def test_norm_act_backward(C, B, shape, use_ss, tail):
    from video_diffusion_nnx_amd import ops
    g = torch.Generator().manual_seed(C + B)
    D = torch.float64
    y = (torch.randn(B, *shape, C, generator=g, dtype=D) * 1.5 + 0.3).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g, dtype=D)).requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=g, dtype=D)).requires_grad_(True)
    ss = (0.3 * torch.randn(B, 2 * C, generator=g, dtype=D)).requires_grad_(True) if use_ss else None
    r = torch.randn(B, *shape, C, generator=g, dtype=D).requires_grad_(True) if tail else None
    lg = (1 + 0.2 * torch.randn(C, generator=g, dtype=D)).requires_grad_(True) if tail else None
    lb = (0.1 * torch.randn(C, generator=g, dtype=D)).requires_grad_(True) if tail else None
    h = R.group_norm(y, gamma, beta, 8)
    if use_ss:
        h = h * (ss[:, None, None, None, :C] + 1) + ss[:, None, None, None, C:]
    out = R.silu(h)
    if tail:
        out = out + R.layer_norm(r, lg, lb)
    dout = torch.randn(out.shape, generator=g, dtype=D)
    wrt = [y, gamma, beta] + ([ss] if use_ss else []) + ([r, lg, lb] if tail else [])
    grads = torch.autograd.grad(out, wrt, dout)
    gy, ggam, gbet = grads[:3]
    yg = y.detach().reshape(B, -1, 8, C // 8)
    stats = torch.zeros(B, 32, 8, 2, dtype=D)
    stats[:, 0, :, 0] = yg.sum(dim=(1, 3)); stats[:, 0, :, 1] = (yg * yg).sum(dim=(1, 3))
    f = lambda t: None if t is None else t.detach().float().to(DEV)
    res = ops.norm_act_backward(f(dout), f(y), stats.reshape(-1).to(DEV), f(gamma), f(beta), 8, scale_shift=f(ss), r=f(r), ln_gamma=f(lg))
    assert _rel(res['dy'].cpu().double(), gy) < 3e-5, _rel(res['dy'].cpu().double(), gy)
    assert _rel(res['d_gamma'].cpu().double(), ggam) < 3e-5 and _rel(res['d_beta'].cpu().double(), gbet) < 3e-5
    i = 3
    if use_ss:
        assert _rel(res['dss'].cpu().double(), grads[i]) < 3e-5
        i += 1
    if tail:
        assert _rel(res['dr'].cpu().double(), grads[i]) < 3e-5
        assert _rel(res['d_ln_gamma'].cpu().double(), grads[i + 1]) < 3e-5 and _rel(res['d_ln_beta'].cpu().double(), grads[i + 2]) < 3e-5


@pytest.mark.parametrize('B,Fr,H,W,heads,temporal', [(1, 16, 4, 4, 8, True), (2, 10, 2, 3, 8, True), (1, 2, 8, 8, 8, False), (1, 3, 5, 5, 4, False)])
def test_attention_core_backward(B, Fr, H, W, heads, temporal):
    from video_diffusion_nnx_amd import ops
    g = torch.Generator().manual_seed(B + Fr + H)
    D = torch.float64
    HD = heads * 32
    npix = B * Fr * H * W
    qkv = torch.randn(npix, 3 * HD, generator=g, dtype=D).requires_grad_(True)
    d_o = torch.randn(npix, HD, generator=g, dtype=D)
    x = qkv.reshape(B, Fr, H * W, 3, heads, 32)
    if temporal:
        seq = x.permute(0, 2, 1, 3, 4, 5)            # b (hw) f ...
    else:
        seq = x                                       # b f (hw) ...
    q, k, v = seq[..., 0, :, :] / 32 ** 0.5, seq[..., 1, :, :], seq[..., 2, :, :]
    sim = torch.einsum('...ihd,...jhd->...hij', q, k)
    o = torch.einsum('...hij,...jhd->...ihd', torch.softmax(sim, -1), v)
    o_rows = (o.permute(0, 2, 1, 3, 4) if temporal else o).reshape(npix, HD)
    (gqkv,) = torch.autograd.grad(o_rows, qkv, d_o)
    og, dq, dk, dv = ops.attention_core_backward(qkv.detach().float().to(DEV), d_o.float().to(DEV), B, Fr, H, W, heads, temporal)
    assert _rel(og.cpu().double(), o_rows.detach()) < 1e-5
    got = torch.cat([dq, dk, dv], -1).cpu().double()
    assert _rel(got, gqkv) < 2e-5, _rel(got, gqkv)
    # bf16-operand MFMA form (what a bf16-mode handle's backward runs for <= 16 tokens; longer sequences fall back to fp32)
    og, dq, dk, dv = ops.attention_core_backward(qkv.detach().float().to(DEV), d_o.float().to(DEV), B, Fr, H, W, heads, temporal, bf16_operands=True)
    tol = 1.5e-2 if (Fr if temporal else H * W) <= 16 else 2e-5
    assert _rel(og.cpu().double(), o_rows.detach()) < tol
    assert _rel(torch.cat([dq, dk, dv], -1).cpu().double(), gqkv) < tol


# y = MHA(x) + x over the frames of every pixel (modules.py:271-327 under Residual); the fused kernel returns dx and the two tensors
# the weight-gradient kernels read (o = attention output before the out projection, d(q|k|v)).  bf16 operands: 1.5e-2 (as the unfused
# bf16 core above); 16 frames = full tiles, 5 frames = masked keys / zero rows, odd pixel counts = ragged last workgroup.
@pytest.mark.parametrize('B,Fr,H,W', [(1, 16, 4, 4), (2, 5, 3, 3), (1, 16, 16, 16)])
def test_temporal_attention_backward_fused(B, Fr, H, W):
    from video_diffusion_nnx_amd import ops
    g = torch.Generator().manual_seed(100 + B + Fr + H)
    D = torch.float64
    x = torch.randn(B, Fr, H, W, 64, generator=g, dtype=D).requires_grad_(True)
    dy = torch.randn(B, Fr, H, W, 64, generator=g, dtype=D)
    wqkv = (torch.randn(64, 768, generator=g, dtype=D) * 0.15).requires_grad_(True)
    bqkv = (torch.randn(768, generator=g, dtype=D) * 0.1).requires_grad_(True)
    wo = (torch.randn(256, 64, generator=g, dtype=D) * 0.1).requires_grad_(True)
    rows = x.reshape(-1, 64)
    qkv = rows @ wqkv + bqkv                                       # [rows][q | k | v], head-major inside each third
    s = qkv.reshape(B, Fr, H * W, 3, 8, 32).permute(0, 2, 1, 3, 4, 5)       # b (hw) f part head d
    q, k, v = s[..., 0, :, :] / 32 ** 0.5, s[..., 1, :, :], s[..., 2, :, :]
    att = torch.softmax(torch.einsum('...ihd,...jhd->...hij', q, k), -1)
    o = torch.einsum('...hij,...jhd->...ihd', att, v).permute(0, 2, 1, 3, 4).reshape(-1, 256)
    y = (o @ wo).reshape(x.shape) + x
    o.retain_grad(); qkv.retain_grad()
    y.backward(dy)
    dx, og, dqkv = ops.temporal_attention_backward_fused(x.detach().float().to(DEV), dy.float().to(DEV), wqkv.detach().float().to(DEV),
                                                         bqkv.detach().float().to(DEV), wo.detach().float().to(DEV))
    torch.cuda.synchronize()
    assert _rel(og.float().cpu().double(), o.detach()) < 1.5e-2
    assert _rel(dqkv.float().cpu().double(), qkv.grad) < 1.5e-2, _rel(dqkv.float().cpu().double(), qkv.grad)
    assert _rel(dx.cpu().double(), x.grad) < 1.5e-2, _rel(dx.cpu().double(), x.grad)
    # the attention branch alone (dx - dy), so that the residual does not mask an error
    assert _rel(dx.cpu().double() - dy, x.grad - dy) < 2.5e-2


@pytest.mark.parametrize('NF,H,W', [(2, 8, 8), (3, 5, 7), (1, 20, 20)])
def test_sla_core_backward(NF, H, W):
    from video_diffusion_nnx_amd import ops
    g = torch.Generator().manual_seed(NF + H)
    D = torch.float64
    N = H * W
    q, k, v = [(2 * torch.randn(NF * N, 256, generator=g, dtype=D)).requires_grad_(True) for _ in range(3)]
    d_out = torch.randn(NF * N, 256, generator=g, dtype=D)
    def hs(t):
        return t.reshape(NF, N, 8, 32).permute(0, 2, 3, 1)      # b h c n
    qs, ks = torch.softmax(hs(q), -2), torch.softmax(hs(k), -1)
    ctx = torch.einsum('bhdn,bhen->bhde', ks, hs(v))
    out = torch.einsum('bhde,bhdn->bhen', ctx, qs).permute(0, 3, 1, 2).reshape(NF * N, 256)
    gq, gk, gv = torch.autograd.grad(out, (q, k, v), d_out)
    f = lambda t: t.detach().float().to(DEV)
    o, dq, dk, dv = ops.sla_core_backward(f(q), f(k), f(v), f(d_out), NF, N)
    assert _rel(o.cpu().double(), out.detach()) < 1e-5
    for got, ref, nm in ((dq, gq, 'dq'), (dk, gk, 'dk'), (dv, gv, 'dv')):
        assert _rel(got.cpu().double(), ref) < 3e-5, (nm, _rel(got.cpu().double(), ref))
    # bf16-operand reductions (pass A on MFMA; pass B unchanged): what a bf16-mode handle's backward uses
    o, dq, dk, dv = ops.sla_core_backward(f(q), f(k), f(v), f(d_out), NF, N, bf16_operands=True)
    assert _rel(o.cpu().double(), out.detach()) < 1e-2
    for got, ref, nm in ((dq, gq, 'dq'), (dk, gk, 'dk'), (dv, gv, 'dv')):
        assert _rel(got.cpu().double(), ref) < 2e-2, ('bf16', nm, _rel(got.cpu().double(), ref))
