"""Shape known-answers from the reference tests + quirk invariants for
oracle/unet3d_ref.py.  Numeric values are 'parity unpinned' against JAX (see the oracle header)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet3d_ref as R


@pytest.fixture(scope='module')
def tiny():
    cfg = R.UnetConfig(dim=16, channels=3, cond_dim=32)
    return cfg, R.random_params(cfg, seed=5)


def _inputs(seed=0, cond=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, 3, 4, 16, 16, generator=g)
    t = torch.randint(0, 1000, (1,), generator=g)
    c = torch.randn(1, 32, generator=g) if cond else None
    return x, t, c


def test_unet_output_shape_with_cond(tiny):        # /root/reference/test_unet3d.py:12-38
    cfg, p = tiny
    x, t, c = _inputs()
    y = R.unet_forward(p, cfg, x, t, cond=c)
    assert y.shape == (1, 4, 16, 16, 3) and y.dtype == x.dtype


def test_unet_output_shape_without_cond():         # /root/reference/test_unet3d.py:40-60
    cfg = R.UnetConfig(dim=16, channels=3)
    y = R.unet_forward(R.random_params(cfg, 1), cfg, *_inputs(cond=False)[:2])
    assert y.shape == (1, 4, 16, 16, 3)


def test_cond_required(tiny):                       # unet3d.py:271-273
    cfg, p = tiny
    x, t, _ = _inputs()
    with pytest.raises(AssertionError):
        R.unet_forward(p, cfg, x, t)


def test_param_counts_match_survey():
    n = lambda cfg: sum(int(np.prod(s)) for _, s in R.param_spec(cfg))
    assert n(R.UnetConfig(dim=64, channels=1)) == 35745665
    assert n(R.UnetConfig(dim=32, channels=1)) == 9993409


def test_module_shapes():                           # /root/reference/test_modules.py:54-69,100-118,273-293
    assert R.sinusoidal_pos_emb(torch.arange(4), 32, torch.float32).shape == (4, 32)
    cfg = R.UnetConfig(dim=16, channels=3)
    p = R.random_params(cfg, 0)
    assert R.relative_position_bias(p, 10, 8).shape == (8, 10, 10)
    x = torch.randn(2, 3, 4, 5, 16)
    assert R.spatial_linear_attention(p, 'downs.0.2.fn.fn', x, 8).shape == x.shape


def test_q1_prenorm_and_posbias_are_dead(tiny):
    """Q1/Q9: output is invariant to every PreNorm gamma/beta and to the rel-pos embedding."""
    cfg, p = tiny
    x, t, c = _inputs(2)
    y0 = R.unet_forward(p, cfg, x, t, cond=c)
    p2 = dict(p)
    for k in p:
        if '.fn.norm.' in k or k.startswith('time_rel_pos_bias'):
            p2[k] = p[k] + 3.0
    assert torch.equal(R.unet_forward(p2, cfg, x, t, cond=c), y0)


def test_q4_final_block_norm1_is_dead(tiny):
    cfg, p = tiny
    x, t, c = _inputs(3)
    p2 = dict(p)
    p2['final_conv.layers.0.norm_1.scale'] = p['final_conv.layers.0.norm_1.scale'] * 7
    assert torch.equal(R.unet_forward(p2, cfg, x, t, cond=c), R.unet_forward(p, cfg, x, t, cond=c))


def test_q2_sla_has_no_scale():
    """Q2/Q3: with q logits all equal, softmax over D gives exactly 1/D (no D^-0.5 factor)."""
    heads, D, C = 2, 4, 8
    p = {'a.q.kernel': torch.zeros(1, C, heads * D), 'a.k.kernel': torch.zeros(1, C, heads * D),
         'a.v.kernel': torch.ones(1, C, heads * D), 'a.to_out.kernel': torch.eye(heads * D, C)[None]}
    x = torch.ones(1, 1, 2, 2, C)
    y = R.spatial_linear_attention(p, 'a', x, heads)
    # k softmax over N=4 -> 1/4 each, v = C=8 everywhere: ctx[d,e] = 8 ; q = 1/D: out[e] = sum_d 8/D = 8
    np.testing.assert_allclose(y, torch.full_like(y, 8.0), rtol=1e-6)


def test_conv_transpose_equals_torch_flipped():
    """App. B.1: dilate+pad(2,2)+unflipped correlate == conv_transpose2d with the flipped kernel, pad 1."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 5, 6, 4, generator=g)
    k = torch.randn(1, 4, 4, 4, 7, generator=g)
    b = torch.randn(7, generator=g)
    y = R.conv_transpose_144(x, k, b)
    xi = x.reshape(6, 5, 6, 4).permute(0, 3, 1, 2)
    wt = k[0].flip(0, 1).permute(2, 3, 0, 1)          # (Cin, Cout, kh, kw)
    y2 = F.conv_transpose2d(xi, wt, b, stride=2, padding=1).permute(0, 2, 3, 1).reshape(2, 3, 10, 12, 7)
    np.testing.assert_allclose(y, y2, atol=1e-5)


def test_downsample_same_padding():
    x = torch.randn(1, 2, 8, 8, 4)
    k = torch.randn(1, 4, 4, 4, 4)
    assert R.conv_1kk(x, k, None, stride=2).shape == (1, 2, 4, 4, 4)
    assert R._same_pad(8, 4, 2) == (1, 1) and R._same_pad(7, 3, 1) == (1, 1)


def test_group_norm_matches_torch():
    x = torch.randn(2, 3, 4, 5, 16)
    s, b = torch.randn(16), torch.randn(16)
    y = R.group_norm(x, s, b, 8)
    y2 = F.group_norm(x.permute(0, 4, 1, 2, 3), 8, s, b, eps=1e-6).permute(0, 2, 3, 4, 1)
    np.testing.assert_allclose(y, y2, atol=2e-5)


def test_cfg_combination(tiny):                     # unet3d.py:254-260
    cfg, p = tiny
    x, t, c = _inputs(4)
    a = R.unet_forward(p, cfg, x, t, cond=c, null_cond_prob=0.0)
    n = R.unet_forward(p, cfg, x, t, cond=c, null_cond_prob=1.0)
    np.testing.assert_allclose(R.forward_with_cond_scale(p, cfg, x, t, cond=c, cond_scale=2.0), n + (a - n) * 2.0, atol=1e-6)
    assert torch.equal(R.forward_with_cond_scale(p, cfg, x, t, cond=c, cond_scale=1.0), a)


@pytest.mark.parametrize('tag', ['tiny_cond', 'tiny_nocond'])
def test_golden_fixture(tag):
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', f'unet_{tag}.npz'))
    cfg = R.UnetConfig(dim=16, channels=3, cond_dim=32 if tag == 'tiny_cond' else None)
    p = {k: v.float() for k, v in R.random_params(cfg, seed=int(z['weight_seed']), dtype=torch.float64).items()}
    cond = torch.from_numpy(z['cond']) if tag == 'tiny_cond' else None
    y = R.unet_forward(p, cfg, torch.from_numpy(z['x']), torch.from_numpy(z['t']), cond=cond)
    np.testing.assert_allclose(y.numpy(), z['eps_fp32'], atol=1e-5)
    rel = np.linalg.norm(z['eps_fp32'] - z['eps_fp64']) / np.linalg.norm(z['eps_fp64'])
    assert rel < 1e-5      # fp32 restatement vs fp64 truth: the stated fp32 tolerance band
