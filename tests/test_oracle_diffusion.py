"""Pins oracle/diffusion_ref.py with the reference's own known answers (values lifted from
/root/reference/gaussian_diffusion_test.py:75-218 and /root/reference/utils_test.py:102-131)."""
import numpy as np
import pytest
import torch

from oracle.diffusion_ref import DiffusionRef, TABLE_NAMES, cosine_beta_schedule, extract, schedule

B, C, F, S, T = 2, 3, 2, 8, 10          # gaussian_diffusion_test.py:40-45


def zeros_denoise(x, t):                 # MockDenoiseFn (gaussian_diffusion_test.py:18-33)
    b, c, f, h, w = x.shape
    return torch.zeros(b, f, h, w, c, dtype=x.dtype)


@pytest.fixture
def diff():
    return DiffusionRef(zeros_denoise, image_size=S, num_frames=F, channels=C, timesteps=T, loss_type='l1')


@pytest.fixture
def x_start():
    return torch.ones(B, C, F, S, S)


def test_table_shapes(diff):             # gaussian_diffusion_test.py:75-86
    for n in TABLE_NAMES:
        assert diff.tab[n].shape == (T,)
        assert diff.tab[n].dtype == torch.float32


def test_extract_known_answer():         # utils_test.py:102-110
    out = extract(torch.arange(10), torch.tensor([1, 3, 5]), (3, 10, 10, 10))
    assert out.shape == (3, 1, 1, 1)
    assert out.flatten().tolist() == [1, 3, 5]


def test_cosine_betas_range():           # utils_test.py:112-117
    b = cosine_beta_schedule(100)
    assert b.shape == (100,) and (b >= 0).all() and (b <= 1).all()


def test_q_mean_variance_t0(diff, x_start):      # gaussian_diffusion_test.py:88-109
    t = torch.tensor([0, T // 2])
    mean, var, logvar = diff.q_mean_variance(x_start, t)
    assert mean.shape == x_start.shape and var.shape == (B, 1, 1, 1, 1) and logvar.shape == (B, 1, 1, 1, 1)
    t0 = torch.zeros(B, dtype=torch.int64)
    mean0, var0, _ = diff.q_mean_variance(x_start, t0)
    np.testing.assert_allclose(mean0, diff.tab['sqrt_alphas_cumprod'][0] * x_start, atol=1e-6)
    np.testing.assert_allclose(var0, torch.full((B, 1, 1, 1, 1), 1.0 - diff.tab['alphas_cumprod'][0].item()), atol=1e-6)


def test_predict_start_inverts_q_sample(diff, x_start):   # gaussian_diffusion_test.py:111-123
    t = torch.full((B,), T // 2)
    noise = torch.zeros_like(x_start)
    x_t = diff.q_sample(x_start, t, noise)
    np.testing.assert_allclose(diff.predict_start_from_noise(x_t, t, noise), x_start, atol=1e-4)


def test_q_posterior_shapes(diff, x_start):      # gaussian_diffusion_test.py:125-133
    t = torch.full((B,), T // 2)
    x_t = diff.q_sample(x_start, t, torch.zeros_like(x_start))
    mean, var, logvar = diff.q_posterior(x_start, x_t, t)
    assert mean.shape == x_start.shape and var.shape == (B, 1, 1, 1, 1) and logvar.shape == (B, 1, 1, 1, 1)


def test_q_sample_t0_closed_form(diff, x_start):  # gaussian_diffusion_test.py:135-158
    noise = torch.randn(x_start.shape, generator=torch.Generator().manual_seed(42))
    t0 = torch.zeros(B, dtype=torch.int64)
    exp = diff.tab['sqrt_alphas_cumprod'][0] * x_start + diff.tab['sqrt_one_minus_alphas_cumprod'][0] * noise
    np.testing.assert_allclose(diff.q_sample(x_start, t0, noise), exp, atol=1e-6)


def test_p_sample_t0_is_mean(diff, x_start):      # gaussian_diffusion_test.py:175-189
    x_t = torch.zeros_like(x_start)
    t0 = torch.zeros(B, dtype=torch.int64)
    mean0, _, _ = diff.p_mean_variance(x_t, t0, clip_denoised=False)
    noise = torch.randn(x_start.shape, generator=torch.Generator().manual_seed(1))
    np.testing.assert_allclose(diff.p_sample(x_t, t0, noise), mean0, atol=1e-5)
    tm = torch.full((B,), T // 2)
    assert diff.p_sample(x_t, tm, noise).shape == x_t.shape


def test_losses_known_answers(diff, x_start):     # gaussian_diffusion_test.py:191-210
    t = torch.tensor([0, T // 2])
    assert abs(diff.p_losses(x_start, t, torch.zeros_like(x_start)).item()) < 1e-6
    half = torch.full_like(x_start, 0.5)
    assert abs(diff.p_losses(x_start, t, half).item() - 0.5) < 1e-6
    diff.loss_type = 'l2'
    assert abs(diff.p_losses(x_start, t, half).item() - 0.25) < 1e-6
    assert diff.loss(x_start, t, half).shape == ()   # :212-218 scalar loss


def test_sample_loop_shape(diff):                 # gaussian_diffusion_test.py:224-230
    g = torch.Generator().manual_seed(0)
    shape = (1, C, F, S, S)
    out = diff.p_sample_loop(torch.randn(shape, generator=g), [torch.randn(shape, generator=g) for _ in range(T)])
    assert out.shape == shape


def test_normalize_roundtrip():                   # utils_test.py:121-131
    x = torch.tensor([0.0, 0.5, 1.0])
    np.testing.assert_allclose(x * 2 - 1, [-1.0, 0.0, 1.0], atol=1e-6)
    np.testing.assert_allclose((torch.tensor([-1.0, 0.0, 1.0]) + 1) * 0.5, [0.0, 0.5, 1.0], atol=1e-6)


def test_schedule_fp32_vs_fp64():
    """Tolerance study: the fp32 tables stay within 2e-3 relative of the fp64 closed form
    (the smallest betas amplify ulp noise of alphas_cumprod ratios)."""
    a, b = schedule(1000, np.float32), schedule(1000, np.float64)
    for n in TABLE_NAMES:
        if n == 'posterior_log_variance_clipped':
            np.testing.assert_allclose(a[n][1:], b[n][1:], rtol=0, atol=5e-3)
            continue
        np.testing.assert_allclose(a[n], b[n], rtol=5e-3, atol=1e-6)


def test_schedule_golden_fixture():
    """Committed fixture tests/golden/schedule_T1000.npz (made by tests/golden/make_golden.py)."""
    import os
    path = os.path.join(os.path.dirname(__file__), 'golden', 'schedule_T1000.npz')
    z = np.load(path)
    cur = schedule(1000, np.float32)
    for n in TABLE_NAMES:
        np.testing.assert_array_equal(z[n], cur[n])


def test_dynamic_threshold_matches_manual(diff, x_start):
    d = DiffusionRef(zeros_denoise, image_size=S, num_frames=F, channels=C, timesteps=T, use_dynamic_thres=True)
    g = torch.Generator().manual_seed(3)
    x = 3 * torch.randn(x_start.shape, generator=g)
    t = torch.full((B,), 3)
    mean, _, _ = d.p_mean_variance(x, t, clip_denoised=True)
    xr = d.predict_start_from_noise(x, t, torch.zeros_like(x))
    s = torch.quantile(xr.abs().reshape(B, -1), 0.9, dim=-1).clamp_min(1.0).reshape(B, 1, 1, 1, 1)
    exp, _, _ = d.q_posterior(torch.maximum(torch.minimum(xr, s), -s) / s, x, t)
    np.testing.assert_allclose(mean, exp, atol=1e-6)


def test_ddim_oracle_properties():
    """DDIM closed form (oracle only; the reference has no DDIM: parity unpinned).  With a PERFECT noise predictor the eta = 0 chain
    recovers x_0 exactly from any x_t on the forward trajectory, for any number of steps; and one S = T chain touches every t."""
    T = 50
    g = torch.Generator().manual_seed(0)
    x0 = torch.rand(2, 1, 3, 4, 4, generator=g, dtype=torch.float64) * 1.6 - 0.8
    noise = torch.randn(2, 1, 3, 4, 4, generator=g, dtype=torch.float64)
    ref = DiffusionRef(None, image_size=4, num_frames=3, channels=1, timesteps=T, dtype=torch.float64)
    ac = ref.tab['alphas_cumprod']
    x_T = ac[T - 1].sqrt() * x0 + (1 - ac[T - 1]).sqrt() * noise
    seen = []
    def perfect(x, t):                                        # eps such that x = sqrt(ac) x0 + sqrt(1-ac) eps, channel-last
        seen.append(int(t[0]))
        a = ac[t[0]]
        return ((x - a.sqrt() * x0) / (1 - a).sqrt()).permute(0, 2, 3, 4, 1)
    ref.denoise = perfect
    for steps in (1, 5, T):
        seen.clear()
        out = ref.ddim_sample_loop(x_T, steps)
        assert (out - x0).abs().max().item() < 1e-9, steps
        assert len(seen) == steps and seen[0] == T - 1 and all(a > b for a, b in zip(seen, seen[1:]))
    assert seen == list(range(T - 1, -1, -1))
