"""ORACLE (test infrastructure only) -- CPU restatement of the reference GaussianDiffusion math.

NOT part of the product path (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import it).  Follows:

  * schedule tables .......... /root/reference/gaussian_diffusion.py:77-98,
                               /root/reference/utils.py:241-256 (cosine_beta_schedule)
  * extract .................. /root/reference/utils.py:225-238
  * q_mean_variance .......... gaussian_diffusion.py:101-117
  * predict_start_from_noise . gaussian_diffusion.py:120-136
  * q_posterior .............. gaussian_diffusion.py:139-159
  * p_mean_variance .......... gaussian_diffusion.py:162-228
  * p_sample ................. gaussian_diffusion.py:231-261
  * p_sample_loop ............ gaussian_diffusion.py:264-320
  * q_sample ................. gaussian_diffusion.py:401-420
  * p_losses / __call__ ...... gaussian_diffusion.py:423-502

PARITY STATUS: pinned by the reference's own closed-form known answers
(/root/reference/gaussian_diffusion_test.py:75-218, /root/reference/utils_test.py:102-131), which
tests/test_oracle_diffusion.py reproduces as values.  Random streams are NOT comparable with
JAX threefry; every function here takes explicit noise (the product's Philox stream is restated
in oracle/philox_ref.py).

Schedule dtype: the reference asks linspace for float64 but JAX x64 is off by
default, so the tables are float32 end-to-end; `schedule(T, dtype=np.float32)` follows the same
operation order in NumPy float32, `dtype=np.float64` is the closed form used for tolerance study.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np
import torch

TABLE_NAMES = (
    'alphas_cumprod', 'sqrt_alphas_cumprod', 'sqrt_one_minus_alphas_cumprod',
    'log_one_minus_alphas_cumprod', 'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod',
    'posterior_variance', 'posterior_log_variance_clipped', 'posterior_mean_coef1',
    'posterior_mean_coef2',
)


def cosine_beta_schedule(timesteps: int, s: float = 0.008, dtype=np.float32) -> np.ndarray:
    """utils.py:241-256 in NumPy `dtype` arithmetic."""
    dt = np.dtype(dtype).type
    steps = timesteps + 1
    x = np.linspace(0, timesteps, steps, dtype=dtype)
    ac = np.cos(((x / dt(timesteps)) + dt(s)) / dt(1 + s) * dt(np.pi) * dt(0.5)) ** 2
    ac = (ac / ac[0]).astype(dtype)
    betas = dt(1) - (ac[1:] / ac[:-1])
    return np.clip(betas, dt(0), dt(0.9999)).astype(dtype)


def schedule(timesteps: int, dtype=np.float32) -> Dict[str, np.ndarray]:
    """The ten length-T tables of gaussian_diffusion.py:85-98 (+ 'betas')."""
    dt = np.dtype(dtype).type
    betas = cosine_beta_schedule(timesteps, dtype=dtype).astype(dtype)
    alphas = dt(1) - betas
    ac = np.cumprod(alphas, axis=0, dtype=dtype)
    ac_prev = np.concatenate([np.ones(1, dtype), ac[:-1]])
    pv = betas * (dt(1) - ac_prev) / (dt(1) - ac)
    with np.errstate(divide='ignore'):
        tabs = {
            'betas': betas,
            'alphas_cumprod': ac,
            'sqrt_alphas_cumprod': np.sqrt(ac),
            'sqrt_one_minus_alphas_cumprod': np.sqrt(dt(1) - ac),
            'log_one_minus_alphas_cumprod': np.log(dt(1) - ac),
            'sqrt_recip_alphas_cumprod': np.sqrt(dt(1) / ac),
            'sqrt_recipm1_alphas_cumprod': np.sqrt(dt(1) / ac - dt(1)),
            'posterior_variance': pv,
            'posterior_log_variance_clipped': np.log(np.maximum(pv, dt(1e-20))),
            'posterior_mean_coef1': betas * np.sqrt(ac_prev) / (dt(1) - ac),
            'posterior_mean_coef2': (dt(1) - ac_prev) * np.sqrt(alphas) / (dt(1) - ac),
        }
    return {k: v.astype(dtype) for k, v in tabs.items()}


def extract(a: torch.Tensor, t: torch.Tensor, x_shape) -> torch.Tensor:
    """utils.py:225-238: a[t] reshaped to (B, 1, ..., 1)."""
    b = t.shape[0]
    out = a.gather(-1, t.long())
    return out.reshape(b, *((1,) * (len(x_shape) - 1)))


class DiffusionRef:
    """Functional restatement of GaussianDiffusion around a `denoise(x, t) -> [B,F,H,W,C]` callable."""

    def __init__(self, denoise: Optional[Callable], *, image_size: int, num_frames: int, channels: int = 3,
                 timesteps: int = 1000, loss_type: str = 'l1', use_dynamic_thres: bool = False,
                 dynamic_thres_percentile: float = 0.9, dtype=torch.float32, tables=None):
        self.denoise = denoise
        self.image_size, self.num_frames, self.channels = image_size, num_frames, channels
        self.num_timesteps = int(timesteps)
        self.loss_type = loss_type
        self.use_dynamic_thres = use_dynamic_thres
        self.dynamic_thres_percentile = dynamic_thres_percentile
        tabs = tables if tables is not None else schedule(self.num_timesteps, np.float32)
        self.tab = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in tabs.items()}

    # -- forward process ---------------------------------------------------------------------
    def q_mean_variance(self, x_start, t):
        mean = extract(self.tab['sqrt_alphas_cumprod'], t, x_start.shape) * x_start
        variance = extract(1.0 - self.tab['alphas_cumprod'], t, x_start.shape)
        log_variance = extract(self.tab['log_one_minus_alphas_cumprod'], t, x_start.shape)
        return mean, variance, log_variance

    def q_sample(self, x_start, t, noise):
        return (extract(self.tab['sqrt_alphas_cumprod'], t, x_start.shape) * x_start
                + extract(self.tab['sqrt_one_minus_alphas_cumprod'], t, x_start.shape) * noise)

    # -- reverse process ---------------------------------------------------------------------
    def predict_start_from_noise(self, x_t, t, noise):
        return (extract(self.tab['sqrt_recip_alphas_cumprod'], t, x_t.shape) * x_t
                - extract(self.tab['sqrt_recipm1_alphas_cumprod'], t, x_t.shape) * noise)

    def q_posterior(self, x_start, x_t, t):
        mean = (extract(self.tab['posterior_mean_coef1'], t, x_t.shape) * x_start
                + extract(self.tab['posterior_mean_coef2'], t, x_t.shape) * x_t)
        var = extract(self.tab['posterior_variance'], t, x_t.shape)
        logvar = extract(self.tab['posterior_log_variance_clipped'], t, x_t.shape)
        return mean, var, logvar

    def p_mean_variance(self, x, t, clip_denoised: bool, eps_pred=None):
        """`eps_pred` ([B,F,H,W,C]) overrides the denoiser call (for elementwise parity tests)."""
        out = eps_pred if eps_pred is not None else self.denoise(x, t)
        predicted_noise = out.permute(0, 4, 1, 2, 3)                      # b f h w c -> b c f h w
        x_recon = self.predict_start_from_noise(x, t, predicted_noise)
        if clip_denoised:
            s = 1.0
            if self.use_dynamic_thres:
                flat = x_recon.abs().reshape(x_recon.shape[0], -1)
                s = torch.quantile(flat, self.dynamic_thres_percentile, dim=-1)   # linear interp, as jnp
                s = s.clamp_min(1.0).reshape(-1, 1, 1, 1, 1)
                x_recon = torch.maximum(torch.minimum(x_recon, s), -s) / s
            else:
                x_recon = x_recon.clamp(-s, s) / s
        return self.q_posterior(x_recon, x, t)

    def p_sample(self, x, t, noise, clip_denoised: bool = True, eps_pred=None):
        mean, _, logvar = self.p_mean_variance(x, t, clip_denoised, eps_pred=eps_pred)
        nonzero = (1.0 - (t == 0).to(x.dtype)).reshape(-1, 1, 1, 1, 1)
        return mean + nonzero * torch.exp(0.5 * logvar) * noise

    def p_sample_loop(self, x_T, noises):
        """x_T and the per-step noise list replace the JAX key stream; returns (img+1)/2."""
        img = x_T
        b = x_T.shape[0]
        for n, i in enumerate(reversed(range(self.num_timesteps))):
            t = torch.full((b,), i, dtype=torch.int64)
            img = self.p_sample(img, t, noises[n])
        return (img + 1) * 0.5

    def ddim_sample_loop(self, x_T, steps: int, clip_denoised: bool = True):
        """DDIM, eta = 0 (Song, Meng & Ermon 2020, eq. 12) over the evenly spaced sub-sequence linspace(-1, T-1, steps+1).
        PARITY UNPINNED: the reference has no DDIM sampler (gaussian_diffusion.py:264-320 is ancestral DDPM only); this is the closed
        form restated from the paper, used to check the HIP step (BASELINE.json configs[3]).  Returns x_0 in [-1, 1]."""
        import numpy as _np
        times = _np.linspace(-1, self.num_timesteps - 1, steps + 1).astype(_np.int64)[::-1]
        ac = self.tab['alphas_cumprod']
        x = x_T.to(ac.dtype)
        b = x.shape[0]
        for k in range(steps):
            t, tn = int(times[k]), int(times[k + 1])
            eps = self.denoise(x, torch.full((b,), t, dtype=torch.long)).permute(0, 4, 1, 2, 3)
            a_t = ac[t]
            a_n = ac[tn] if tn >= 0 else torch.ones((), dtype=ac.dtype)
            x0 = (x - (1 - a_t).sqrt() * eps) / a_t.sqrt()
            if clip_denoised and self.use_dynamic_thres:      # the threshold of p_mean_variance (gaussian_diffusion.py:205-220) applied to DDIM's x0
                s = torch.quantile(x0.abs().reshape(b, -1), self.dynamic_thres_percentile, dim=-1).clamp_min(1.0).reshape(-1, 1, 1, 1, 1)
                x0 = torch.maximum(torch.minimum(x0, s), -s) / s
            elif clip_denoised:
                x0 = x0.clamp(-1.0, 1.0)
            eps2 = (x - a_t.sqrt() * x0) / (1 - a_t).sqrt()
            x = a_n.sqrt() * x0 + (1 - a_n).sqrt() * eps2
        return x

    # -- training loss -----------------------------------------------------------------------
    def p_losses(self, x_start, t, noise, eps_pred=None):
        x_noisy = self.q_sample(x_start, t, noise)
        out = eps_pred if eps_pred is not None else self.denoise(x_noisy, t)
        predicted = out.permute(0, 4, 1, 2, 3)
        if self.loss_type == 'l1':
            return (predicted - noise).abs().mean()
        if self.loss_type == 'l2':
            return ((predicted - noise) ** 2).mean()
        raise ValueError(f'Unsupported loss type: {self.loss_type}')

    def loss(self, x, t, noise):
        """__call__ (gaussian_diffusion.py:473-502) with explicit t / noise: normalize then p_losses."""
        b, c, f, h, w = x.shape
        assert (c, f, h, w) == (self.channels, self.num_frames, self.image_size, self.image_size)
        return self.p_losses(x * 2 - 1, t, noise)
