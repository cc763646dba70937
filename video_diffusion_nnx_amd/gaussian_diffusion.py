"""GaussianDiffusion -- host-side mirror of the reference class (/root/reference/gaussian_diffusion.py:23-502).

Same constructor / method surface; `key` arguments are integer seeds of the library's counter-based
Philox stream (JAX PRNGKeys are not reproducible outside JAX).  The heavy methods
(q_sample, p_sample, p_sample_loop, p_losses, __call__) run hand-written HIP through libvdx.so; the
closed-form accessors (q_mean_variance, predict_start_from_noise, q_posterior) are table look-ups.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib as L
from .unet3d import Unet3D

_vp = C.c_void_p
_u64 = C.c_uint64
vdx_randn = L._sig('vdx_randn', C.c_int, [_vp, C.c_long, _u64, _u64, _vp, _vp])
vdx_q_sample = L._sig('vdx_q_sample', C.c_int, [_vp] * 6 + [C.c_int, C.c_long, C.c_float, C.c_float, _vp])
vdx_p_sample_step = L._sig('vdx_p_sample_step', C.c_int, [_vp] * 5 + [C.c_int, _vp, _u64, _u64, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_long, _vp])
vdx_loss_sum = L._sig('vdx_loss_sum', C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_long, C.c_int, _vp])
vdx_affine = L._sig('vdx_affine', C.c_int, [_vp, _vp, C.c_long, C.c_float, C.c_float, _vp])
vdx_p_sample_loop = L._sig('vdx_p_sample_loop', C.c_int, [_vp] * 8 + [C.c_int, C.c_int, _vp, _u64, C.c_int, _vp, C.c_size_t, C.c_int, C.c_int, _vp])
vdx_p_sample_loop_dyn = L._sig('vdx_p_sample_loop_dyn', C.c_int, [_vp] * 8 + [C.c_int, C.c_int, _vp, _u64, C.c_int, C.c_float, _vp, _vp, C.c_size_t,
                                                                 C.c_int, C.c_int, _vp])
vdx_dynamic_threshold = L._sig('vdx_dynamic_threshold', C.c_int, [_vp] * 4 + [C.c_int, C.c_float, _vp, C.c_int, C.c_int, C.c_long, _vp])
vdx_ddim_step = L._sig('vdx_ddim_step', C.c_int, [_vp] * 7 + [C.c_int, C.c_int, C.c_int, C.c_long, _vp])
vdx_ddim_sample_loop = L._sig('vdx_ddim_sample_loop', C.c_int, [_vp] * 9 + [C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_size_t, C.c_int, C.c_int, _vp])
vdx_ddim_sample_loop_dyn = L._sig('vdx_ddim_sample_loop_dyn', C.c_int, [_vp] * 9 + [C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_float, _vp, _vp, C.c_size_t,
                                                                       C.c_int, C.c_int, _vp])


def ddim_time_sequence(timesteps: int, steps: int) -> np.ndarray:
    """The S + 1 times of an S-step DDIM chain over a T-step schedule: T-1 = t_0 > t_1 > ... > t_{S-1} >= 0, then -1 (= the data).
    Evenly spaced as linspace(-1, T-1, S+1), the usual choice (denoising-diffusion-pytorch); the reference has no DDIM."""
    assert 1 <= steps <= timesteps
    return np.ascontiguousarray(np.linspace(-1, timesteps - 1, steps + 1).astype(np.int32)[::-1])

TABLE_NAMES = (
    'alphas_cumprod', 'sqrt_alphas_cumprod', 'sqrt_one_minus_alphas_cumprod', 'log_one_minus_alphas_cumprod',
    'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod', 'posterior_variance',
    'posterior_log_variance_clipped', 'posterior_mean_coef1', 'posterior_mean_coef2',
)


def cosine_beta_schedule(timesteps: int, s: float = 0.008) -> np.ndarray:
    """reference utils.py:241-256; float32 arithmetic (JAX x64 is off in the reference)."""
    f = np.float32
    x = np.linspace(0, timesteps, timesteps + 1, dtype=np.float32)
    ac = np.cos(((x / f(timesteps)) + f(s)) / f(1 + s) * f(np.pi) * f(0.5)) ** 2
    ac = (ac / ac[0]).astype(np.float32)
    betas = f(1) - (ac[1:] / ac[:-1])
    return np.clip(betas, f(0), f(0.9999)).astype(np.float32)


def make_tables(timesteps: int) -> dict:
    """The ten schedule tables of reference gaussian_diffusion.py:78-98 (float32)."""
    f = np.float32
    betas = cosine_beta_schedule(timesteps)
    alphas = f(1) - betas
    ac = np.cumprod(alphas, axis=0, dtype=np.float32)
    ac_prev = np.concatenate([np.ones(1, np.float32), ac[:-1]])
    pv = betas * (f(1) - ac_prev) / (f(1) - ac)
    with np.errstate(divide='ignore'):
        t = {
            'alphas_cumprod': ac, 'sqrt_alphas_cumprod': np.sqrt(ac), 'sqrt_one_minus_alphas_cumprod': np.sqrt(f(1) - ac),
            'log_one_minus_alphas_cumprod': np.log(f(1) - ac), 'sqrt_recip_alphas_cumprod': np.sqrt(f(1) / ac),
            'sqrt_recipm1_alphas_cumprod': np.sqrt(f(1) / ac - f(1)), 'posterior_variance': pv,
            'posterior_log_variance_clipped': np.log(np.maximum(pv, f(1e-20))),
            'posterior_mean_coef1': betas * np.sqrt(ac_prev) / (f(1) - ac),
            'posterior_mean_coef2': (f(1) - ac_prev) * np.sqrt(alphas) / (f(1) - ac),
        }
    return {k: v.astype(np.float32) for k, v in t.items()}


def split_key(key: int, num: int = 2):
    """Deterministic seed derivation standing in for jax.random.split (splitmix64 of (key, index))."""
    out = []
    for i in range(num):
        z = (int(key) * 0x9E3779B97F4A7C15 + (i + 1) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        out.append(z ^ (z >> 31))
    return out


def dist_rank_world():
    """(rank, world) of the default torch.distributed group, (0, 1) outside one."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_key(key: int, rank: int, world: int) -> int:
    """Philox seed of rank `rank`'s shard of a data-parallel sampling batch: a function of (key, rank) only, so the videos a rank
    draws do not depend on how many other ranks there are; one rank keeps `key` itself."""
    return int(key) & 0xFFFFFFFFFFFFFFFF if world == 1 else split_key(key, rank + 1)[-1]


def extract(a: torch.Tensor, t: torch.Tensor, x_shape) -> torch.Tensor:
    """reference utils.py:225-238."""
    b = t.shape[0]
    return a.gather(-1, t.long()).reshape(b, *((1,) * (len(x_shape) - 1)))


def is_list_str(x) -> bool:
    """reference utils.py:282-293 (an empty list/tuple counts as a list of strings)."""
    return isinstance(x, (list, tuple)) and all(type(el) == str for el in x)


class GaussianDiffusion:
    def __init__(self, denoise_fn: Unet3D, *, image_size: int, num_frames: int, text_use_bert_cls: bool = False,
                 channels: int = 3, timesteps: int = 1000, loss_type: str = 'l1', use_dynamic_thres: bool = False,
                 dynamic_thres_percentile: float = 0.9, sample_act_bf16: bool = True):
        # sample_act_bf16 (extension): with a mode='bf16' Unet3D the sampling loops store the UNet's inter-kernel activations
        # as bf16 (vdx_set_activation_storage); training forwards are unaffected
        self.sample_act_bf16 = sample_act_bf16
        self.channels = channels
        self.image_size = image_size
        self.num_frames = num_frames
        self.denoise_fn = denoise_fn
        self.loss_type = loss_type
        self.text_use_bert_cls = text_use_bert_cls
        self.use_dynamic_thres = use_dynamic_thres
        self.dynamic_thres_percentile = dynamic_thres_percentile
        self.num_timesteps = int(timesteps)
        self.device = denoise_fn.device
        tabs = make_tables(self.num_timesteps)
        for name in TABLE_NAMES:
            setattr(self, name, torch.from_numpy(tabs[name]).to(self.device))
        # the five tables the reverse step needs, stacked [5][T] for the kernel
        self._ptab = torch.stack([self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod, self.posterior_mean_coef1,
                                  self.posterior_mean_coef2, self.posterior_log_variance_clipped]).contiguous()
        self._sample_stream = None

    # -- closed forms (table look-ups) -------------------------------------------------------------
    def q_mean_variance(self, x_start, t):
        mean = extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
        variance = extract(1.0 - self.alphas_cumprod, t, x_start.shape)
        log_variance = extract(self.log_one_minus_alphas_cumprod, t, x_start.shape)
        return mean, variance, log_variance

    def predict_start_from_noise(self, x_t, t, noise):
        return (extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t
                - extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * noise)

    def q_posterior(self, x_start, x_t, t):
        mean = (extract(self.posterior_mean_coef1, t, x_t.shape) * x_start + extract(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        return mean, extract(self.posterior_variance, t, x_t.shape), extract(self.posterior_log_variance_clipped, t, x_t.shape)

    # -- helpers -----------------------------------------------------------------------------------
    def _dev(self, x, dtype=torch.float32):
        return torch.as_tensor(x).to(self.device, dtype).contiguous()

    def randn(self, shape, key: int, offset: int = 0) -> torch.Tensor:
        out = torch.empty(tuple(shape), dtype=torch.float32, device=self.device)
        L.check(vdx_randn(L.ptr(out), out.numel(), int(key) & 0xFFFFFFFFFFFFFFFF, offset, 0, L.stream_ptr()))
        return out

    def _per_sample(self, x):
        return x.numel() // x.shape[0]

    def _dynamic_threshold(self, x, t, eps_hat):
        """Imagen dynamic thresholding (reference :205-217): s = max(quantile(|x0_hat| per sample, percentile), 1), exact radix
        select on the device (vdx_dynamic_threshold)."""
        s = torch.empty(x.shape[0], dtype=torch.float32, device=self.device)
        L.check(vdx_dynamic_threshold(L.ptr(x), L.ptr(eps_hat), L.ptr(t), L.ptr(self._ptab), self.num_timesteps,
                                      float(self.dynamic_thres_percentile), L.ptr(s), x.shape[0], self.channels, self._per_sample(x),
                                      L.stream_ptr()))
        return s

    # -- reverse process ---------------------------------------------------------------------------
    def p_mean_variance(self, x, t, clip_denoised: bool, cond=None, cond_scale: float = 1.0):
        """reference :162-228 (returns (mean, variance, log_variance))."""
        x = self._dev(x)
        t32 = self._dev(t, torch.int32)
        eps_hat = self.denoise_fn.forward_with_cond_scale(x, t32, cond=cond, cond_scale=cond_scale)
        thres = self._dynamic_threshold(x, t32, eps_hat) if (clip_denoised and self.use_dynamic_thres) else None
        mean = torch.empty_like(x)
        zeros = torch.zeros_like(x)      # z = 0 turns the fused step into the posterior mean
        L.check(vdx_p_sample_step(L.ptr(x), L.ptr(eps_hat), L.ptr(mean), L.ptr(t32), L.ptr(self._ptab), self.num_timesteps,
                                  L.ptr(zeros), 0, 0, 0, L.ptr(thres), int(clip_denoised), x.shape[0], self.channels,
                                  self._per_sample(x), L.stream_ptr()))
        return mean, extract(self.posterior_variance, t32, x.shape), extract(self.posterior_log_variance_clipped, t32, x.shape)

    def p_sample(self, x, t, key, cond=None, cond_scale: float = 1.0, clip_denoised: bool = True, *, noise=None):
        """reference :231-261.  `key`: seed of the noise draw (ignored when explicit `noise` is given)."""
        x = self._dev(x)
        t32 = self._dev(t, torch.int32)
        eps_hat = self.denoise_fn.forward_with_cond_scale(x, t32, cond=cond, cond_scale=cond_scale)
        thres = self._dynamic_threshold(x, t32, eps_hat) if (clip_denoised and self.use_dynamic_thres) else None
        out = torch.empty_like(x)
        nz = None if noise is None else self._dev(noise)
        L.check(vdx_p_sample_step(L.ptr(x), L.ptr(eps_hat), L.ptr(out), L.ptr(t32), L.ptr(self._ptab), self.num_timesteps,
                                  L.ptr(nz), int(key or 0) & 0xFFFFFFFFFFFFFFFF, 0, 0, L.ptr(thres), int(clip_denoised),
                                  x.shape[0], self.channels, self._per_sample(x), L.stream_ptr()))
        return out

    def p_sample_loop(self, shape, key, cond=None, cond_scale: float = 1.0, *, use_graph: bool = True, x_T=None):
        """reference :264-320.  As there, the caller's spatial `shape` is replaced by the model's own (Q10).

        x_T = Philox(key, draw 0); step k (t = T-1-k) uses draw 1+k.  Returns unnormalize_img(x_0) in [0,1].
        With cond given and cond_scale != 1 the loop runs step-by-step (two forwards per step, classifier-free
        guidance) -- an extension: the reference drops cond here.
        """
        B = int(shape[0])
        shape = (B, self.channels, self.num_frames, self.image_size, self.image_size)
        seed = int(key) & 0xFFFFFFFFFFFFFFFF
        unet = self.denoise_fn
        T = self.num_timesteps
        if self._sample_stream is None:
            self._sample_stream = torch.cuda.Stream(device=self.device)
        cur = torch.cuda.current_stream(self.device)
        st = self._sample_stream
        st.wait_stream(cur)
        keep_storage = unet.act_bf16
        unet.act_bf16 = bool(self.sample_act_bf16 and unet.mode == 'bf16')
        try:
            out = self._p_sample_loop_on(st, unet, shape, B, T, seed, cond, cond_scale, use_graph, x_T)
        finally:
            unet.act_bf16 = keep_storage
        cur.wait_stream(st)
        return out

    def _p_sample_loop_on(self, st, unet, shape, B, T, seed, cond, cond_scale, use_graph, x_T):
        with torch.cuda.stream(st):
            img = self.randn(shape, seed, 0) if x_T is None else self._dev(x_T).clone()
            guided = cond is not None and unet.has_cond and cond_scale != 1
            if guided:
                for k, i in enumerate(reversed(range(T))):
                    t = torch.full((B,), i, dtype=torch.int32, device=self.device)
                    eps_hat = unet.forward_with_cond_scale(img, t, cond=cond, cond_scale=cond_scale)
                    thres = self._dynamic_threshold(img, t, eps_hat) if self.use_dynamic_thres else None
                    L.check(vdx_p_sample_step(L.ptr(img), L.ptr(eps_hat), L.ptr(img), L.ptr(t), L.ptr(self._ptab), T, 0, seed, 1 + k, 0,
                                              L.ptr(thres), 1, B, self.channels, self._per_sample(img), L.stream_ptr()))
            else:
                condd = None if (cond is None or not unet.has_cond) else self._dev(cond)
                h = unet.handle(self.num_frames, self.image_size)
                unet.apply_activation_storage(h)
                ws = unet.workspace(B, self.num_frames, self.image_size)
                eps = torch.empty(B, self.num_frames, self.image_size, self.image_size, unet.out_dim, dtype=torch.float32, device=self.device)
                t_dev = torch.full((B,), T - 1, dtype=torch.int32, device=self.device)
                step_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
                thres = torch.empty(B, dtype=torch.float32, device=self.device) if self.use_dynamic_thres else None
                L.check(vdx_p_sample_loop_dyn(h.ptr, L.ptr(unet.flat_params), L.ptr(unet.packed()), L.ptr(img), L.ptr(eps), L.ptr(t_dev),
                                              L.ptr(step_dev), L.ptr(self._ptab), T, T, L.ptr(condd), seed, 1,
                                              float(self.dynamic_thres_percentile) if self.use_dynamic_thres else 0.0, L.ptr(thres),
                                              L.ptr(ws), ws.numel(), B, int(use_graph), L.stream_ptr()))
            out = torch.empty_like(img)
            L.check(vdx_affine(L.ptr(img), L.ptr(out), img.numel(), 0.5, 0.5, L.stream_ptr()))     # unnormalize_img
        return out

    def ddim_sample_loop(self, shape, key, steps: int = 100, cond=None, cond_scale: float = 1.0, *, use_graph: bool = True, x_T=None):
        """DDIM sampling with eta = 0 in `steps` network evaluations (EXTENSION, BASELINE.json configs[3]; the reference has ancestral
        sampling only).  x_T = Philox(key, draw 0), deterministic afterwards.  Returns unnormalize_img(x_0) in [0, 1]."""
        B = int(shape[0])
        shape = (B, self.channels, self.num_frames, self.image_size, self.image_size)
        unet = self.denoise_fn
        seq_host = ddim_time_sequence(self.num_timesteps, steps)
        if self._sample_stream is None:
            self._sample_stream = torch.cuda.Stream(device=self.device)
        cur, st = torch.cuda.current_stream(self.device), self._sample_stream
        st.wait_stream(cur)
        keep_storage = unet.act_bf16
        unet.act_bf16 = bool(self.sample_act_bf16 and unet.mode == 'bf16')
        try:
            with torch.cuda.stream(st):
                img = self.randn(shape, int(key) & 0xFFFFFFFFFFFFFFFF, 0) if x_T is None else self._dev(x_T).clone()
                seq = torch.from_numpy(seq_host).to(self.device)
                guided = cond is not None and unet.has_cond and cond_scale != 1
                if guided:
                    step_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
                    for k in range(steps):
                        t = torch.full((B,), int(seq_host[k]), dtype=torch.int32, device=self.device)
                        eps_hat = unet.forward_with_cond_scale(img, t, cond=cond, cond_scale=cond_scale)
                        step_dev.fill_(k)
                        thres = self._dynamic_threshold(img, t, eps_hat) if self.use_dynamic_thres else None
                        L.check(vdx_ddim_step(L.ptr(img), L.ptr(eps_hat), L.ptr(img), L.ptr(self.alphas_cumprod), L.ptr(seq), L.ptr(step_dev), L.ptr(thres), 1,
                                              B, self.channels, self._per_sample(img), L.stream_ptr()))
                else:
                    condd = None if (cond is None or not unet.has_cond) else self._dev(cond)
                    h = unet.handle(self.num_frames, self.image_size)
                    unet.apply_activation_storage(h)
                    ws = unet.workspace(B, self.num_frames, self.image_size)
                    eps = torch.empty(B, self.num_frames, self.image_size, self.image_size, unet.out_dim, dtype=torch.float32, device=self.device)
                    t_dev = torch.full((B,), int(seq_host[0]), dtype=torch.int32, device=self.device)
                    step_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
                    thres = torch.empty(B, dtype=torch.float32, device=self.device) if self.use_dynamic_thres else None
                    L.check(vdx_ddim_sample_loop_dyn(h.ptr, L.ptr(unet.flat_params), L.ptr(unet.packed()), L.ptr(img), L.ptr(eps), L.ptr(t_dev),
                                                     L.ptr(step_dev), L.ptr(self.alphas_cumprod), L.ptr(seq), steps, steps, L.ptr(condd), 1,
                                                     L.ptr(self._ptab), self.num_timesteps,
                                                     float(self.dynamic_thres_percentile) if self.use_dynamic_thres else 0.0, L.ptr(thres),
                                                     L.ptr(ws), ws.numel(), B, int(use_graph), L.stream_ptr()))
                out = torch.empty_like(img)
                L.check(vdx_affine(L.ptr(img), L.ptr(out), img.numel(), 0.5, 0.5, L.stream_ptr()))     # unnormalize_img
        finally:
            unet.act_bf16 = keep_storage
        cur.wait_stream(st)
        return out

    def sample(self, key, cond=None, cond_scale: float = 1.0, batch_size: int = 16, *, ddim_steps: Optional[int] = None, **kw):
        """reference :323-357.  ddim_steps (extension): sample with an S-step DDIM chain instead of the T-step ancestral one.

        Data parallel (reference :278-298: the batch is split over the local devices, `P('data')`): inside an initialised
        torch.distributed group of W > 1 ranks, `batch_size` (and `cond`) describe the GLOBAL batch; rank r draws videos
        [r * per, (r + 1) * per), per = batch_size / W, from its own Philox stream shard_key(key, r) and returns ITS shard --
        no data-path collective.  W = 1 uses `key` itself (single-process behaviour unchanged)."""
        if is_list_str(cond):
            raise NotImplementedError('text -> BERT embedding needs the external video_diffusion_pytorch.text (network fetch); '
                                      'pass a ready [B, 768] tensor instead')
        if cond is not None:
            batch_size = cond.shape[0]
        rank, world = dist_rank_world()
        if world > 1:
            assert batch_size % world == 0, 'batch_size must be divisible by number of devices'      # as reference trainer.py:163
            per = batch_size // world
            batch_size, key = per, shard_key(key, rank, world)
            if cond is not None:
                cond = cond[rank * per:(rank + 1) * per]
        shape = (batch_size, self.channels, self.num_frames, self.image_size, self.image_size)
        if ddim_steps:
            return self.ddim_sample_loop(shape, key, steps=int(ddim_steps), cond=cond, cond_scale=cond_scale, **kw)
        return self.p_sample_loop(shape, key, cond=cond, cond_scale=cond_scale, **kw)

    def interpolate(self, x1, x2, t: Optional[int] = None, lam: float = 0.5, key: int = 0):
        """reference :360-398 with the intended behaviour (the reference omits the mandatory keys, Q18)."""
        b = x1.shape[0]
        t = t if t is not None else self.num_timesteps - 1
        assert x1.shape == x2.shape and 0.0 <= lam <= 1.0
        tb = torch.full((b,), t, dtype=torch.int32, device=self.device)
        k1, k2, k3 = split_key(key, 3)
        img = (1 - lam) * self.q_sample(x1, tb, k1) + lam * self.q_sample(x2, tb, k2)
        for n, i in enumerate(reversed(range(0, t))):
            img = self.p_sample(img, torch.full((b,), i, dtype=torch.int32, device=self.device), split_key(k3, n + 1)[-1])
        return img

    # -- forward process / loss ----------------------------------------------------------------------
    def q_sample(self, x_start, t, key=None, noise=None, *, _pre=(1.0, 0.0)):
        """reference :401-420."""
        x_start = self._dev(x_start)
        t32 = self._dev(t, torch.int32)
        if noise is None:
            assert key is not None, 'A key must be provided to q_sample if noise is not.'
            noise = self.randn(x_start.shape, key, 0)
        noise = self._dev(noise)
        out = torch.empty_like(x_start)
        L.check(vdx_q_sample(L.ptr(x_start), L.ptr(t32), L.ptr(noise), L.ptr(out), L.ptr(self.sqrt_alphas_cumprod),
                             L.ptr(self.sqrt_one_minus_alphas_cumprod), x_start.shape[0], self._per_sample(x_start),
                             float(_pre[0]), float(_pre[1]), L.stream_ptr()))
        return out

    def p_losses(self, x_start, t, key=None, cond=None, noise=None, *, _pre=(1.0, 0.0), **kwargs):
        """reference :423-470 (forward value; the training step with gradients lives in trainer.py)."""
        if self.loss_type not in ('l1', 'l2'):
            raise ValueError(f'Unsupported loss type: {self.loss_type}')
        x_start = self._dev(x_start)
        t32 = self._dev(t, torch.int32)
        if noise is None:
            assert key is not None
            _, noise_key, _ = split_key(key, 3)
            noise = self.randn(x_start.shape, noise_key, 0)
        noise = self._dev(noise)
        x_noisy = self.q_sample(x_start, t32, noise=noise, _pre=_pre)
        if is_list_str(cond):
            raise NotImplementedError('pass text conditioning as a ready embedding tensor')
        eps_hat = self.denoise_fn(x_noisy, t32, cond=cond, **kwargs)
        acc = torch.zeros(1, dtype=torch.float64, device=self.device)
        B = x_start.shape[0]
        fhw = self._per_sample(x_start) // self.channels
        L.check(vdx_loss_sum(L.ptr(eps_hat), L.ptr(noise), L.ptr(acc), B, self.channels, fhw, int(self.loss_type == 'l2'), L.stream_ptr()))
        return (acc / float(x_start.numel())).to(torch.float32).reshape(())

    def __call__(self, x, key, *args, **kwargs):
        """reference :473-502: random t, normalize_img, p_losses."""
        b, c, f, h, w = x.shape
        assert (c, f, h, w) == (self.channels, self.num_frames, self.image_size, self.image_size), \
            f'expected [b, {self.channels}, {self.num_frames}, {self.image_size}, {self.image_size}], got {tuple(x.shape)}'
        _, t_key, loss_key = split_key(key, 3)
        g = torch.Generator().manual_seed(t_key & 0x7FFFFFFFFFFFFFFF)
        t = torch.randint(0, self.num_timesteps, (b,), generator=g, dtype=torch.int32)
        return self.p_losses(x, t, loss_key, *args, _pre=(2.0, -1.0), **kwargs)      # normalize_img folded into q_sample
