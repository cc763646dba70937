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


def run_train_step(tr, batch: torch.Tensor, step: int, t: torch.Tensor = None, noise: torch.Tensor = None) -> torch.Tensor:
    """loss, grads = value_and_grad(p_losses); Adam; EMA  (trainer.py:337-382) for this rank's shard of the batch.

    `t` [B] / `noise` [B,C,F,H,W] override this rank's own draws (the reference threads `noise` through p_losses the same way,
    gaussian_diffusion.py:423-445); the data-parallel tests use them to give N ranks the shards of ONE global draw."""
    gd, unet = tr.model, tr.unet
    dev = tr.device
    x = torch.as_tensor(batch).to(dev, torch.float32).contiguous()
    B = x.shape[0]
    assert tuple(x.shape[1:]) == (gd.channels, gd.num_frames, gd.image_size, gd.image_size), \
        f'expected [b, {gd.channels}, {gd.num_frames}, {gd.image_size}, {gd.image_size}], got {tuple(x.shape)}'     # check_shape (:490)
    # keys: one stream per (seed, rank), split per step like `key, step_key = split(key)` (trainer.py:541)
    step_key = split_key(split_key(tr.rng_seed, tr.rank + 1)[-1], step + 1)[-1]
    _, t_key, loss_key = split_key(step_key, 3)
    _, noise_key, _ = split_key(loss_key, 3)
    if t is None:
        g = torch.Generator().manual_seed(t_key & 0x7FFFFFFFFFFFFFFF)
        t = torch.randint(0, gd.num_timesteps, (B,), generator=g, dtype=torch.int32)
    t = torch.as_tensor(t)
    if t.device.type == 'cpu' and dev.type == 'cuda':
        # (pinned + non-blocking: a pageable host-to-device copy waits for the stream to drain -- one host sync per step that kept the
        # launch queue from running ahead of the GPU)
        t = t.to(torch.int32).pin_memory().to(dev, non_blocking=True)
    else:
        t = t.to(dev, torch.int32)
    noise = gd.randn(x.shape, noise_key, 0) if noise is None else torch.as_tensor(noise).to(dev, torch.float32).contiguous()
    tr.last_t, tr.last_noise_key = t, noise_key
    x_noisy = gd.q_sample(x, t, noise=noise, _pre=(2.0, -1.0))                    # normalize_img folded in (:499)
    keep_storage = unet.act_bf16
    unet.act_bf16 = 2 if (unet.mode == 'bf16' and tr.train_act_bf16) else False
    try:
        eps_hat = unet(x_noisy, t)
    finally:
        unet.act_bf16 = keep_storage
    acc = torch.zeros(1, dtype=torch.float64, device=dev)
    fhw = x.numel() // (B * gd.channels)
    l2 = int(gd.loss_type == 'l2')
    L.check(vdx_loss_sum(L.ptr(eps_hat), L.ptr(noise), L.ptr(acc), B, gd.channels, fhw, l2, L.stream_ptr()))
    loss = (acc / float(x.numel())).to(torch.float32).reshape(())
    d_eps = torch.empty_like(eps_hat)
    L.check(vdx_loss_grad(L.ptr(eps_hat), L.ptr(noise), L.ptr(d_eps), B, gd.channels, fhw, l2, L.stream_ptr()))
    # reverse pass in groups of stages that end where a gradient bucket becomes complete: finished buckets are all-reduced (RCCL)
    # while earlier stages still run.  One call per group, not per stage: a call ends with the main stream waiting for the
    # weight-gradient stream (vdx_unet_backward), which nothing needs between bucket boundaries -- and never on one GPU.
    reducer = tr.make_reducer()
    ns = unet.num_stages
    cuts = sorted({min(max(b[2], 0), ns - 1) for b in tr.buckets} | {0}, reverse=True) if reducer.enabled else [0]
    hi = ns - 1
    for lo in cuts:
        if lo > hi:
            continue
        unet.backward(d_eps, tr.grads, hi, lo)
        for stage in range(hi, lo - 1, -1):
            reducer.stage_done(stage)
        hi = lo - 1
    reducer.finish()
    lr = tr.current_lr(tr.opt_count)                                               # schedule at the pre-increment count (B.2)
    do_ema = int(step >= tr.step_start_ema and step % tr.update_ema_every == 0)    # trainer.py:373-374
    L.check(vdx_adam_ema_step(L.ptr(unet.flat_params), L.ptr(tr.grads), L.ptr(tr.m), L.ptr(tr.v), L.ptr(tr.ema), unet.flat_params.numel(),
                              lr, 0.9, 0.999, 1e-8, tr.opt_count, 1.0 / reducer.world, do_ema, tr.ema_decay, L.stream_ptr()))
    tr.opt_count += 1
    unet.mark_params_updated()
    return loss
