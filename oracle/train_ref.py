"""ORACLE (test infrastructure only) -- CPU restatement of the reference train step.

NOT part of the product path.  Follows /root/reference/trainer.py:138-150 (optimizer + LR
schedule), :322-392 (`_pjit_train_step`: value_and_grad -> optax.adam -> EMA).  optax is not
vendored/installable here; its published algorithms are restated:

  optax.adam(lr_schedule): mu = b1 mu + (1-b1) g ; nu = b2 nu + (1-b2) g^2 ;
      p -= lr(count) * (mu / (1-b1^t)) / (sqrt(nu / (1-b2^t)) + eps),  t = count + 1,
      b1 0.9, b2 0.999, eps 1e-8 (outside the sqrt), no weight decay; lr is evaluated at the
      PRE-increment count.
  optax.piecewise_interpolate_schedule('cosine', init, {b_i: s_i}): values = cumprod(init, s_i);
      inside interval i: end + (start-end)/2 * (cos(pi*pct)+1); after the last boundary: last value.
      A zero-length interval contributes nothing ("parity unpinned" corner: optax would evaluate
      0/0 there; every shipped YAML that sets the keys has non-zero intervals).

PARITY STATUS: unpinned by reference fixtures (test_trainer.py mocks the model; only the
checkpoint cadence 2,4,5 is pinned, tested in tests/test_trainer_host.py).  Gradients come from
torch autograd through oracle/unet3d_ref.py + oracle/diffusion_ref.py.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch


def lr_schedule(step: int, train_lr: float, lr_decay_start_step: int = 0, lr_decay_steps: int = 0,
                lr_decay_coeff: float = 1.0) -> float:
    """trainer.py:138-145."""
    bs = {lr_decay_start_step: 1.0}
    bs[lr_decay_start_step + lr_decay_steps] = lr_decay_coeff      # duplicate key collapses, as the dict literal
    items = sorted(bs.items())
    bounds = [0] + [b for b, _ in items]
    values = [train_lr]
    for _, s in items:
        values.append(values[-1] * s)
    for i in range(len(bounds) - 1):
        lo, hi = bounds[i], bounds[i + 1]
        if lo <= step < hi:
            pct = (step - lo) / (hi - lo)
            start, end = values[i], values[i + 1]
            return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1.0)
    return values[-1]


def adam_update(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], mu, nu, count: int, lr: float,
                b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
    """One optax.adam step; returns (new_params, new_mu, new_nu).  `count` is the pre-increment count."""
    t = count + 1
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    new_p, new_mu, new_nu = {}, {}, {}
    for k, p in params.items():
        g = grads[k]
        m = b1 * mu[k] + (1 - b1) * g
        v = b2 * nu[k] + (1 - b2) * g * g
        upd = (m / bc1) / (torch.sqrt(v / bc2) + eps)
        new_p[k] = p - lr * upd
        new_mu[k], new_nu[k] = m, v
    return new_p, new_mu, new_nu


def ema_update(ema, new_params, step: int, step_start_ema: int, update_ema_every: int, decay: float):
    """trainer.py:373-382."""
    if step >= step_start_ema and step % update_ema_every == 0:
        return {k: decay * ema[k] + (1 - decay) * new_params[k] for k in ema}
    return ema


def loss_and_grads(params: Dict[str, torch.Tensor], loss_fn) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """jax.value_and_grad(loss_fn)(params) (trainer.py:361) via torch autograd."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    loss = loss_fn(leaves)
    grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    out = {}
    for (k, v), g in zip(leaves.items(), grads):
        out[k] = torch.zeros_like(v) if g is None else g
    return loss.detach(), out
