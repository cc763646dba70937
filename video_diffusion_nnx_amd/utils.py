"""Small host helpers mirroring the reference's utils.py surface (behaviour per /root/reference/utils.py and its
tests /root/reference/utils_test.py:60-176).  Pure Python / NumPy / torch-CPU: none of this is on the device path."""
from __future__ import annotations

import numpy as np
import torch


def exists(x):
    return x is not None


def noop(*args, **kwargs):
    pass


def is_odd(n):
    return (n % 2) == 1


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


def identity(t, *args, **kwargs):
    return t


def cycle(dl):
    while True:
        for data in dl:
            yield data


def num_to_groups(num, divisor):
    groups, remainder = divmod(num, divisor)
    arr = [divisor] * groups
    if remainder:
        arr.append(remainder)
    return arr


def cast_num_frames(t, *, frames):
    """(C, F, H, W): equal -> unchanged, longer -> truncated, shorter -> zero padded (utils.py:380-397)."""
    f = t.shape[1]
    if f == frames:
        return t
    if f > frames:
        return t[:, :frames, ...]
    pad = [(0, 0), (0, frames - f), (0, 0), (0, 0)]
    if isinstance(t, torch.Tensor):
        return torch.nn.functional.pad(t, (0, 0, 0, 0, 0, frames - f))
    return np.pad(t, pad)


def get_text_from_path(path):
    out = path.split('/')[-1].split('.')[0]
    return out.replace('-', ' ').replace('_', ' ')


def prob_mask_like(shape, prob, generator=None):
    """utils.py:85-101 with the host generator for 0 < prob < 1."""
    if prob == 1:
        return torch.ones(shape, dtype=torch.bool)
    if prob == 0:
        return torch.zeros(shape, dtype=torch.bool)
    return torch.rand(shape, generator=generator) < prob


def normalize_img(t):
    return t * 2 - 1


def unnormalize_img(t):
    return (t + 1) * 0.5


def clip_grad_norm(grads, max_grad_norm, epsilon=1e-6):
    """utils.py:127-152 over a dict (or single tensor) of gradients; returns (clipped, pre-clip l2 norm).
    Never called by the reference trainer; kept for API completeness."""
    items = grads if isinstance(grads, dict) else {'g': grads}
    total = sum((g.double() ** 2).sum() for g in items.values())
    l2 = torch.sqrt(total + epsilon)
    scale = torch.clamp(max_grad_norm / (l2 + epsilon), max=1.0)
    out = {k: (g * scale.to(g.dtype)) for k, g in items.items()}
    return (out if isinstance(grads, dict) else out['g']), l2
