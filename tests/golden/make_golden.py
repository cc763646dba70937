"""Regenerates the committed golden fixtures from the oracle (run from the repo root).

The real reference cannot be imported here (jax/flax/optax absent), so fixtures are
produced by the CPU restatement itself: fp64 mode as truth, fp32 mode as the tolerance reference.
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.diffusion_ref import schedule, TABLE_NAMES  # noqa: E402
from oracle.unet3d_ref import UnetConfig, random_params, unet_forward  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    tabs = schedule(1000, np.float32)
    np.savez(os.path.join(HERE, 'schedule_T1000.npz'), **{k: tabs[k] for k in TABLE_NAMES})

    # (x_t, t, cond, weight-seed) -> eps_hat for the reference's own end-to-end test config
    # (/root/reference/test_unet3d.py:12-60): dim 16, channels 3, cond_dim 32, x (1,3,4,16,16).
    for tag, cfg in (('tiny_cond', UnetConfig(dim=16, channels=3, cond_dim=32)),
                     ('tiny_nocond', UnetConfig(dim=16, channels=3))):
        g = torch.Generator().manual_seed(11)
        x = torch.randn(1, 3, 4, 16, 16, generator=g, dtype=torch.float64)
        t = torch.tensor([437])
        cond = torch.randn(1, 32, generator=g, dtype=torch.float64) if cfg.has_cond else None
        p64 = random_params(cfg, seed=5, dtype=torch.float64)
        p32 = {k: v.float() for k, v in p64.items()}
        y64 = unet_forward(p64, cfg, x, t, cond=cond)
        y32 = unet_forward(p32, cfg, x.float(), t, cond=None if cond is None else cond.float())
        np.savez(os.path.join(HERE, f'unet_{tag}.npz'), x=x.float().numpy(), t=t.numpy(),
                 cond=np.zeros(0, np.float32) if cond is None else cond.float().numpy(),
                 eps_fp64=y64.numpy(), eps_fp32=y32.numpy(), weight_seed=np.int64(5))


if __name__ == '__main__':
    main()
