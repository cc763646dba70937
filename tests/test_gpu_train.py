"""GPU parity of ONE train step (q_sample -> forward -> loss -> backward -> Adam -> EMA) vs the oracle's train step
(oracle/train_ref.py over oracle/unet3d_ref.py + oracle/diffusion_ref.py with the restated Philox noise), then a short
loss-decrease run through Trainer.train()."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import philox_ref, train_ref, unet3d_ref as R
from oracle.diffusion_ref import DiffusionRef


def _mk(tmp_path, mode='f32', loss='l2', steps=3, **kw):
    from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion
    from video_diffusion_nnx_amd.trainer import Trainer
    from video_diffusion_nnx_amd.unet3d import Unet3D
    ukw = dict(dim=16, channels=1, dim_mults=(1, 2))
    unet = Unet3D(rngs=1, mode=mode, **ukw)
    gd = GaussianDiffusion(unet, image_size=8, num_frames=4, channels=1, timesteps=50, loss_type=loss)
    tr = Trainer(gd, str(tmp_path), dataset_path='synthetic:8', train_batch_size=2, train_num_steps=steps, train_lr=1e-3,
                 checkpoint_every_steps=2, results_folder=str(tmp_path / 'res'), step_start_ema=0, update_ema_every=1, ema_decay=0.9, **kw)
    return ukw, unet, gd, tr


@pytest.mark.parametrize('loss', ['l2', 'l1'])
def test_one_train_step_matches_oracle(tmp_path, loss):
    ukw, unet, gd, tr = _mk(tmp_path, loss=loss)
    cfg = R.UnetConfig(**ukw)
    p0 = {k: v.detach().cpu().double().clone() for k, v in unet.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    batch = torch.rand(2, 1, 4, 8, 8, generator=g)
    loss_dev = tr.train_step(batch, step=0)
    torch.cuda.synchronize()
    t = tr.last_t.cpu().long()
    noise = torch.from_numpy(philox_ref.randn(batch.numel(), tr.last_noise_key, 0)).double().reshape(batch.shape)
    def loss_fn(params):
        ref = DiffusionRef(lambda a, b: R.unet_forward(params, cfg, a, b), image_size=8, num_frames=4, channels=1, timesteps=50,
                           loss_type=loss, dtype=torch.float64)
        return ref.loss(batch.double(), t, noise)
    ref_loss, grads = train_ref.loss_and_grads(p0, loss_fn)
    assert abs(loss_dev.item() - ref_loss.item()) < 2e-5 * max(1.0, abs(ref_loss.item()))
    zeros = {k: torch.zeros_like(v) for k, v in p0.items()}
    lr = train_ref.lr_schedule(0, 1e-3)
    p1, m1, v1 = train_ref.adam_update(p0, grads, zeros, zeros, count=0, lr=lr)
    ema1 = train_ref.ema_update(p0, p1, step=0, step_start_ema=0, update_ema_every=1, decay=0.9)
    got = {k: v.detach().cpu().double() for k, v in unet.state_dict().items()}
    # Adam's first step is lr * sign(g) (|update| = lr): compare the UPDATE, tolerating sign flips of ~zero gradients
    num = sum(((got[k] - p0[k]) - (p1[k] - p0[k])).pow(2).sum() for k in p0)
    den = sum((p1[k] - p0[k]).pow(2).sum() for k in p0)
    assert (num / den).sqrt().item() < 2e-2, (num / den).sqrt().item()
    ema_got = {n: tr.ema[o:o + int(np.prod(s))].cpu().double().reshape(s) for n, s, o in unet.param_table}
    num = sum((ema_got[k] - ema1[k]).pow(2).sum() for k in p0)
    den = sum((ema1[k] - p0[k]).pow(2).sum() for k in p0)
    assert (num / den).sqrt().item() < 2e-2
    assert tr.opt_count == 1


def test_short_training_run_decreases_loss(tmp_path):
    _, unet, gd, tr = _mk(tmp_path, mode='bf16', steps=30)
    losses = []
    tr.train(log_fn=lambda d: losses.append(d['loss']))
    assert len(losses) == 30 and all(np.isfinite(losses))
    assert np.mean(losses[-5:]) < 0.7 * np.mean(losses[:5]), (losses[:5], losses[-5:])
    assert tr.ckpt_manager.all_steps()[-1] == 30 and torch.isfinite(unet.flat_params).all()


def test_device_prefetcher_pinned_side_stream():
    """DevicePrefetcher on the GPU: pinned staging buffers, copies on a side stream, the consumer's stream waits on
    the copy event; every batch arrives intact and in order while the consumer keeps the device busy."""
    from video_diffusion_nnx_amd.datasets import DevicePrefetcher
    g = torch.Generator().manual_seed(0)
    batches = [torch.rand(4, 1, 4, 16, 16, generator=g) for _ in range(6)]
    pf = DevicePrefetcher(iter(batches), 'cuda', select=lambda b: b[2:4])
    assert pf.on_gpu and pf._pinned[0].is_pinned()
    busy = torch.zeros(1 << 22, device='cuda')
    for i, dev in enumerate(pf):
        busy += 1.0                                            # keep the compute stream occupied between batches
        assert dev.is_cuda and dev.shape == (2, 1, 4, 16, 16)
        assert torch.equal(dev.cpu(), batches[i][2:4])
    assert i == 5


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_train_steps_are_bit_reproducible(tmp_path, mode):
    """Two trainers built alike take the same three steps on the same batches: losses, parameters, Adam moments and EMA end bit-identical
    (round 3: forward statistics in f64, backward without float atomics, Philox noise keyed by (seed, step) -- a train step is a function of
    its inputs only)."""
    ends = []
    for run in range(2):
        _, unet, gd, tr = _mk(tmp_path / f'r{run}', mode=mode, steps=3)
        g = torch.Generator().manual_seed(5)
        losses = []
        for step in range(3):
            batch = torch.rand(2, 1, 4, 8, 8, generator=g)
            losses.append(tr.train_step(batch, step=step).item())
        torch.cuda.synchronize()
        ends.append((losses, unet.flat_params.clone(), tr.ema.clone()))
    assert ends[0][0] == ends[1][0], (ends[0][0], ends[1][0])
    assert torch.equal(ends[0][1], ends[1][1]) and torch.equal(ends[0][2], ends[1][2])
