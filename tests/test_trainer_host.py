"""CPU tests of the Trainer host logic: LR schedule vs the oracle, checkpoint cadence (reference test_trainer.py:147-161:
saves at steps 2, 4 in-loop + 5 final), bucket partition, and the 2-rank gloo gradient all-reduce."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import train_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lr_schedule_matches_oracle():
    from video_diffusion_nnx_amd.trainer import lr_schedule
    for args in ((1e-4, 20000, 80000, 0.1), (1e-4, 0, 0, 1.0), (3e-4, 10, 5, 0.5), (1e-5, 0, 100, 0.1)):
        for step in (0, 1, 9, 10, 12, 14, 15, 50, 20000, 60000, 100000, 150000):
            assert abs(lr_schedule(step, *args) - train_ref.lr_schedule(step, *args)) < 1e-15
    assert abs(lr_schedule(60000, 1e-4, 20000, 80000, 0.1) - (1e-5 + (1e-4 - 1e-5) / 2 * (np.cos(np.pi * 0.5) + 1))) < 1e-12
    assert lr_schedule(10 ** 6, 1e-4, 20000, 80000, 0.1) == pytest.approx(1e-5)


def _mock_trainer(tmp_path, steps=5, every=2, **kw):
    from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion
    from video_diffusion_nnx_amd.trainer import Trainer
    from video_diffusion_nnx_amd.unet3d import Unet3D
    unet = Unet3D(dim=16, rngs=0, channels=1, device='cpu')
    gd = GaussianDiffusion(unet, image_size=8, num_frames=2, channels=1, timesteps=10)
    return Trainer(gd, str(tmp_path), dataset_path='synthetic:8', train_batch_size=2, train_num_steps=steps,
                   checkpoint_every_steps=every, results_folder=str(tmp_path / 'res'), **kw)


def test_checkpoint_cadence_2_4_5(tmp_path, monkeypatch):
    tr = _mock_trainer(tmp_path)
    saved, losses = [], []
    monkeypatch.setattr(tr, '_save', lambda step: saved.append(step))
    monkeypatch.setattr(tr, 'train_step', lambda batch, step: torch.tensor(1.0))      # MockDiffusionModel: loss == 1.0
    tr.train(log_fn=lambda d: losses.append(d))
    assert saved == [2, 4, 5]                                                           # test_trainer.py:147-161
    assert [d['step'] for d in losses] == [0, 1, 2, 3, 4] and all(d['loss'] == 1.0 for d in losses)   # :138-145
    assert tr.step == 5


def test_trainer_fields_and_real_checkpoint(tmp_path, monkeypatch):
    tr = _mock_trainer(tmp_path, steps=3, every=2, max_to_keep=1)
    assert tr.per_device_bs == 2 and tr.world == 1 and tr.ema.shape == tr.unet.flat_params.shape
    monkeypatch.setattr(tr, 'train_step', lambda batch, step: torch.tensor(0.5))
    tr.train()
    assert tr.ckpt_manager.all_steps() == [3]                 # step 2 pruned by max_to_keep=1, final save at 3
    assert next(iter(tr.dl)).shape == (2, 1, 2, 8, 8)
    with pytest.raises(AssertionError):
        _mock_trainer(tmp_path, num_model_shards=2)


def test_buckets_cover_buffer_in_ready_order():
    from video_diffusion_nnx_amd.trainer import make_buckets
    from video_diffusion_nnx_amd.train_step import stage_of_param
    from video_diffusion_nnx_amd.unet3d import Unet3D
    m = Unet3D(dim=32, rngs=0, channels=1, device='cpu')
    total = m.flat_params.numel()
    n_stages = stage_of_param('__count__', len(m.dim_mults))
    b = make_buckets(m.param_table, total, lambda n: stage_of_param(n, len(m.dim_mults)), n_stages, min_bucket_floats=1 << 20)
    assert b[0][1] == total and b[-1][0] == 0
    for (lo, hi, st), (lo2, hi2, st2) in zip(b, b[1:]):
        assert lo == hi2 and st >= st2                          # contiguous, ready-stage non-increasing
    assert sum(hi - lo for lo, hi, _ in b) == total


def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from video_diffusion_nnx_amd.trainer import GradBucketReducer
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    red = GradBucketReducer(g, [(600, 1000, 3), (250, 600, 1), (0, 250, 0)])
    red.stage_done(3)
    red.stage_done(2)
    red.stage_done(1)
    red.finish()
    q.put((rank, g.clone()))
    dist.destroy_process_group()


def test_two_rank_gloo_bucket_allreduce():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 500
    ps = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    outs = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    exp = torch.arange(1000, dtype=torch.float32) * 3          # sum over ranks of (rank + 1) * arange
    assert torch.equal(outs[0], exp) and torch.equal(outs[1], exp)
