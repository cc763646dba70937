"""CPU tests of the host-side helpers and CLI plumbing (values from /root/reference/utils_test.py:145-176)."""
import numpy as np
import pytest
import torch
import yaml


def test_utils_known_answers():
    from video_diffusion_nnx_amd import utils as U
    assert U.num_to_groups(10, 3) == [3, 3, 3, 1] and U.num_to_groups(9, 3) == [3, 3, 3]
    assert U.num_to_groups(5, 5) == [5] and U.num_to_groups(2, 3) == [2] and U.num_to_groups(0, 3) == []
    t = np.ones((3, 10, 4, 4), np.float32)
    assert U.cast_num_frames(t, frames=10) is t
    assert U.cast_num_frames(t, frames=5).shape == (3, 5, 4, 4)
    p = U.cast_num_frames(t, frames=15)
    assert p.shape == (3, 15, 4, 4) and (p[:, 10:] == 0).all() and (p[:, :10] == 1).all()
    assert U.cast_num_frames(torch.ones(3, 10, 4, 4), frames=15).shape == (3, 15, 4, 4)
    assert U.get_text_from_path('/a/b/c/cool-video_test.gif') == 'cool video test' and U.get_text_from_path('simple.mp4') == 'simple'
    assert U.default(None, lambda: 15) == 15 and U.default(0, 10) == 0
    m1, m0, mh = U.prob_mask_like((10, 10), 1.0), U.prob_mask_like((10, 10), 0.0), U.prob_mask_like((10, 10), 0.5)
    assert m1.all() and not m0.any() and mh.shape == (10, 10) and mh.dtype == torch.bool
    np.testing.assert_allclose(U.unnormalize_img(torch.tensor([-1., 0., 1.])), [0., 0.5, 1.])
    g, n = U.clip_grad_norm({'a': torch.tensor([3.0, 4.0])}, 1.0, epsilon=1e-9)
    assert abs(n.item() - 5.0) < 1e-6 and abs(g['a'].norm().item() - 1.0) < 1e-5


def test_checkpoint_roundtrip_and_max_to_keep(tmp_path):
    from video_diffusion_nnx_amd.checkpoint import CheckpointManager, load_checkpoint, save_checkpoint
    from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion
    from video_diffusion_nnx_amd.unet3d import Unet3D
    unet = Unet3D(dim=16, rngs=1, channels=1, device='cpu')
    gd = GaussianDiffusion(unet, image_size=8, num_frames=2, channels=1, timesteps=10)
    mgr = CheckpointManager(tmp_path, max_to_keep=2)
    sd = unet.state_dict()
    ema = {k: v * 0.5 for k, v in sd.items()}
    for step in (2, 4, 5):
        save_checkpoint(mgr, sd, ema, step)
    assert mgr.all_steps() == [4, 5]
    unet2 = Unet3D(dim=16, rngs=2, channels=1, device='cpu')
    gd2 = GaussianDiffusion(unet2, image_size=8, num_frames=2, channels=1, timesteps=10)
    _, ema_l = load_checkpoint(gd2, 5, str(tmp_path))
    assert torch.equal(unet2.flat_params, unet.flat_params)
    load_checkpoint(gd2, 5, str(tmp_path), load_ema_params=True)
    assert torch.allclose(unet2.flat_params, unet.flat_params * 0.5)
    with pytest.raises(FileNotFoundError):
        load_checkpoint(gd2, 99, str(tmp_path))


def test_gif_writer_and_uint8(tmp_path):
    from PIL import Image
    from video_diffusion_nnx_amd.media import video_array_to_gif, videos_to_uint8
    v = np.random.default_rng(0).random((2, 1, 5, 8, 8)).astype(np.float32)
    u = videos_to_uint8(v)
    assert u.shape == (2, 5, 8, 8, 1) and u.dtype == np.uint8 and u.min() == 0 and u.max() == 255
    p = tmp_path / 's.gif'
    video_array_to_gif(u[0], p)
    im = Image.open(p)
    assert im.n_frames == 5 and im.size == (8, 8) and im.info.get('duration') == 120


def test_configs_parse_with_reference_schema():
    for name in ('config_v1_0', 'config_v2_2', 'config_v2_3', 'config_v2_2_northstar'):
        c = yaml.safe_load(open(f'configs/{name}.yaml'))
        assert set(c) == {'unet', 'diffusion', 'trainer'}
        assert {'dim', 'rngs_seed', 'dim_mults', 'channels', 'use_bert_text_cond'} <= set(c['unet'])
        assert {'image_size', 'num_frames', 'timesteps', 'loss_type', 'channels'} <= set(c['diffusion'])
    c = yaml.safe_load(open('configs/config_v2_2.yaml'))
    assert (c['unet']['dim'], c['diffusion']['num_frames'], c['trainer']['train_batch_size']) == (32, 10, 4)


def test_bench_parent_spawns_ranks_without_touching_torch(monkeypatch):
    """`python bench.py --gpus N` with no launcher: the parent must start N ranks through torch.distributed.run on 127.0.0.1
    and must not import torch (importing it is harmless, initialising HIP in the parent is not: the driver's contract)."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, subprocess, bench\n"
        "calls = []\n"
        "subprocess.call = lambda cmd, env=None: (calls.append((cmd, env)), 0)[1]\n"
        "sys.argv = ['bench.py', '--gpus', '4', '--steps', '7', '--warmup', '2']\n"
        "try:\n"
        "    bench.main()\n"
        "except SystemExit as e:\n"
        "    assert e.code == 0\n"
        "assert 'torch' not in sys.modules, 'the spawning parent imported torch'\n"
        "cmd, env = calls[0]\n"
        "assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and '--nproc-per-node=4' in cmd\n"
        "assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'\n"
        "assert cmd[-6:] == ['--gpus', '4', '--steps', '7', '--warmup', '2']\n"
        "assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'\n"
        "print('ok')\n")
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, '-c', code], cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith('ok'), out.stderr
