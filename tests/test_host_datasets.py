"""CPU tests of the data path: the MovingMNIST loader's known answers, lifted from the reference's own test file
(/root/reference/test_datasets.py:10-106: 5 sequences of 15 frames of 32 x 32 written as (F, B, H, W); len 5; item
(1, frames, 32, 32); pad to 25; truncate to 10; untouched when force_num_frames=False), and the batch prefetcher."""
import numpy as np
import pytest
import torch


@pytest.fixture
def npy(tmp_path):
    """test_datasets.py:12-23: (15, 5, 32, 32) random data saved as uint8."""
    path = tmp_path / 'test_mnist_data.npy'
    rng = np.random.default_rng(0)
    data = (rng.random((15, 5, 32, 32)) * 255).astype(np.uint8)
    np.save(path, data)
    return str(path), data


def test_initialization_and_len(npy):                                    # test_datasets.py:35-52
    from video_diffusion_nnx_amd.datasets import MovingMNIST
    path, _ = npy
    ds = MovingMNIST(file_path=path, image_size=64, num_frames=20, channels=1, force_num_frames=True)
    assert len(ds) == 5
    assert ds.image_size == 64
    assert ds.channnels == 1                                             # the reference's (misspelt) attribute
    assert ds.cast_num_frames_fn.keywords['frames'] == 20
    assert len(MovingMNIST(file_path=path, image_size=64)) == 5


def test_getitem_raw_shape_and_type(npy):                                # test_datasets.py:54-69
    from video_diffusion_nnx_amd.datasets import MovingMNIST
    path, data = npy
    item = MovingMNIST(file_path=path, image_size=64, num_frames=20, force_num_frames=True)[0]
    assert isinstance(item, np.ndarray)
    assert item.shape == (1, 20, 32, 32)                                 # (C, F, H, W): original spatial size (no resize is applied)
    assert item.dtype == np.float32
    # values stay raw [0, 255] floats, frames keep their order, the padding is zeros
    np.testing.assert_array_equal(item[0, :15], data[:, 0].astype(np.float32))
    assert (item[0, 15:] == 0).all()


def test_frame_casting(npy):                                             # test_datasets.py:71-106
    from video_diffusion_nnx_amd.datasets import MovingMNIST
    path, data = npy
    assert MovingMNIST(file_path=path, image_size=64, num_frames=25, force_num_frames=True)[0].shape[1] == 25      # padded
    short = MovingMNIST(file_path=path, image_size=64, num_frames=10, force_num_frames=True)[3]
    assert short.shape[1] == 10                                          # truncated: the first 10 frames
    np.testing.assert_array_equal(short[0], data[:10, 3].astype(np.float32))
    assert MovingMNIST(file_path=path, image_size=64, num_frames=20, force_num_frames=False)[0].shape[1] == 15     # untouched


def test_prefetcher_passes_every_batch_in_order_on_cpu():
    from video_diffusion_nnx_amd.datasets import DevicePrefetcher
    batches = [torch.full((4, 1, 2, 3, 3), float(i)) for i in range(5)]
    got = list(DevicePrefetcher(iter(batches), 'cpu', select=lambda b: b[1:3]))
    assert len(got) == 5
    for i, g in enumerate(got):
        assert g.shape == (2, 1, 2, 3, 3) and g.dtype == torch.float32 and (g == i).all()


def test_trainer_feeds_the_rank_shard_through_the_prefetcher(tmp_path, monkeypatch):
    from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion
    from video_diffusion_nnx_amd.trainer import Trainer
    from video_diffusion_nnx_amd.unet3d import Unet3D
    unet = Unet3D(dim=16, rngs=0, channels=1, device='cpu')
    gd = GaussianDiffusion(unet, image_size=8, num_frames=2, channels=1, timesteps=10)
    tr = Trainer(gd, str(tmp_path), dataset_path='synthetic:8', train_batch_size=4, train_num_steps=3,
                 checkpoint_every_steps=100, results_folder=str(tmp_path / 'res'))
    seen = []
    monkeypatch.setattr(tr, 'train_step', lambda batch, step: (seen.append(tuple(batch.shape)), torch.tensor(1.0))[1])
    monkeypatch.setattr(tr, '_save', lambda step: None)
    tr.train()
    assert seen == [(4, 1, 2, 8, 8)] * 3
