"""CPU tests of the product's GaussianDiffusion host logic (tables, closed forms, key splitting) against the oracle
and the reference's known answers (gaussian_diffusion_test.py:75-158)."""
import numpy as np
import pytest
import torch

from oracle.diffusion_ref import DiffusionRef, schedule


@pytest.fixture(scope='module')
def gd():
    from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion
    from video_diffusion_nnx_amd.unet3d import Unet3D
    unet = Unet3D(dim=16, rngs=0, channels=3, device='cpu')
    return GaussianDiffusion(unet, image_size=8, num_frames=2, channels=3, timesteps=10, loss_type='l1')


def test_tables_bit_equal_to_oracle(gd):
    from video_diffusion_nnx_amd.gaussian_diffusion import TABLE_NAMES, make_tables
    for T in (10, 200, 1000):
        a, b = make_tables(T), schedule(T, np.float32)
        for n in TABLE_NAMES:
            np.testing.assert_array_equal(a[n], b[n])
    for n in TABLE_NAMES:
        assert getattr(gd, n).shape == (10,)


def test_closed_forms_match_oracle_and_reference_known_answers(gd):
    ref = DiffusionRef(None, image_size=8, num_frames=2, channels=3, timesteps=10)
    x = torch.ones(2, 3, 2, 8, 8)
    t0 = torch.zeros(2, dtype=torch.int64)
    mean0, var0, _ = gd.q_mean_variance(x, t0)
    np.testing.assert_allclose(mean0, gd.sqrt_alphas_cumprod[0] * x, atol=1e-6)
    np.testing.assert_allclose(var0, torch.full((2, 1, 1, 1, 1), 1.0 - gd.alphas_cumprod[0].item()), atol=1e-6)
    t = torch.tensor([0, 5])
    g = torch.Generator().manual_seed(0)
    n = torch.randn(x.shape, generator=g)
    for a, b in zip(gd.q_posterior(x, n, t), ref.q_posterior(x, n, t)):
        np.testing.assert_allclose(a, b, atol=1e-7)
    np.testing.assert_allclose(gd.predict_start_from_noise(n, t, x), ref.predict_start_from_noise(n, t, x), atol=1e-6)
    assert gd.q_posterior(x, n, t)[1].shape == (2, 1, 1, 1, 1)


def test_helpers():
    from video_diffusion_nnx_amd.gaussian_diffusion import extract, is_list_str, split_key
    out = extract(torch.arange(10), torch.tensor([1, 3, 5]), (3, 10, 10, 10))          # utils_test.py:102-110
    assert out.shape == (3, 1, 1, 1) and out.flatten().tolist() == [1, 3, 5]
    assert is_list_str(['a', 'b']) and is_list_str(()) and not is_list_str(['a', 1]) and not is_list_str('a')   # :133-143
    ks = split_key(7, 3)
    assert len(set(ks)) == 3 and ks == split_key(7, 3) and ks != split_key(8, 3)


def test_shape_check_and_loss_type(gd):
    with pytest.raises(AssertionError):
        gd(torch.zeros(2, 3, 2, 8, 9), 0)


def test_shard_key_and_rank_world_outside_a_group():
    """Data-parallel sampling (reference gaussian_diffusion.py:278-298): rank r's Philox seed depends on (key, r) only; one rank keeps the key."""
    from video_diffusion_nnx_amd.gaussian_diffusion import dist_rank_world, shard_key, split_key
    assert dist_rank_world() == (0, 1)
    assert shard_key(5, 0, 1) == 5
    ks = [shard_key(5, r, 4) for r in range(4)]
    assert len(set(ks)) == 4 and ks[:2] == [shard_key(5, r, 2) for r in range(2)]
    assert ks[2] == split_key(5, 3)[-1]
