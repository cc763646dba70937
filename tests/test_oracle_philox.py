"""Pins oracle/philox_ref.py with the Random123 known-answer vectors for philox4x32-10."""
import numpy as np

from oracle.philox_ref import philox4x32_10, randint_below, randn


def _hex(t):
    return [int(v) for v in t]


def test_random123_kat():
    assert _hex(philox4x32_10(0, 0, 0, 0, 0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xffffffff
    assert _hex(philox4x32_10(f, f, f, f, f, f)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _hex(philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_randn_moments_and_determinism():
    z = randn(1 << 18, seed=7, offset=3)
    assert z.dtype == np.float32 and z.shape == (1 << 18,)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    np.testing.assert_array_equal(z[:1001], randn(1001, seed=7, offset=3))   # prefix-stable, ragged n
    assert not np.array_equal(z[:16], randn(16, seed=7, offset=4)[:16])
    assert randn(0, 1, 1).shape == (0,)


def test_randint_bounds():
    r = randint_below(10000, 1000, seed=5, offset=0)
    assert r.min() >= 0 and r.max() < 1000 and len(np.unique(r)) > 900
