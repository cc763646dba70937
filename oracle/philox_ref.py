"""ORACLE (test infrastructure only) -- NumPy restatement of the product's device RNG.

The reference draws noise with `jax.random.normal(key, shape)` (gaussian_diffusion.py:254,309,
416,445), i.e. JAX threefry, whose streams the new path does not (and need not) reproduce
(SURVEY.md §7 "DP semantic").  The product instead defines its own counter-based stream,
restated here so that sampling-loop parity can be checked end to end:

  Philox4x32-10 (Salmon et al., SC'11), key = (seed_lo, seed_hi),
  counter = (i_lo, i_hi, off_lo, off_hi) where i = element_index // 4 and `off` is the draw
  (subsequence) number; the four 32-bit outputs x0..x3 give four normals
      u_k = ((x_k >> 8) + 0.5) * 2^-24            (strictly inside (0,1), exact in fp32)
      z0, z1 = BoxMuller(u0, u1);  z2, z3 = BoxMuller(u2, u3)
      BoxMuller(a, b) = sqrt(-2 ln a) * (cos(2 pi b), sin(2 pi b))
  element e of a tensor takes z[e % 4] of counter e // 4.

Integer part is bit-exact by construction (known-answer vectors of Random123 are checked in
tests/test_oracle_philox.py); the float part differs from the device only by libm ulp effects.
"""
from __future__ import annotations

import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All arguments uint32 arrays (broadcastable)."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0 = np.asarray(k0, dtype=np.uint32)
    k1 = np.asarray(k1, dtype=np.uint32)
    c0, c1, c2, c3, k0, k1 = np.broadcast_arrays(c0, c1, c2, c3, k0, k1)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & MASK).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return c0, c1, c2, c3


def randn(n: int, seed: int, offset: int, dtype=np.float32) -> np.ndarray:
    """n standard normals of draw `offset` of stream `seed` (see module docstring)."""
    nctr = (n + 3) // 4
    i = np.arange(nctr, dtype=np.uint64)
    c0 = (i & MASK).astype(np.uint32)
    c1 = (i >> np.uint64(32)).astype(np.uint32)
    c2 = np.uint32(offset & 0xFFFFFFFF)
    c3 = np.uint32((offset >> 32) & 0xFFFFFFFF)
    k0 = np.uint32(seed & 0xFFFFFFFF)
    k1 = np.uint32((seed >> 32) & 0xFFFFFFFF)
    x = philox4x32_10(c0, c1, c2, c3, k0, k1)
    f = np.dtype(dtype).type
    u = [((xi >> np.uint32(8)).astype(dtype) + f(0.5)) * f(2.0 ** -24) for xi in x]
    out = np.empty((nctr, 4), dtype=dtype)
    two_pi = f(6.283185307179586)
    for j in (0, 2):
        r = np.sqrt(f(-2.0) * np.log(u[j]))
        th = two_pi * u[j + 1]
        out[:, j] = r * np.cos(th)
        out[:, j + 1] = r * np.sin(th)
    return out.reshape(-1)[:n]


def randint_below(n: int, bound: int, seed: int, offset: int) -> np.ndarray:
    """n integers in [0, bound): x0 of counter i, reduced by 64-bit multiply-shift (no modulo)."""
    i = np.arange(n, dtype=np.uint64)
    c0 = (i & MASK).astype(np.uint32)
    c1 = (i >> np.uint64(32)).astype(np.uint32)
    x0, _, _, _ = philox4x32_10(c0, c1, np.uint32(offset & 0xFFFFFFFF), np.uint32((offset >> 32) & 0xFFFFFFFF),
                                np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    return ((x0.astype(np.uint64) * np.uint64(bound)) >> np.uint64(32)).astype(np.int64)
