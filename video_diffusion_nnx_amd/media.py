"""GIF writer (reference utils.py:343-373 as its caller uses it: (F, H, W, C) uint8 frames, 120 ms, loop 0).  PIL only."""
from __future__ import annotations

import numpy as np

CHANNELS_TO_MODE = {1: 'L', 3: 'RGB', 4: 'RGBA'}


def video_array_to_gif(arr, path, duration=120, loop=0, optimize=True):
    from PIL import Image
    arr = np.asarray(arr)
    assert arr.ndim == 4, 'expected (frames, height, width, channels)'
    frames = []
    for fr in arr:
        c = fr.shape[-1]
        img = Image.fromarray(fr[..., 0] if c == 1 else fr, mode=CHANNELS_TO_MODE[c])
        frames.append(img)
    first, *rest = frames
    first.save(str(path), save_all=True, append_images=rest, duration=duration, loop=loop, optimize=optimize)
    return frames


def videos_to_uint8(videos, lo_hi=None):
    """sample.py:106-110: 'b c f h w -> b f h w c', min-max over the WHOLE batch (Q19), * 255 -> uint8.  lo_hi: the extrema of the
    GLOBAL batch when `videos` is one rank's shard of it."""
    v = np.asarray(videos).transpose(0, 2, 3, 4, 1)
    lo, hi = (v.min(), v.max()) if lo_hi is None else lo_hi
    return ((v - lo) / (hi - lo) * 255).astype(np.uint8)
