"""MovingMNIST .npy loader mirroring the reference's datasets.py:11-64 as it BEHAVES (no resize, no /255: values stay
raw floats, (F,B,H,W) -> per item (1, frames, H, W) with pad/truncate), plus a synthetic source for benchmarks.
NumPy only (torchvision is not needed: the reference builds a transform pipeline but never applies it)."""
from __future__ import annotations

import numpy as np
import torch
import torch.utils.data as data

from .utils import cast_num_frames, identity


class MovingMNIST(data.Dataset):
    def __init__(self, file_path, image_size, channels=1, num_frames=20, horizontal_flip=False, force_num_frames=True):
        super().__init__()
        self.file_path, self.image_size, self.channels = file_path, image_size, channels
        arr = np.load(file_path)                                      # (f, b, h, w)
        arr = np.transpose(arr, (1, 0, 2, 3))[:, None, ...]           # (b, 1, f, h, w)
        self.arrays = arr.astype(np.float32)
        self.cast = (lambda a: cast_num_frames(a, frames=num_frames)) if force_num_frames else identity

    def __len__(self):
        return self.arrays.shape[0]

    def __getitem__(self, index):
        return self.cast(self.arrays[index])


class SyntheticVideo(data.Dataset):
    """`dataset_path: synthetic:N` -> N seeded uniform [0,1) videos of shape (channels, frames, size, size)."""

    def __init__(self, n, channels, num_frames, image_size, seed=0):
        self.n, self.shape, self.seed = int(n), (channels, num_frames, image_size, image_size), seed

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1000003 + int(index))
        return torch.rand(self.shape, generator=g).numpy()
