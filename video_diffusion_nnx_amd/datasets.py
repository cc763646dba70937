"""MovingMNIST .npy loader mirroring the reference's datasets.py:11-64 as it BEHAVES (no resize, no /255: values stay
raw floats, (F,B,H,W) -> per item (1, frames, H, W) with pad/truncate), plus a synthetic source for benchmarks.
NumPy only (torchvision is not needed: the reference builds a transform pipeline but never applies it)."""
from __future__ import annotations

import numpy as np
import torch
import torch.utils.data as data

from functools import partial

from .utils import cast_num_frames, identity


class MovingMNIST(data.Dataset):
    def __init__(self, file_path, image_size, channels=1, num_frames=20, horizontal_flip=False, force_num_frames=True):
        super().__init__()
        self.file_path, self.image_size, self.channels = file_path, image_size, channels
        self.channnels = channels                                     # the reference's attribute name (datasets.py:38)
        arr = np.load(file_path)                                      # (f, b, h, w)
        arr = np.transpose(arr, (1, 0, 2, 3))[:, None, ...]           # (b, 1, f, h, w)
        self.arrays = arr.astype(np.float32)
        # datasets.py:47-48: a functools.partial (the reference's tests read .keywords['frames']) or the identity
        self.cast_num_frames_fn = partial(cast_num_frames, frames=num_frames) if force_num_frames else identity
        self.cast = self.cast_num_frames_fn

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


class DevicePrefetcher:
    """Host -> device staging of the training batches (reference trainer.py:546-547 hands each host batch to a synchronous
    `device_put` inside the step).  One batch ahead: the NEXT batch's shard is copied into a pinned host buffer and
    sent to the GPU on a side stream while the current step runs, so `Trainer.train()` never waits on PCIe.  `next()` returns a
    device tensor whose copy the CURRENT stream has been made to wait for.  Without a GPU (CPU tests) it passes batches through.

    `select(batch) -> shard` picks this rank's slice (P('data', None), trainer.py:309)."""

    def __init__(self, iterator, device, select=None):
        self.it, self.device, self.select = iterator, torch.device(device), (select or (lambda b: b))
        self.on_gpu = self.device.type == 'cuda' and torch.cuda.is_available()
        self.stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self._pinned = [None, None]                                   # two pinned staging buffers, used alternately
        self._copied = [None, None]                                   # event behind the last H2D copy OUT of each buffer
        self._k = 0
        self._next = None
        self._stage()

    def _stage(self):
        try:
            host = self.select(torch.as_tensor(np.asarray(next(self.it)))).to(torch.float32)
        except StopIteration:
            self._next = None
            return
        if not self.on_gpu:
            self._next = (host.contiguous(), None)
            return
        buf = self._pinned[self._k]
        if buf is None or buf.shape != host.shape:
            buf = self._pinned[self._k] = torch.empty(host.shape, dtype=torch.float32).pin_memory()
        k = self._k
        self._k ^= 1
        if self._copied[k] is not None:
            self._copied[k].synchronize()                             # the DMA that read this buffer two batches ago has finished (almost always already true)
        buf.copy_(host)
        with torch.cuda.stream(self.stream):
            dev = buf.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._copied[k] = ev
        self._next = (dev, ev)

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        dev, ev = self._next
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)    # device-side wait only: the host does not block
            dev.record_stream(torch.cuda.current_stream(self.device))
        self._stage()                                                 # the following batch goes out while this one is consumed
        return dev
