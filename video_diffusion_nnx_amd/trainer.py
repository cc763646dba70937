"""Trainer -- host-side mirror of the reference class (/root/reference/trainer.py:52-628).

Same constructor signature and `.train(prob_focus_present, focus_present_mask, log_fn)` entry.  One train step
(reference `_pjit_train_step`, trainer.py:322-392) = q_sample -> Unet3D forward -> loss -> backward -> gradient
all-reduce -> Adam -> EMA, all on the device through libvdx.so; data parallelism = one process per GPU with
`torch.distributed` (RCCL), the flat gradient buffer reduced in buckets in reverse-layer order so the
all-reduce of early buckets overlaps the rest of the backward.
Accepted-and-unused arguments are the reference's own: folder, num_frames, gradient_accumulate_every,
save_and_sample_every, num_sample_rows, max_grad_norm, sample_text, cond_scale, add_loss_plot.
"""
from __future__ import annotations

import json
import logging
import math
import os
import time
from pathlib import Path
from typing import List, Optional, Tuple

import numpy as np
import torch

from .checkpoint import CheckpointManager, load_checkpoint, save_checkpoint
from .utils import cycle, noop


def lr_schedule(step: int, train_lr: float, lr_decay_start_step: int = 0, lr_decay_steps: int = 0, lr_decay_coeff: float = 1.0) -> float:
    """optax.piecewise_interpolate_schedule('cosine', train_lr, {start: 1.0, start + steps: coeff}) (trainer.py:138-145).
    The dict literal's duplicate key collapses when steps == 0; a zero-length interval contributes nothing."""
    bs = {lr_decay_start_step: 1.0}
    bs[lr_decay_start_step + lr_decay_steps] = lr_decay_coeff
    items = sorted(bs.items())
    bounds = [0] + [b for b, _ in items]
    values = [train_lr]
    for _, s in items:
        values.append(values[-1] * s)
    for i in range(len(bounds) - 1):
        lo, hi = bounds[i], bounds[i + 1]
        if lo <= step < hi:
            pct = (step - lo) / (hi - lo)
            return values[i + 1] + (values[i] - values[i + 1]) / 2.0 * (math.cos(math.pi * pct) + 1.0)
    return values[-1]


def make_buckets(param_table, total: int, stage_of, n_stages: int, min_bucket_floats: int = 4 << 20) -> List[Tuple[int, int, int]]:
    """Partition the flat gradient buffer into contiguous (lo, hi, ready_stage) buckets.

    Backward visits stages n_stages-1 .. 0; parameters are laid out in forward order, so a bucket [lo, hi) is complete once
    the backward has finished the LOWEST stage it contains.  Buckets are listed in the order they become ready."""
    edges = []            # (offset, stage) per tensor, forward order
    for name, shape, off in param_table:
        edges.append((off, stage_of(name)))
    buckets = []
    hi = total
    cur_stage = None
    for off, st in reversed(edges):
        cur_stage = st if cur_stage is None else min(cur_stage, st)
        if hi - off >= min_bucket_floats:
            buckets.append((off, hi, cur_stage))
            hi, cur_stage = off, None
    if hi > 0:
        buckets.append((0, hi, 0 if cur_stage is None else min(cur_stage, 0)))
    return buckets


class GradBucketReducer:
    """Sum all-reduce of a flat gradient tensor in buckets (async), averaged by the caller (1/world in the Adam read)."""

    def __init__(self, flat_grads: torch.Tensor, buckets, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.flat, self.buckets, self.group = flat_grads, list(buckets), group
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.enabled else 1
        self._work, self._next = [], 0

    def stage_done(self, stage: int):
        """Called after the backward has enqueued stage `stage`: launches every bucket that became complete."""
        while self._next < len(self.buckets) and self.buckets[self._next][2] >= stage:
            lo, hi, _ = self.buckets[self._next]
            if self.enabled:
                self._work.append(self.dist.all_reduce(self.flat[lo:hi], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._next += 1

    def finish(self):
        self.stage_done(-1)
        for w in self._work:
            w.wait()
        self._work, self._next = [], 0


class AbiBucketReducer:
    """GradBucketReducer's contract with the collective behind the C ABI: each finished bucket goes through vdx_allreduce_bucket
    (RCCL communicator owned by the vdx handle, csrc/comm.cpp) on a side stream that waits for the stage's kernels through an event;
    the caller's stream waits for the side stream in finish().  SURVEY 8(b): vdx_comm_init / vdx_allreduce_bucket."""

    def __init__(self, flat_grads: torch.Tensor, buckets, handle, world: int, comm_stream):
        self.flat, self.buckets, self.h, self.world, self.cs = flat_grads, list(buckets), handle, world, comm_stream
        self.enabled = True
        self._next = 0

    def stage_done(self, stage: int):
        from .unet3d import vdx_allreduce_bucket
        from . import _lib as L
        ready = []
        while self._next < len(self.buckets) and self.buckets[self._next][2] >= stage:
            ready.append(self.buckets[self._next]); self._next += 1
        if not ready or not self.enabled:
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.flat.device))
        self.cs.wait_event(ev)
        for lo, hi, _ in ready:
            L.check(vdx_allreduce_bucket(self.h.ptr, self.flat.data_ptr() + 4 * lo, hi - lo, self.cs.cuda_stream))

    def finish(self):
        self.stage_done(-1)
        torch.cuda.current_stream(self.flat.device).wait_stream(self.cs)
        self._next = 0


class Trainer:
    # 'torch': torch.distributed.all_reduce per bucket (RCCL under the nccl backend; gloo in the CPU / one-GPU tests).
    # 'abi': the communicator inside libvdx.so (vdx_comm_init / vdx_allreduce_bucket); needs a CUDA device.  VDX_COMM overrides.
    comm_backend = os.environ.get('VDX_COMM', 'torch')
    # mode='bf16' networks: the training forward stores its inter-kernel activations as bf16 (vdx_set_activation_storage(h, 2)) -- the
    # forward then runs on the sampling path's kernels and every activation read of the backward halves; False = fp32 slots
    train_act_bf16 = True
    # gradient all-reduce bucket size (floats): >= 16 MB per RCCL call keeps every xGMI ring step bandwidth-bound; the
    # constructor signature stays the reference's, so this is a class attribute
    min_bucket_floats = 4 << 20

    def __init__(self, diffusion_model, folder: str, *, rng_seed: int = 0, dataset_path: str, num_frames: int = 16,
                 train_batch_size: int = 4, train_lr: float = 1e-4, train_num_steps: int = 100000,
                 gradient_accumulate_every: int = 2, step_start_ema: int = 2000, update_ema_every: int = 10,
                 save_and_sample_every: int = 100000, results_folder: str = './results', num_sample_rows: int = 4,
                 max_grad_norm: Optional[float] = None, use_path_as_cond: bool = False, sample_text: Optional[str] = None,
                 cond_scale: float = 2.0, checkpoint_every_steps: int = 10, checkpoint_dir_path: str = '',
                 add_loss_plot: bool = False, tensorboard_dir: str = '', resume_training_step: int = 0, ema_decay: float = 0.9999,
                 max_to_keep: Optional[int] = None, lr_decay_start_step: int = 0, lr_decay_steps: int = 0, lr_decay_coeff: float = 1.0,
                 profile_flush_step: int = 100, num_model_shards: int = 1):
        import torch.distributed as dist
        assert num_model_shards == 1, 'only data parallelism is supported (the GSPMD "model" axis of trainer.py:407-426 is out of scope)'
        self.model = diffusion_model
        self.unet = diffusion_model.denoise_fn
        self.device = self.unet.device
        self.rng_seed = int(rng_seed)
        self.step_start_ema, self.update_ema_every, self.ema_decay = step_start_ema, update_ema_every, ema_decay
        self.train_lr, self.lr_args = train_lr, (lr_decay_start_step, lr_decay_steps, lr_decay_coeff)
        self.train_num_steps, self.max_grad_norm = train_num_steps, max_grad_norm
        self.use_path_as_cond, self.gradient_accumulate_every = use_path_as_cond, gradient_accumulate_every
        self.profile_flush_step = profile_flush_step
        # ---- data-parallel layout (trainer.py:161-177): the reference's batch is GLOBAL and split over devices ----
        self.dist_on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.dist_on else 1
        self.rank = dist.get_rank() if self.dist_on else 0
        assert train_batch_size % self.world == 0, 'batch_size must be divisible by number of devices'     # trainer.py:163
        if self.dist_on and self.device.type == 'cuda':
            # parameters, gradients and workspaces must live on the GPU this rank's process group and HIP stream are bound to
            assert self.device.index == torch.cuda.current_device(), \
                f'Unet3D lives on {self.device} but this rank runs on cuda:{torch.cuda.current_device()}: build the model after torch.cuda.set_device(LOCAL_RANK)'
        self.batch_size = train_batch_size
        self.per_device_bs = train_batch_size // self.world
        # ---- optimizer state: Adam m, v + EMA copy, flat like the parameters ----
        n = self.unet.flat_params.numel()
        self.m = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.v = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.ema = self.unet.flat_params.clone()
        self.grads = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.opt_count = 0                                       # optax count: restarts at 0 on resume
        from .train_step import stage_of_param
        nlev = len(self.unet.dim_mults)
        self.buckets = make_buckets(self.unet.param_table, n, lambda nm: stage_of_param(nm, nlev), stage_of_param('__count__', nlev),
                                    min_bucket_floats=self.min_bucket_floats)
        # ---- dataset ----
        self.image_size = diffusion_model.image_size
        if str(dataset_path).startswith('synthetic'):
            from .datasets import SyntheticVideo
            n_items = int(str(dataset_path).split(':')[1]) if ':' in str(dataset_path) else 64
            self.ds = SyntheticVideo(n_items, diffusion_model.channels, diffusion_model.num_frames, self.image_size, seed=self.rng_seed)
        else:
            from .datasets import MovingMNIST
            self.ds = MovingMNIST(dataset_path, image_size=(self.image_size, self.image_size), num_frames=diffusion_model.num_frames,
                                  force_num_frames=True)
        assert len(self.ds) > 0, 'Dataset is empty. Check path and format.'
        g = torch.Generator().manual_seed(self.rng_seed)
        self.dl = cycle(torch.utils.data.DataLoader(self.ds, batch_size=self.batch_size, shuffle=True, drop_last=True, generator=g))
        # ---- results / checkpoints / logs ----
        self.results_folder = Path(results_folder).resolve()
        self.results_folder.mkdir(exist_ok=True, parents=True)
        self.checkpoint_dir_path = Path(checkpoint_dir_path).resolve() if checkpoint_dir_path else (self.results_folder / 'checkpoints').resolve()
        self.checkpoint_every_steps = checkpoint_every_steps
        self.ckpt_manager = CheckpointManager(self.checkpoint_dir_path, max_to_keep=max_to_keep) if self.rank == 0 else None
        self.log_dir = Path(tensorboard_dir).resolve() if tensorboard_dir else self.results_folder / 'tensorboard'
        self.log_dir.mkdir(exist_ok=True, parents=True)
        self._scalars = open(self.log_dir / f'scalars_rank{self.rank}.jsonl', 'a') if self.rank == 0 else None
        # ---- resume (params + EMA only; optimizer state re-initialised, as the reference) ----
        self.step = resume_training_step
        if self.step > 0:
            try:
                _, ema_params = load_checkpoint(self.model, self.step, str(self.checkpoint_dir_path))
                sd = self.unet.state_dict()
                for k, vv in ema_params.items():
                    shape, off = self.unet._index[k]
                    self.ema[off:off + vv.numel()] = vv.reshape(-1).to(self.device)
                logging.info(f'Successfully loaded checkpoint state for step {self.step}')
            except FileNotFoundError:
                logging.warning(f'Checkpoint for step {self.step} not found at {self.checkpoint_dir_path}.')
                self.step = 0

    # ------------------------------------------------------------------------------------------------
    def _scalar(self, tag, value, step):
        if self._scalars:
            self._scalars.write(json.dumps({'tag': tag, 'value': float(value), 'step': int(step)}) + '\n')

    def current_lr(self, count: int) -> float:
        return lr_schedule(count, self.train_lr, *self.lr_args)

    def _save(self, step: int):
        if self.ckpt_manager is None:
            return
        try:
            names = [(n, s, o) for n, s, o in self.unet.param_table]
            ema_sd = {n: self.ema[o:o + int(np.prod(s))].view(*s) for n, s, o in names}
            save_checkpoint(self.ckpt_manager, self.unet.state_dict(), ema_sd, step)
        except Exception as e:                                   # reference: log and continue (trainer.py:595-602)
            logging.error(f'Error saving checkpoint at step {step}: {e}')

    def _abi_comm(self):
        """Joins (once) the RCCL communicator of the training handle: rank 0 creates the unique id, the group's own channel carries it."""
        if getattr(self, '_abi', None) is None:
            import ctypes as C
            import torch.distributed as dist
            from .unet3d import vdx_comm_init, vdx_comm_unique_id
            from . import _lib as L
            h = self.unet.handle(self.model.num_frames, self.model.image_size)
            buf = C.create_string_buffer(128)
            if self.rank == 0:
                L.check(vdx_comm_unique_id(buf))
            box = [bytes(buf.raw)]
            if self.dist_on and self.world > 1:
                dist.broadcast_object_list(box, src=0)
            L.check(vdx_comm_init(h.ptr, self.rank, self.world, box[0]))
            self._abi = (h, torch.cuda.Stream(device=self.device))
        return self._abi

    def make_reducer(self):
        if self.comm_backend == 'abi':
            h, cs = self._abi_comm()
            red = AbiBucketReducer(self.grads, self.buckets, h, self.world, cs)
        else:
            red = GradBucketReducer(self.grads, self.buckets)
        if not getattr(self, 'comm_enabled', True):              # timing aid (bench.py: the same step without the collective)
            red.enabled = False
        return red

    def train_step(self, batch: torch.Tensor, step: int, t=None, noise=None) -> torch.Tensor:
        """One `_pjit_train_step` on this rank's shard of the batch.  Returns the (device) scalar loss of the shard.
        t / noise: optional explicit timesteps / noise of the shard (default: this rank's own Philox draws)."""
        from .train_step import run_train_step
        return run_train_step(self, batch, step, t=t, noise=noise)

    def train(self, prob_focus_present: float = 0.0, focus_present_mask=None, log_fn=noop):
        assert callable(log_fn)
        logging.info(f'Starting training loop from step {self.step}...')
        import torch.distributed as dist
        from .datasets import DevicePrefetcher
        lo, hi = self.rank * self.per_device_bs, (self.rank + 1) * self.per_device_bs
        # this rank's shard of every batch (P('data', None), trainer.py:309), staged one batch ahead on a side stream
        shards = DevicePrefetcher(self.dl, self.device, select=lambda b: b[lo:hi])
        # trace hook (reference trainer.py:524-533,607: jax.profiler trace from the first step until profile_flush_step): every step up to
        # profile_flush_step is one roctx range ("train_step N", via torch.cuda.nvtx = roctx on ROCm), so `rocprofv3 --marker-trace
        # --kernel-trace -- python train.py ...` groups the kernels per step; no-op without the marker library
        def _range(push, name=''):
            try:
                import torch.cuda.nvtx as nvtx
                nvtx.range_push(name) if push else nvtx.range_pop()
            except Exception:                                    # marker library absent: tracing is optional
                pass
        first_step = self.step
        while self.step < self.train_num_steps:
            shard = next(shards)
            t0 = time.time()
            traced = self.device.type == 'cuda' and self.step - first_step < max(0, int(self.profile_flush_step))
            if traced:
                _range(True, f'train_step {self.step}')
            loss = self.train_step(shard, self.step)
            if traced:
                _range(False)
            if self.dist_on and self.world > 1:                  # global mean loss = mean of equal-size shard means (C2)
                dist.all_reduce(loss, op=dist.ReduceOp.SUM)
                loss = loss / self.world
            current_loss = float(loss.item())                    # device -> host sync every step, as trainer.py:581
            self._scalar('step_time', time.time() - t0, self.step)
            logging.info(f'Step: {self.step}/{self.train_num_steps} | Loss: {current_loss:.4f}')
            log_fn({'loss': current_loss, 'step': self.step})
            self._scalar('loss/train', current_loss, self.step)
            self._scalar('lr/train', self.current_lr(self.step), self.step)
            if self.step > 0 and self.step % self.checkpoint_every_steps == 0:     # cadence: before step += 1 (trainer.py:593,604)
                logging.info(f'Step: {self.step} | Saving checkpoint...')
                self._save(self.step)
            self.step += 1
        logging.info('Training completed!')
        logging.info('Saving final checkpoint...')
        self._save(self.step)                                    # final save at train_num_steps (trainer.py:619-622)
        if self._scalars:
            self._scalars.flush()
