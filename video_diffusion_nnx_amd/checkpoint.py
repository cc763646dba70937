"""Checkpoints: {'model': params, 'ema_params': ema} per step (schema of reference utils.py:432-508), stored as
safetensors (the reference's Orbax/tensorstore files cannot be read or written without orbax).
Keys are the nnx state-tree paths: 'model.denoise_fn.<unet path>' / 'ema_params.denoise_fn.<unet path>'."""
from __future__ import annotations

import logging
import os
import shutil
from typing import Dict, Optional

import torch


class CheckpointManager:
    """Minimal stand-in for orbax.CheckpointManager(dir, max_to_keep, create=True) (trainer.py:271-272)."""

    def __init__(self, directory, max_to_keep: Optional[int] = None):
        self.directory = str(directory)
        self.max_to_keep = max_to_keep
        os.makedirs(self.directory, exist_ok=True)

    def all_steps(self):
        steps = []
        for d in os.listdir(self.directory):
            if d.isdigit() and os.path.exists(os.path.join(self.directory, d, 'state.safetensors')):
                steps.append(int(d))
        return sorted(steps)

    def path(self, step: int) -> str:
        return os.path.join(self.directory, str(step), 'state.safetensors')

    def save(self, step: int, tensors: Dict[str, torch.Tensor]):
        from safetensors.torch import save_file
        os.makedirs(os.path.dirname(self.path(step)), exist_ok=True)
        tmp = self.path(step) + '.tmp'
        save_file({k: v.detach().cpu().contiguous() for k, v in tensors.items()}, tmp)
        os.replace(tmp, self.path(step))                      # force=True semantics: overwrite an existing step
        if self.max_to_keep:
            for old in self.all_steps()[:-self.max_to_keep]:
                shutil.rmtree(os.path.join(self.directory, str(old)), ignore_errors=True)

    def restore(self, step: int) -> Dict[str, torch.Tensor]:
        from safetensors.torch import load_file
        p = self.path(step)
        if not os.path.exists(p):
            raise FileNotFoundError(p)
        return load_file(p)


def _prefixed(prefix: str, sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {f'{prefix}.denoise_fn.{k}': v for k, v in sd.items()}


def save_checkpoint(ckpt_manager: CheckpointManager, model_params: Dict[str, torch.Tensor], ema_params: Dict[str, torch.Tensor], step: int):
    """reference utils.py:432-458 (same argument order)."""
    ckpt_manager.save(step, {**_prefixed('model', model_params), **_prefixed('ema_params', ema_params)})
    logging.info(f'Checkpoint saved at step {step}')


def load_checkpoint(model, step: int, path: str, ckpt_manager: Optional[CheckpointManager] = None, load_ema_params: bool = False):
    """reference utils.py:460-508: returns (model, ema_params); `model` is a GaussianDiffusion whose denoise_fn is loaded."""
    if ckpt_manager is None:
        ckpt_manager = CheckpointManager(path)
    flat = ckpt_manager.restore(step)
    pick = lambda pre: {k[len(pre) + len('.denoise_fn.'):]: v for k, v in flat.items() if k.startswith(pre + '.denoise_fn.')}
    model_params, ema_params = pick('model'), pick('ema_params')
    model.denoise_fn.load_state_dict(ema_params if load_ema_params else model_params)
    logging.info('Loaded EMA parameters' if load_ema_params else 'Loaded model parameters')
    logging.info(f'Checkpoint loaded from step: {step}')
    return model, ema_params
