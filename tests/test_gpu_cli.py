"""End-to-end CLI run on the GPU: train.py a few steps on synthetic data (checkpoint written at the reference's cadence),
then sample.py from that checkpoint (reference train.py:23-42, sample.py:19-62).  Tiny shapes; checks artefacts only —
numerical parity of the pieces is covered by test_gpu_unet / test_gpu_diffusion / test_gpu_train."""
import json
import pathlib
import sys

import pytest
import yaml

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

pytestmark = pytest.mark.gpu


def tiny_cfg(tmp):
    return {
        'rng_seed': 3,
        'unet': dict(dim=16, dim_mults=[1, 2], channels=1, rngs_seed=0, use_bert_text_cond=False),
        'diffusion': dict(image_size=16, num_frames=4, channels=1, timesteps=6, loss_type='l2'),
        'trainer': dict(folder=str(tmp / 'res'), dataset_path='synthetic:8', num_frames=4, train_batch_size=2, train_lr=1e-3,
                        train_num_steps=3, gradient_accumulate_every=2, step_start_ema=1, update_ema_every=1,
                        save_and_sample_every=100, checkpoint_every_steps=2, results_folder=str(tmp / 'res'),
                        checkpoint_dir_path=str(tmp / 'ckpt'), tensorboard_dir=str(tmp / 'tb'), max_to_keep=5,
                        add_loss_plot=False, lr_decay_start_step=10, lr_decay_steps=10, lr_decay_coeff=0.1,
                        max_grad_norm=1e7, num_sample_rows=1, cond_scale=2.0, sample_text=None, use_path_as_cond=False,
                        resume_training_step=0),
    }


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_train_then_sample(tmp_path, mode):
    import sample
    import train
    cfg_path = tmp_path / 'cfg.yaml'
    cfg_path.write_text(yaml.safe_dump(tiny_cfg(tmp_path)))
    train.main(['--config', str(cfg_path), '--mode', mode])
    ckpts = sorted(p.name for p in (tmp_path / 'ckpt').iterdir())
    assert ckpts, 'no checkpoint written'
    scalars = [json.loads(l) for l in (tmp_path / 'tb' / 'scalars_rank0.jsonl').read_text().splitlines()]
    losses = [s['value'] for s in scalars if s['tag'] == 'loss/train']
    assert len(losses) == 3 and all(0 < v < 10 for v in losses)
    step = 2
    out = tmp_path / 'gifs'
    sample.main(['--config', str(cfg_path), '--checkpoint-path', str(tmp_path / 'ckpt'), '--step', str(step),
                 '--batch-size', '2', '--seed', '1', '--output-path', str(out), '--mode', mode, '--load-ema-params'])
    gifs = sorted(out.glob('sample_*.gif'))
    assert len(gifs) == 2 and all(g.stat().st_size > 100 for g in gifs)
