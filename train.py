"""Training CLI -- same flags and YAML schema as the reference's train.py (/root/reference/train.py:14-121):
    python train.py --config configs/config_v2_2.yaml [--resume_step N] [--rng_seed S]
Multi-GPU: `python -m torch.distributed.run --nproc-per-node N train.py --config ...` (one process per GPU, RCCL);
`train_batch_size` stays the GLOBAL batch, split over ranks as the reference splits it over devices (trainer.py:161-166).
Keys missing from a YAML fall back to the Trainer defaults (the reference raises KeyError there).
Extra (non-reference) flags: --mode {bf16,f32}, --train_num_steps (override, for smoke runs), --dataset_path."""
import argparse
import logging
import os
from pathlib import Path

import yaml


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format='%(levelname)s:%(name)s:%(message)s', force=True)
    parser = argparse.ArgumentParser(description='Train diffusion model')
    parser.add_argument('--config', type=str, default=str(Path(__file__).parent / 'configs' / 'config.yaml'), help='Path to the YAML config file')
    parser.add_argument('--resume_step', type=int, default=0, help='Step to resume training from')
    parser.add_argument('--rng_seed', type=int, default=None, help='RNG seed to use for training')
    parser.add_argument('--mode', choices=['bf16', 'f32'], default='bf16', help='MFMA operand precision (extension)')
    parser.add_argument('--train_num_steps', type=int, default=None, help='override trainer.train_num_steps (extension)')
    parser.add_argument('--dataset_path', type=str, default=None, help='override trainer.dataset_path, e.g. synthetic:64 (extension)')
    args = parser.parse_args(argv)

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        import torch.distributed as dist
        local = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))

    from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion
    from video_diffusion_nnx_amd.trainer import Trainer
    from video_diffusion_nnx_amd.unet3d import Rngs, Unet3D

    logging.info(f'Loading configuration from: {args.config}')
    with open(args.config) as f:
        config = yaml.safe_load(f)
    master_seed = args.rng_seed if args.rng_seed is not None else config.get('rng_seed', 0)
    logging.info(f'Using master RNG seed: {master_seed}')
    unet_cfg, diff_cfg, tc = config['unet'], config['diffusion'], dict(config['trainer'])
    unet_model = Unet3D(dim=unet_cfg['dim'], rngs=Rngs(unet_cfg['rngs_seed']), dim_mults=tuple(unet_cfg['dim_mults']),
                        channels=unet_cfg['channels'], use_bert_text_cond=unet_cfg['use_bert_text_cond'], mode=args.mode)
    diffusion_model = GaussianDiffusion(denoise_fn=unet_model, image_size=diff_cfg['image_size'], num_frames=diff_cfg['num_frames'],
                                        timesteps=diff_cfg['timesteps'], loss_type=diff_cfg['loss_type'], channels=diff_cfg['channels'])
    if args.train_num_steps is not None:
        tc['train_num_steps'] = args.train_num_steps
    if args.dataset_path is not None:
        tc['dataset_path'] = args.dataset_path
    tc.pop('resume_training_step', None)                       # the CLI flag wins, as in the reference (train.py:101)
    folder = tc.pop('folder')
    trainer = Trainer(diffusion_model=diffusion_model, folder=folder, resume_training_step=args.resume_step, rng_seed=master_seed, **tc)
    logging.info('Starting training...')
    trainer.train()
    logging.info('Training finished.')
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
