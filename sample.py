"""Sampling CLI -- same flags and YAML schema as the reference's sample.py (/root/reference/sample.py:17-119):
    python sample.py --checkpoint-path DIR --step N --output-path OUT --config configs/config_v2_2.yaml
                     [--seed S] [--batch-size B] [--load-ema-params]
Extra (non-reference) flags: --mode {bf16,f32}, --random-init (sample from un-trained weights, no checkpoint)."""
import argparse
import logging
from pathlib import Path

import numpy as np
import yaml

logging.basicConfig(level=logging.INFO, force=True)


def main(argv=None):
    parser = argparse.ArgumentParser(description='Generate samples using diffusion model')
    parser.add_argument('--config', type=str, default=str(Path(__file__).parent / 'configs' / 'config.yaml'), help='Path to the YAML config file')
    parser.add_argument('--output-path', type=str, default=str(Path(__file__).parent / 'outputs'), help='Directory to save sampled GIFs')
    parser.add_argument('--checkpoint-path', type=str, required=False, default=None, help='Path to the model checkpoint directory')
    parser.add_argument('--step', type=int, default=0, help='Checkpoint step number to load')
    parser.add_argument('--seed', type=int, default=0, help='Random seed for sampling')
    parser.add_argument('--batch-size', type=int, default=2, help='Number of videos to generate')
    parser.add_argument('--load-ema-params', action='store_true', default=False, help='Whether to load EMA parameters')
    parser.add_argument('--mode', choices=['bf16', 'f32'], default='bf16', help='MFMA operand precision (extension)')
    parser.add_argument('--random-init', action='store_true', help='skip checkpoint loading (extension, for smoke runs)')
    parser.add_argument('--timesteps', type=int, default=None, help='override diffusion.timesteps (extension, for smoke runs)')
    args = parser.parse_args(argv)
    if not args.random_init and args.checkpoint_path is None:
        parser.error('--checkpoint-path is required (or pass --random-init)')

    from video_diffusion_nnx_amd.checkpoint import load_checkpoint
    from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion
    from video_diffusion_nnx_amd.media import video_array_to_gif, videos_to_uint8
    from video_diffusion_nnx_amd.unet3d import Rngs, Unet3D

    output_path = Path(args.output_path)
    output_path.mkdir(parents=True, exist_ok=True)
    logging.info(f'Loading configuration from: {args.config}')
    with open(args.config) as f:
        config = yaml.safe_load(f)
    unet_cfg, diff_cfg = config['unet'], config['diffusion']
    unet_model = Unet3D(dim=unet_cfg['dim'], rngs=Rngs(unet_cfg['rngs_seed']), dim_mults=tuple(unet_cfg['dim_mults']),
                        channels=unet_cfg['channels'], use_bert_text_cond=unet_cfg['use_bert_text_cond'], mode=args.mode)
    diffusion_model = GaussianDiffusion(denoise_fn=unet_model, image_size=diff_cfg['image_size'], num_frames=diff_cfg['num_frames'],
                                        timesteps=args.timesteps or diff_cfg['timesteps'], loss_type=diff_cfg['loss_type'], channels=diff_cfg['channels'])
    if not args.random_init:
        checkpoint_path = Path(args.checkpoint_path).resolve()
        diffusion_model, _ = load_checkpoint(diffusion_model, args.step, str(checkpoint_path), load_ema_params=args.load_ema_params)
        logging.info(f'Loaded checkpoint from {checkpoint_path} at step {args.step}')
    sampled_videos = diffusion_model.sample(args.seed, batch_size=args.batch_size)
    logging.info(f'Sampled {len(sampled_videos)} videos')
    uint8_videos = videos_to_uint8(sampled_videos.cpu().numpy())
    for i, video_np in enumerate(uint8_videos):
        output_filename = output_path / f'sample_{i}.gif'
        video_array_to_gif(video_np, output_filename)
        logging.info(f'Saved sample {i} to {output_filename}')


if __name__ == '__main__':
    main()
