"""Sampling CLI for the MI355X path.  Accepts the reference CLI's flags (reference sample.py:19-62: --config,
--output-path, --checkpoint-path, --step, --seed, --batch-size, --load-ema-params) and its YAML schema, then runs
GaussianDiffusion.sample on the GPU and writes one GIF per video (batch-global min-max to uint8, 120 ms/frame).
Extensions: --mode {bf16,f16,f32}; --random-init (no checkpoint); --timesteps N (shorter chain for smoke runs); --ddim-steps S;
--attn-fp8 (bf16 mode: QK^T / PV of the <= 16-token attention blocks on fp8 MFMA operands)."""
import argparse
import logging
import os
import pathlib

import yaml

HERE = pathlib.Path(__file__).resolve().parent
FLAGS = (   # (flag, kwargs)
    ('--config', dict(type=str, default=str(HERE / 'configs' / 'config.yaml'), help='YAML with unet / diffusion / trainer sections')),
    ('--output-path', dict(type=str, default=str(HERE / 'outputs'), help='where the sample_<i>.gif files go')),
    ('--checkpoint-path', dict(type=str, default=None, help='checkpoint directory (required unless --random-init)')),
    ('--step', dict(type=int, default=0, help='which saved step to load')),
    ('--seed', dict(type=int, default=0, help='Philox seed of the sampling chain')),
    ('--batch-size', dict(type=int, default=2, help='videos to draw')),
    ('--load-ema-params', dict(action='store_true', help='sample from the EMA weights')),
    ('--mode', dict(choices=('bf16', 'f16', 'f32'), default='bf16', help='MFMA operand precision')),
    ('--random-init', dict(action='store_true', help='skip the checkpoint, use freshly initialised weights')),
    ('--timesteps', dict(type=int, default=None, help='override diffusion.timesteps')),
    ('--ddim-steps', dict(type=int, default=None, help='sample with an S-step DDIM chain (eta = 0) instead of the T-step ancestral one')),
    ('--attn-fp8', dict(action='store_true', help='bf16 mode: fp8 (e4m3) QK^T / PV in the attention blocks over <= 16 tokens')),
)


def build_models(cfg, mode, timesteps=None, attn_fp8=False):
    from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion
    from video_diffusion_nnx_amd.unet3d import Rngs, Unet3D
    u, d = cfg['unet'], cfg['diffusion']
    unet = Unet3D(dim=u['dim'], rngs=Rngs(u['rngs_seed']), dim_mults=tuple(u['dim_mults']), channels=u['channels'],
                  use_bert_text_cond=u['use_bert_text_cond'], mode=mode, attn_fp8=attn_fp8)
    gd = GaussianDiffusion(denoise_fn=unet, image_size=d['image_size'], num_frames=d['num_frames'], channels=d['channels'],
                           timesteps=timesteps or d['timesteps'], loss_type=d['loss_type'])
    return unet, gd


def main(argv=None):
    logging.basicConfig(level=logging.INFO, force=True)
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    for flag, kw in FLAGS:
        ap.add_argument(flag, **kw)
    a = ap.parse_args(argv)
    if a.checkpoint_path is None and not a.random_init:
        ap.error('--checkpoint-path is required (or pass --random-init)')

    # one process per GPU under `python -m torch.distributed.run --nproc-per-node N sample.py ...` (reference gaussian_diffusion.py:278-298
    # shards the batch over the local devices): every rank draws batch_size / N of the videos and writes its own GIFs
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world > 1:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            local = int(os.environ.get('LOCAL_RANK', '0'))
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            torch.cuda.set_device(local)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    try:
        _run(a, ap, rank, world)
    finally:
        if world > 1:
            import torch.distributed as dist
            if dist.is_initialized() and os.environ.get('VDX_KEEP_PROCESS_GROUP') != '1':
                dist.destroy_process_group()


def _run(a, ap, rank, world):
    from video_diffusion_nnx_amd.checkpoint import load_checkpoint
    from video_diffusion_nnx_amd.media import video_array_to_gif, videos_to_uint8

    out_dir = pathlib.Path(a.output_path)
    out_dir.mkdir(parents=True, exist_ok=True)
    with open(a.config) as fh:
        cfg = yaml.safe_load(fh)
    logging.info('config %s', a.config)
    _, gd = build_models(cfg, a.mode, a.timesteps, attn_fp8=a.attn_fp8)
    if not a.random_init:
        ckpt = pathlib.Path(a.checkpoint_path).resolve()
        gd, _ = load_checkpoint(gd, a.step, str(ckpt), load_ema_params=a.load_ema_params)
        logging.info('restored step %d from %s', a.step, ckpt)
    videos = gd.sample(a.seed, batch_size=a.batch_size, ddim_steps=a.ddim_steps)          # this rank's shard of the global batch
    logging.info('rank %d drew %d videos', rank, len(videos))
    lo_hi = None
    if world > 1:                                      # the uint8 scaling is batch-GLOBAL (reference sample.py:107-110): two scalars cross ranks
        import torch
        import torch.distributed as dist
        mm = torch.stack([videos.min(), -videos.max()])
        dist.all_reduce(mm, op=dist.ReduceOp.MIN)
        lo_hi = (float(mm[0].item()), float(-mm[1].item()))
    first = rank * len(videos)
    for i, frames in enumerate(videos_to_uint8(videos.cpu().numpy(), lo_hi=lo_hi)):
        target = out_dir / f'sample_{first + i}.gif'
        video_array_to_gif(frames, target)
        logging.info('wrote %s', target)


if __name__ == '__main__':
    main()
