"""Training CLI for the MI355X path.  Accepts the reference CLI's flags (reference train.py:23-42: --config,
--resume_step, --rng_seed) and its YAML schema (unet / diffusion / trainer sections).  Keys a YAML leaves out fall back
to the Trainer defaults (the reference indexes them and raises KeyError for its own v1_0..v2_2 files).
Multi-GPU: `python -m torch.distributed.run --nproc-per-node N train.py --config ...` = one process per GPU over RCCL;
`train_batch_size` stays the GLOBAL batch and is split over ranks like the reference splits it over devices.
Extensions: --mode {bf16,f32}; --train_num_steps N; --dataset_path P (e.g. synthetic:64)."""
import argparse
import logging
import os
import pathlib

import yaml

HERE = pathlib.Path(__file__).resolve().parent
FLAGS = (
    ('--config', dict(type=str, default=str(HERE / 'configs' / 'config.yaml'), help='YAML with unet / diffusion / trainer sections')),
    ('--resume_step', dict(type=int, default=0, help='restore params + EMA of this step first (optimizer state restarts)')),
    ('--rng_seed', dict(type=int, default=None, help='master seed; default: config rng_seed, else 0')),
    ('--mode', dict(choices=('bf16', 'f16', 'f32'), default='bf16', help='MFMA operand precision')),
    ('--train_num_steps', dict(type=int, default=None, help='override trainer.train_num_steps')),
    ('--dataset_path', dict(type=str, default=None, help='override trainer.dataset_path')),
)


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format='%(levelname)s:%(name)s:%(message)s', force=True)
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    for flag, kw in FLAGS:
        ap.add_argument(flag, **kw)
    a = ap.parse_args(argv)

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        import torch.distributed as dist
        local = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))

    from sample import build_models
    from video_diffusion_nnx_amd.trainer import Trainer

    with open(a.config) as fh:
        cfg = yaml.safe_load(fh)
    seed = a.rng_seed if a.rng_seed is not None else cfg.get('rng_seed', 0)
    logging.info('config %s, master seed %s', a.config, seed)
    _, gd = build_models(cfg, a.mode)
    tc = dict(cfg['trainer'])
    if a.train_num_steps is not None:
        tc['train_num_steps'] = a.train_num_steps
    if a.dataset_path is not None:
        tc['dataset_path'] = a.dataset_path
    tc.pop('resume_training_step', None)             # the command-line flag wins, as in the reference
    trainer = Trainer(diffusion_model=gd, folder=tc.pop('folder'), resume_training_step=a.resume_step, rng_seed=seed, **tc)
    trainer.train()
    logging.info('done')
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
