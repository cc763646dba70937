"""Training throughput of the north-star shape (BASELINE.json configs[2]: dim 64, 16f x 64x64, batch 4 per GPU, l2).
    python tools/train_bench.py [--batch 4] [--steps 5] [--mode bf16]
One process per GPU (launch with torch.distributed.run for N > 1); prints samples/s and ms/step."""
import argparse, os, sys, time, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4); ap.add_argument('--steps', type=int, default=5); ap.add_argument('--mode', default='bf16')
    ap.add_argument('--dim', type=int, default=64); ap.add_argument('--frames', type=int, default=16); ap.add_argument('--size', type=int, default=64)
    a = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1')); local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion
    from video_diffusion_nnx_amd.trainer import Trainer
    from video_diffusion_nnx_amd.unet3d import Unet3D
    unet = Unet3D(dim=a.dim, rngs=0, channels=1, mode=a.mode)
    gd = GaussianDiffusion(unet, image_size=a.size, num_frames=a.frames, channels=1, timesteps=1000, loss_type='l2')
    tmp = tempfile.mkdtemp()
    tr = Trainer(gd, tmp, dataset_path='synthetic:64', train_batch_size=a.batch * world, train_num_steps=10 ** 9, results_folder=tmp)
    x = torch.rand(a.batch, 1, a.frames, a.size, a.size).to(torch.device('cuda', local))      # resident, as bench.py's train leg
    for i in range(2):
        tr.train_step(x, i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = tr.train_step(x, 2 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    if int(os.environ.get('RANK', '0')) == 0:
        print(f'train: mode={a.mode} world={world} batch/gpu={a.batch} {dt*1e3:.1f} ms/step {a.batch*world/dt:.2f} samples/s loss={loss.item():.4f}', flush=True)


if __name__ == '__main__':
    main()
