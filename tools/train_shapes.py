"""Per-(kernel, shape) times of the launches of ONE training step that go through LaunchScope (the convolutions, attention / SLA and tail
kernels of the forward and of the data-gradient chain; weight gradients and norm backward have no scope): HIP events around every launch of
5 eager steps (bench.time_kernels_in_step), median per row.  usage: python tools/train_shapes.py [--batch 4]"""
import argparse, os, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4); ap.add_argument('--top', type=int, default=40)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion
    from video_diffusion_nnx_amd.trainer import Trainer
    from video_diffusion_nnx_amd.unet3d import Unet3D
    unet = Unet3D(dim=64, rngs=0, channels=1, mode='bf16')
    gd = GaussianDiffusion(unet, image_size=64, num_frames=16, channels=1, timesteps=1000, loss_type='l2')
    tmp = tempfile.mkdtemp()
    tr = Trainer(gd, tmp, dataset_path='synthetic:64', train_batch_size=a.batch, train_num_steps=10 ** 9, results_folder=tmp)
    x = torch.rand(a.batch, 1, 16, 64, 64).to(dev)
    k = [0]

    def run(n):
        for _ in range(n):
            tr.train_step(x, k[0]); k[0] += 1
    run(2)
    per_key, seq, ov = bench.time_kernels_in_step(run, dev)
    rows = sorted(per_key.items(), key=lambda kv: -kv[1]['ms'])
    tot = sum(d['ms'] for _, d in rows)
    print(f'# {len(seq)} scoped launches per step, {tot:.2f} ms bracketed (event pair {ov:.1f} us subtracted)')
    for (kern, shape), d in rows[:a.top]:
        n = max(d['launches'], 1)
        print(f"{kern:26s} {shape[:78]:78s} n {n:3d} {d['ms'] / n * 1e3:7.1f} us {d['ms']:6.3f} ms  {d['flops'] / n / (d['ms'] / n * 1e-3) / 1e12 if d['ms'] else 0:6.0f} TF/s")


if __name__ == '__main__':
    main()
