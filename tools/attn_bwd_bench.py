"""Standalone timing of the fused temporal-attention backward (attn_bwd16x_kernel) at the widest level of the training shape
(batch 4, 16 frames, 64 x 64, C = 64): us per launch and the HBM rate of its algorithmic bytes (x, dy in; o, dqkv, dx out).
    python tools/attn_bwd_bench.py [--batch 4] [--iters 20]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4); ap.add_argument('--frames', type=int, default=16); ap.add_argument('--size', type=int, default=64)
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    from video_diffusion_nnx_amd import ops
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(a.batch, a.frames, a.size, a.size, 64, generator=g).to(dev)
    dy = torch.randn(a.batch, a.frames, a.size, a.size, 64, generator=g).to(dev)
    wqkv = (torch.randn(64, 768, generator=g) * 0.15).to(dev); bqkv = torch.zeros(768, device=dev); wo = (torch.randn(256, 64, generator=g) * 0.1).to(dev)
    for _ in range(3):
        ops.temporal_attention_backward_fused(x, dy, wqkv, bqkv, wo)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        ops.temporal_attention_backward_fused(x, dy, wqkv, bqkv, wo)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    rows = x.numel() // 64
    bytes_ = rows * (64 * 4 * 3 + 256 * 2 + 768 * 2)
    print(f'fused temporal attention backward: {rows} rows, {us:.1f} us/launch (incl. weight packing + output allocation), {bytes_ / us / 1e6:.2f} TB/s of {bytes_ / 1e6:.0f} MB')


if __name__ == '__main__':
    main()
