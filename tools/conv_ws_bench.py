"""Times the weight-streaming 3x3 conv (conv_ws.hip) at the wide-level shapes of the north-star config, standalone.
    python tools/conv_ws_bench.py [--batch 64] [--reps 10]
Prints per shape: us per launch, TFLOP/s (2 * B * F * H * W * Cin * Cout * 9), for the plain and the fused-prologue form."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_nnx_amd import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64); ap.add_argument('--reps', type=int, default=10); ap.add_argument('--frames', type=int, default=16)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    B, Fr = a.batch, a.frames
    shapes = [(32, 64, 0, 128), (32, 128, 0, 128), (16, 128, 0, 256), (16, 256, 0, 256), (16, 256, 256, 128), (8, 256, 0, 512), (8, 512, 0, 512), (8, 512, 512, 256), (8, 256, 0, 256)]
    st = torch.cuda.current_stream()
    tot = {False: [0.0, 0.0], True: [0.0, 0.0]}
    for S, c0, c1, cout in shapes:
        cin = c0 + c1
        x0 = torch.randn(B, Fr, S, S, c0, device=dev).to(torch.bfloat16)
        x1 = torch.randn(B, Fr, S, S, c1, device=dev).to(torch.bfloat16) if c1 else None
        w = torch.randn(1, 3, 3, cin, cout, device=dev) / (9 * cin) ** 0.5
        pw = ops.pack_conv_weights(w, 'bf16')
        bias = torch.zeros(cout, device=dev)
        sin = ops.gn_stats_zeros(B, 8, dev)
        sin.view(B, 32, 8, 2)[:, 0, :, 1] = float(Fr * S * S * cin // 8)
        sout = ops.gn_stats_zeros(B, 8, dev)
        gamma, beta = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
        ss = torch.zeros(B, 2 * cin, device=dev)
        flops = 2.0 * B * Fr * S * S * cin * cout * 9
        for pro in (False, True):
            if pro and c1:
                continue
            kw = dict(mode='bf16', bias=bias, x1=x1, out_stats=sout, y_bf16=True)
            if pro:
                kw.update(in_stats=sin, gamma=gamma, beta=beta, scale_shift=ss)
            ops.conv_forward(x0, pw, cout, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(a.reps):
                ops.conv_forward(x0, pw, cout, **kw)
            e1.record(st); e1.synchronize()
            us = e0.elapsed_time(e1) / a.reps * 1e3
            tot[pro][0] += us; tot[pro][1] += flops
            print(f'S={S:2d} {cin:4d}->{cout:3d} pro={int(pro)}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s', flush=True)
    for pro in (False, True):
        print(f'total pro={int(pro)}: {tot[pro][0]:.0f} us, {tot[pro][1] / tot[pro][0] / 1e6:.1f} TFLOP/s')


if __name__ == '__main__':
    main()
