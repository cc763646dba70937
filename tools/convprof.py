import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from video_diffusion_nnx_amd.unet3d import Unet3D
from video_diffusion_nnx_amd import ops
B, Fr, S, mode = int(os.environ.get('CONVPROF_B', 8)), 16, 64, 'bf16'
ACT16 = os.environ.get('CONVPROF_ACT', 'bf16') == 'bf16'      # bf16 activation storage: bf16 tensors in and out
dev = torch.device('cuda:0')
seen = {}
for layer in bench.conv_layers(64, (1,2,4,8), Fr, S, B):
    if layer in seen: seen[layer][1] += 1; continue
    cin, cout, s, taps, kind = layer
    k = {9: 3, 1: 1, 16: 4}[taps]
    x = torch.randn(B, Fr, s, s, cin, device=dev)
    if ACT16: x = x.to(torch.bfloat16)
    w = torch.randn(1, k, k, cin, cout, device=dev) / (taps*cin)**0.5
    pw = ops.pack_conv_weights(w, mode); bias = torch.zeros(cout, device=dev)
    so = ops.gn_stats_zeros(B, 8, dev); si = ops.gn_stats_zeros(B, 8, dev); si.view(B,32,8,2)[:,0,:,1] = float(Fr*s*s*cin//8)
    g = torch.ones(cin, device=dev); be = torch.zeros(cin, device=dev)
    kw = dict(mode=mode, bias=bias, y_bf16=ACT16)
    if kind == 'c3': kw.update(k=3, out_stats=so)
    elif kind == 'c3p': kw.update(k=3, in_stats=si, gamma=g, beta=be, out_stats=so)
    elif kind == 'c1': kw.update(k=1)
    elif kind == 'down': kw.update(k=4, stride=2)
    else: kw.update(k=4, kind=1)
    ops.conv_forward(x, pw, cout, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv_forward(x, pw, cout, **kw)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1)/10
    seen[layer] = [ms, 1, bench.conv_flops(layer, Fr, B)]
tot = 0
for l, (ms, n, fl) in seen.items():
    tot += ms*n
    print(f'{str(l):36s} x{n}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s   total {ms*n*1e3:8.1f} us')
print('sum ms', tot)
