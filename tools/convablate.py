import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_nnx_amd import ops
B, Fr, mode = 8, 16, 'bf16'
dev = torch.device('cuda:0')
for (cin, cout, s) in ((64, 64, 64), (256, 256, 16), (512, 512, 8)):
    x = torch.randn(B, Fr, s, s, cin, device=dev); w = torch.randn(1, 3, 3, cin, cout, device=dev) / (9*cin)**0.5
    pw = ops.pack_conv_weights(w, mode); bias = torch.zeros(cout, device=dev); so = ops.gn_stats_zeros(B, 8, dev)
    ops.conv_forward(x, pw, cout, mode=mode, bias=bias, k=3, out_stats=so)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv_forward(x, pw, cout, mode=mode, bias=bias, k=3, out_stats=so)
    e1.record(); e1.synchronize()
    print(f'dbg={os.environ.get("VDX_CONV_DBG","0"):>2s} conv {cin}->{cout} @{s}: {e0.elapsed_time(e1)/10*1e3:8.1f} us')
