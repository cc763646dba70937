"""One 3x3 conv shape of the bench (default: 128 -> 128 at level 1, bf16 tensors, B = CONV_B; CONV_CIN / CONV_COUT / CONV_S / CONV_PRO
override): target of pmc_passes.sh."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_nnx_amd import ops
B, Fr, mode = int(os.environ.get('CONV_B', 32)), 16, 'bf16'
dev = torch.device('cuda:0')
cin, cout, s = int(os.environ.get('CONV_CIN', 128)), int(os.environ.get('CONV_COUT', 128)), int(os.environ.get('CONV_S', 32))
x = torch.randn(B, Fr, s, s, cin, device=dev).to(torch.bfloat16)
w = torch.randn(1, 3, 3, cin, cout, device=dev) / (9 * cin) ** 0.5
pw = ops.pack_conv_weights(w, mode); bias = torch.zeros(cout, device=dev); so = ops.gn_stats_zeros(B, 8, dev)
kw = {}
if int(os.environ.get('CONV_PRO', 0)):
    si = ops.gn_stats_zeros(B, 8, dev); si.view(B, 32, 8, 2)[:, 0, :, 1] = float(Fr * s * s * cin // 8)
    kw = dict(in_stats=si, gamma=torch.ones(cin, device=dev), beta=torch.zeros(cin, device=dev))
for _ in range(6):
    ops.conv_forward(x, pw, cout, mode=mode, bias=bias, k=3, out_stats=so, y_bf16=True, **kw)
torch.cuda.synchronize()
