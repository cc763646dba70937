import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_nnx_amd import ops
B, Fr, mode = 8, 16, 'bf16'
dev = torch.device('cuda:0')
for (C, s) in ((64, 64), (128, 32), (512, 8)):
    x = torch.randn(B, Fr, s, s, C, device=dev)
    g = torch.Generator(device='cpu').manual_seed(0)
    mk = lambda *sh: torch.randn(*sh, generator=g).to(dev)
    packed = ops.pack_mha((mk(C, 8, 32) / C ** 0.5, mk(8, 32)), (mk(C, 8, 32) / C ** 0.5, mk(8, 32)), (mk(C, 8, 32) / C ** 0.5, mk(8, 32)), (mk(8, 32, C) / 16, mk(C)), mode)
    for _ in range(6):
        y = ops.attention_forward(x, packed, 8, True, mode)
    torch.cuda.synchronize()
    print('ran', C, s)
