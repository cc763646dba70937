import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_nnx_amd.unet3d import Unet3D
def rel(a, b): return ((a - b).norm() / (b.norm() + 1e-30)).item()
for mode in ('f32', 'bf16'):
    for kw, shape in [(dict(dim=16, channels=3), (2, 3, 4, 16, 16)), (dict(dim=64, channels=1), (1, 1, 16, 64, 64))]:
        m = Unet3D(rngs=0, mode=mode, **kw)
        g = torch.Generator().manual_seed(1)
        x = torch.randn(*shape, generator=g); t = torch.randint(0, 1000, (shape[0],), generator=g)
        B, Fr, S = shape[0], shape[2], shape[3]
        names = list(m.handle(Fr, S).slot_table().keys())
        m(x, t); torch.cuda.synchronize()
        first = {n: m.slot(n, B, Fr, S).clone() for n in names}
        m(x, t); torch.cuda.synchronize()
        out = []
        for n in names:
            if n.endswith('#y1') or n.endswith('#y2'):
                if mode == 'bf16':
                    k = first[n].numel() // 2
                    a = first[n].view(torch.bfloat16)[:k].float(); b = m.slot(n, B, Fr, S).view(torch.bfloat16)[:k].float()
                else:
                    a, b = first[n], m.slot(n, B, Fr, S)
            else:
                a, b = first[n], m.slot(n, B, Fr, S)
            out.append((n, rel(a.double(), b.double())))
        print(mode, kw, [(n, '%.1e' % r) for n, r in out if r > 0][:6])
