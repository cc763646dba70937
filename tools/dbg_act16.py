import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet3d_ref as R
from video_diffusion_nnx_amd.unet3d import Unet3D
def rel(a, b): return ((a - b).norm() / (b.norm() + 1e-30)).item()
for kw, shape in [(dict(dim=16, channels=3), (2, 3, 4, 16, 16)), (dict(dim=32, channels=1), (1, 1, 10, 32, 32))]:
    cfg = R.UnetConfig(**kw)
    p = R.random_params(cfg, seed=5, dtype=torch.float64)
    m = Unet3D(rngs=0, mode='bf16', **kw)
    m.load_state_dict({k: v.float() for k, v in p.items()})
    g = torch.Generator().manual_seed(1)
    x = torch.randn(*shape, generator=g); t = torch.randint(0, 1000, (shape[0],), generator=g)
    ref = R.unet_forward(p, cfg, x.double(), t)
    a = m(x, t).cpu().double(); b = m(x, t).cpu().double()
    m.act_bf16 = True; h1 = m(x, t).cpu().double(); h2 = m(x, t).cpu().double()
    m.act_bf16 = False; c = m(x, t).cpu().double(); d = m(x, t).cpu().double()
    print(kw, 'a-ref %.2e b-a %.2e h1-ref %.2e h2-h1 %.2e c-ref %.2e c-a %.2e d-c %.2e' % (rel(a, ref), rel(b, a), rel(h1, ref), rel(h2, h1), rel(c, ref), rel(c, a), rel(d, c)))
