import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import philox_ref, unet3d_ref as R
from oracle.diffusion_ref import DiffusionRef
from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion
from video_diffusion_nnx_amd.unet3d import Unet3D
kw = dict(dim=16, channels=1); cfg = R.UnetConfig(**kw); p = R.random_params(cfg, seed=1, dtype=torch.float64)
for T in (4, 50):
    shape = (1, 1, 4, 16, 16); n = int(np.prod(shape))
    ref = DiffusionRef(lambda a, b: R.unet_forward(p, cfg, a, b), image_size=16, num_frames=4, channels=1, timesteps=T, dtype=torch.float64)
    exp = ref.p_sample_loop(torch.from_numpy(philox_ref.randn(n, 7, 0)).double().reshape(shape),
                            [torch.from_numpy(philox_ref.randn(n, 7, 1 + k)).double().reshape(shape) for k in range(T)])
    for mode, act in (('f32', False), ('bf16', False), ('bf16', True)):
        unet = Unet3D(rngs=0, mode=mode, **kw); unet.load_state_dict({k: v.float() for k, v in p.items()})
        gd = GaussianDiffusion(unet, image_size=16, num_frames=4, channels=1, timesteps=T, sample_act_bf16=act)
        out = gd.p_sample_loop(shape, 7).cpu().double()
        d = (out - exp).abs()
        print(f'T={T} mode={mode} act_bf16={act}: max-abs {d.max().item():.3e} mean-abs {d.mean().item():.3e} rel-L2 {((out-exp).norm()/exp.norm()).item():.3e}')
