"""MI355X-native video-diffusion hot path (drop-in for maxsonate/video-diffusion-nnx's Unet3D /
GaussianDiffusion / Trainer surface).  Python host on PyTorch-ROCm tensors -> ctypes -> libvdx.so
(hand-written HIP for gfx950).  There is no CPU fallback."""
__version__ = '0.1.0'
