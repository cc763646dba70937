"""Unet3D -- host-side mirror of the reference class (/root/reference/unet3d.py:21-387).

Same constructor signature and call surface; the device work is one `vdx_unet_forward` call into
libvdx.so (hand-written HIP, gfx950).  Parameters live in ONE flat fp32 device buffer whose layout
(names = nnx state-tree paths, Flax shapes) is defined by the C++ runtime (`vdx_param_info`).
There is no CPU fallback: calling the model without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import logging
import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib as L

BERT_MODEL_DIM = 768     # constant of the external `video_diffusion_pytorch.text` (reference unet3d.py:10)


class Rngs:
    """Stand-in for `flax.nnx.Rngs(seed)` (reference train.py:58): carries the parameter-init seed."""

    def __init__(self, seed: int = 0, **_):
        self.seed = int(seed)


def _seed_of(rngs) -> int:
    if isinstance(rngs, (int, np.integer)):
        return int(rngs)
    for attr in ('seed', 'default', 'params'):
        v = getattr(rngs, attr, None)
        if isinstance(v, (int, np.integer)):
            return int(v)
    return 0


class VdxConfig(C.Structure):
    _fields_ = [('dim', C.c_int), ('n_mults', C.c_int), ('dim_mults', C.c_int * 8), ('channels', C.c_int),
                ('out_dim', C.c_int), ('cond_dim', C.c_int), ('attn_heads', C.c_int), ('attn_dim_head', C.c_int),
                ('init_dim', C.c_int), ('init_kernel_size', C.c_int), ('use_sparse_linear_attn', C.c_int),
                ('resnet_groups', C.c_int), ('image_size', C.c_int), ('num_frames', C.c_int), ('mode', C.c_int)]


_vp = C.c_void_p
vdx_create = L._sig('vdx_create', C.c_int, [C.POINTER(VdxConfig), C.POINTER(_vp)])
vdx_destroy = L._sig('vdx_destroy', None, [_vp])
vdx_param_count = L._sig('vdx_param_count', C.c_int, [_vp])
vdx_param_total = L._sig('vdx_param_total', C.c_long, [_vp])
vdx_param_info = L._sig('vdx_param_info', C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int),
                                                    C.POINTER(C.c_long), C.POINTER(C.c_long)])
vdx_packed_bytes = L._sig('vdx_packed_bytes', C.c_size_t, [_vp])
vdx_pack_params = L._sig('vdx_pack_params', C.c_int, [_vp, _vp, _vp, _vp])
vdx_workspace_bytes = L._sig('vdx_workspace_bytes', C.c_size_t, [_vp, C.c_int])
vdx_slot_count = L._sig('vdx_slot_count', C.c_int, [_vp])
vdx_slot_info = L._sig('vdx_slot_info', C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)])
vdx_num_stages = L._sig('vdx_num_stages', C.c_int, [_vp])
vdx_set_activation_storage = L._sig('vdx_set_activation_storage', C.c_int, [_vp, C.c_int])
vdx_set_attention_fp8 = L._sig('vdx_set_attention_fp8', C.c_int, [_vp, C.c_int])
vdx_get_activation_storage = L._sig('vdx_get_activation_storage', C.c_int, [_vp])
vdx_packed_bwd_bytes = L._sig('vdx_packed_bwd_bytes', C.c_size_t, [_vp])
vdx_pack_params_bwd = L._sig('vdx_pack_params_bwd', C.c_int, [_vp, _vp, _vp, _vp])
vdx_bwd_workspace_bytes = L._sig('vdx_bwd_workspace_bytes', C.c_size_t, [_vp, C.c_int])
vdx_unet_backward = L._sig('vdx_unet_backward', C.c_int, [_vp] * 7 + [_vp, C.c_int, _vp, _vp, _vp, C.c_size_t, _vp, C.c_int, C.c_int, C.c_int, _vp])
vdx_comm_unique_id = L._sig('vdx_comm_unique_id', C.c_int, [C.c_char_p])
vdx_comm_init = L._sig('vdx_comm_init', C.c_int, [_vp, C.c_int, C.c_int, C.c_char_p])
vdx_allreduce_bucket = L._sig('vdx_allreduce_bucket', C.c_int, [_vp, _vp, C.c_size_t, _vp])
vdx_comm_world = L._sig('vdx_comm_world', C.c_int, [_vp])
vdx_comm_destroy = L._sig('vdx_comm_destroy', C.c_int, [_vp])
vdx_unet_forward = L._sig('vdx_unet_forward', C.c_int, [_vp] * 7 + [C.c_int, _vp, _vp, C.c_size_t, C.c_int, _vp])


class _Handle:
    """RAII wrapper of a vdx_handle for one activation geometry (num_frames, image_size)."""

    def __init__(self, cfg: VdxConfig):
        self.ptr = _vp()
        L.check(vdx_create(C.byref(cfg), C.byref(self.ptr)))

    def __del__(self):
        if getattr(self, 'ptr', None):
            vdx_destroy(self.ptr)
            self.ptr = None

    def param_table(self):
        out = []
        name = C.create_string_buffer(256)
        ndim, off = C.c_int(), C.c_long()
        shape = (C.c_long * 6)()
        for i in range(vdx_param_count(self.ptr)):
            L.check(vdx_param_info(self.ptr, i, name, 256, C.byref(ndim), shape, C.byref(off)))
            out.append((name.value.decode(), tuple(shape[:ndim.value]), off.value))
        return out

    def slot_table(self):
        out = {}
        name = C.create_string_buffer(256)
        n, off = C.c_long(), C.c_long()
        for i in range(vdx_slot_count(self.ptr)):
            L.check(vdx_slot_info(self.ptr, i, name, 256, C.byref(n), C.byref(off)))
            out[name.value.decode()] = (n.value, off.value)
        return out


def _trunc_normal(rng: np.random.Generator, shape, std: float) -> np.ndarray:
    """lecun_normal-style truncated normal in [-2, 2] sigma, rescaled to the requested std (Flax default init)."""
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2
    return (out * (std / 0.87962566103423978)).astype(np.float32)


class Unet3D:
    """Space-time factorised 3-D U-Net denoiser (reference unet3d.py:58-75 signature).

    Extra keyword `mode` ('bf16' | 'f16' | 'f32') selects the MFMA arithmetic; `device` the GPU; `attn_fp8` (mode 'bf16', forward /
    sampling only) runs QK^T and PV of the <= 16-token attention blocks on fp8 (e4m3) MFMA operands (BASELINE.json configs[4]).
    """

    def __init__(self, dim: int, rngs=0, dim_mults=(1, 2, 4, 8), cond_dim=None, out_dim=None, channels=3,
                 attn_heads=8, attn_dim_head=32, use_bert_text_cond=False, init_dim=None, init_kernel_size=7,
                 use_sparse_linear_attn=True, block_type='resnet', resnet_groups=8, log_dims=False,
                 *, mode: str = 'bf16', device=None, attn_fp8: bool = False):
        assert init_kernel_size % 2 == 1                                       # unet3d.py:105
        self.dim = dim
        self.dim_mults = tuple(dim_mults)
        self.channels = channels
        self.log_dims = log_dims
        self.has_cond = (cond_dim is not None) or use_bert_text_cond           # unet3d.py:136
        self.cond_dim = BERT_MODEL_DIM if use_bert_text_cond else (cond_dim or 0)
        self.out_dim = out_dim if out_dim is not None else channels
        self.attn_heads, self.attn_dim_head = attn_heads, attn_dim_head
        self.init_dim = init_dim if init_dim is not None else dim
        self.init_kernel_size = init_kernel_size
        self.use_sparse_linear_attn = use_sparse_linear_attn
        self.resnet_groups = resnet_groups
        self.mode = mode
        self.mode_id = L.MODES[mode]
        # bf16 mode only: store every inter-kernel activation as bf16.  True / 1: inference (GaussianDiffusion turns it on for its
        # sampling loops; every fusion on, backward() refuses such a forward); 2: the training forward (every slot the backward reads
        # is materialised as a bf16 tensor -- what Trainer uses); False: fp32 slots (inspectable with slot(), the parity default).
        self.act_bf16 = False
        if attn_fp8 and mode != 'bf16':
            raise ValueError("attn_fp8 needs mode='bf16'")
        self.attn_fp8 = bool(attn_fp8)
        # default: the process's CURRENT device (a rank launched by torch.distributed.run has called set_device(LOCAL_RANK))
        if device is not None:
            self.device = torch.device(device)
            if self.device.type == 'cuda' and self.device.index is None:
                self.device = torch.device('cuda', torch.cuda.current_device())
        else:
            self.device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
        self._handles: Dict[Tuple[int, int], _Handle] = {}
        self._ws: Dict[Tuple[int, int, int], torch.Tensor] = {}
        self._packed: Optional[torch.Tensor] = None
        self._packed_version = -1
        self._packed_t: Optional[torch.Tensor] = None
        self._packed_t_version = -1
        self._bws: Dict[Tuple[int, int, int], torch.Tensor] = {}
        self._last_fwd = None
        self._param_version = 0
        # layout comes from the C++ runtime; geometry does not affect it
        self._layout_handle = self._make_handle(frames=1, size=2 ** (len(self.dim_mults) - 1))
        self.param_table = self._layout_handle.param_table()
        self._index = {n: (shape, off) for n, shape, off in self.param_table}
        total = vdx_param_total(self._layout_handle.ptr)
        host = self._init_params(total, _seed_of(rngs))
        self.flat_params = torch.from_numpy(host).to(self.device)

    # ------------------------------------------------------------------------------------------
    def _config(self, frames: int, size: int) -> VdxConfig:
        c = VdxConfig()
        c.dim = self.dim
        c.n_mults = len(self.dim_mults)
        for i, m in enumerate(self.dim_mults):
            c.dim_mults[i] = int(m)
        c.channels, c.out_dim, c.cond_dim = self.channels, self.out_dim, self.cond_dim
        c.attn_heads, c.attn_dim_head = self.attn_heads, self.attn_dim_head
        c.init_dim, c.init_kernel_size = self.init_dim, self.init_kernel_size
        c.use_sparse_linear_attn = int(bool(self.use_sparse_linear_attn))
        c.resnet_groups = self.resnet_groups
        c.image_size, c.num_frames = size, frames
        c.mode = self.mode_id
        return c

    def _make_handle(self, frames: int, size: int) -> _Handle:
        return _Handle(self._config(frames, size))

    def handle(self, frames: int, size: int) -> _Handle:
        key = (frames, size)
        if key not in self._handles:
            self._handles[key] = self._make_handle(frames, size)
        return self._handles[key]

    def _init_params(self, total: int, seed: int) -> np.ndarray:
        """Flax default initialisers: lecun-normal kernels, zero biases, unit norm scales."""
        rng = np.random.default_rng(seed)
        flat = np.zeros(total, np.float32)
        for name, shape, off in self.param_table:
            n = int(np.prod(shape))
            leaf = name.rsplit('.', 1)[-1]
            if leaf == 'kernel':
                if '.fn.fn.fn.out.' in name:
                    fan_in = shape[0] * shape[1]                 # LinearGeneral((H,D) -> C): flattened fan-in
                elif '.fn.fn.fn.' in name:
                    fan_in = shape[0]                            # LinearGeneral(C -> (H,D))
                else:
                    fan_in = int(np.prod(shape[:-1]))            # conv: receptive field x Cin ; Linear: in
                v = _trunc_normal(rng, shape, math.sqrt(1.0 / fan_in))
            elif leaf == 'scale':
                v = np.ones(shape, np.float32)
            elif leaf == 'embedding':
                v = (rng.standard_normal(shape) / math.sqrt(shape[-1])).astype(np.float32)
            elif name == 'null_cond_emb':
                # reference: float(randint(PRNGKey(0), (1, cond_dim), 1, cond_dim)) (unet3d.py:139-146); the JAX
                # threefry stream is not reproducible here, so the same distribution is drawn from `rng`.
                v = rng.integers(1, max(2, self.cond_dim), size=shape).astype(np.float32)
            else:
                v = np.zeros(shape, np.float32)
            flat[off:off + n] = v.reshape(-1)
        return flat

    # -- parameter access ----------------------------------------------------------------------
    def named_parameters(self):
        for name, shape, off in self.param_table:
            n = int(np.prod(shape))
            yield name, self.flat_params[off:off + n].view(*shape)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return dict(self.named_parameters())

    def get_param(self, name: str) -> torch.Tensor:
        shape, off = self._index[name]
        return self.flat_params[off:off + int(np.prod(shape))].view(*shape)

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        missing = [n for n in self._index if n not in sd]
        extra = [n for n in sd if n not in self._index]
        if strict and (missing or extra):
            raise KeyError(f'load_state_dict: missing {missing[:5]}, unexpected {extra[:5]}')
        with torch.no_grad():
            for name, t in sd.items():
                if name in self._index:
                    shape, _ = self._index[name]
                    self.get_param(name).copy_(torch.as_tensor(t).reshape(shape).to(self.device, torch.float32))
        self.mark_params_updated()

    def set_flat_params(self, flat: torch.Tensor) -> None:
        assert flat.numel() == self.flat_params.numel()
        self.flat_params = flat.to(self.device, torch.float32).contiguous()
        self.mark_params_updated()

    def mark_params_updated(self) -> None:
        self._param_version += 1

    @property
    def null_cond_emb(self):
        return self.get_param('null_cond_emb') if self.has_cond else 0.0

    # -- device state --------------------------------------------------------------------------
    def _require_gpu(self):
        if self.device.type != 'cuda':
            raise RuntimeError('Unet3D needs an MI355X (HIP) device: there is no CPU path in this package')

    def packed(self) -> torch.Tensor:
        """Derived MFMA-layout weights; re-packed lazily after parameter updates."""
        self._require_gpu()
        if self._packed is None:
            self._packed = torch.empty(vdx_packed_bytes(self._layout_handle.ptr), dtype=torch.uint8, device=self.device)
        if self._packed_version != self._param_version:
            L.check(vdx_pack_params(self._layout_handle.ptr, L.ptr(self.flat_params), L.ptr(self._packed), L.stream_ptr()))
            self._packed_version = self._param_version
        return self._packed

    def packed_t(self) -> torch.Tensor:
        """Transposed packing used by the backward's data gradients; re-packed lazily after parameter updates."""
        self._require_gpu()
        if self._packed_t is None:
            self._packed_t = torch.empty(vdx_packed_bwd_bytes(self._layout_handle.ptr), dtype=torch.uint8, device=self.device)
        if self._packed_t_version != self._param_version:
            L.check(vdx_pack_params_bwd(self._layout_handle.ptr, L.ptr(self.flat_params), L.ptr(self._packed_t), L.stream_ptr()))
            self._packed_t_version = self._param_version
        return self._packed_t

    def bwd_workspace(self, batch: int, frames: int, size: int) -> torch.Tensor:
        key = (batch, frames, size)
        if key not in self._bws:
            self._bws[key] = torch.empty(vdx_bwd_workspace_bytes(self.handle(frames, size).ptr, batch), dtype=torch.uint8, device=self.device)
        return self._bws[key]

    @property
    def num_stages(self) -> int:
        return vdx_num_stages(self._layout_handle.ptr)

    def backward(self, d_out: torch.Tensor, grads: torch.Tensor, stage_hi: Optional[int] = None, stage_lo: int = 0) -> None:
        """Reverse pass of the LAST __call__ (same inputs, same workspace): accumulates dL/dparams into the flat `grads`
        (zeroed by the head stage).  d_out = dL/d(output) [B,F,H,W,out_dim].  Stages descend from num_stages-1 to 0."""
        assert self._last_fwd is not None, 'backward() needs a preceding forward'
        x, t32, cond, cm, null_all, B, Fr, S = self._last_fwd
        if stage_hi is None:
            stage_hi = self.num_stages - 1
        h = self.handle(Fr, S)
        ws, bws = self.workspace(B, Fr, S), self.bwd_workspace(B, Fr, S)
        L.check(vdx_unet_backward(h.ptr, L.ptr(self.flat_params), L.ptr(self.packed()), L.ptr(self.packed_t()), L.ptr(x), L.ptr(t32),
                                  L.ptr(cond) if self.has_cond else 0, L.ptr(cm), null_all, L.ptr(d_out), L.ptr(ws), L.ptr(bws), bws.numel(),
                                  L.ptr(grads), stage_hi, stage_lo, B, L.stream_ptr()))

    def apply_activation_storage(self, h) -> None:
        if self.act_bf16 and self.mode != 'bf16':
            raise ValueError("act_bf16 needs mode='bf16'")
        L.check(vdx_set_activation_storage(h.ptr, 2 if self.act_bf16 == 2 and self.act_bf16 is not True else int(bool(self.act_bf16))))
        L.check(vdx_set_attention_fp8(h.ptr, int(self.attn_fp8)))

    def workspace(self, batch: int, frames: int, size: int) -> torch.Tensor:
        key = (batch, frames, size)
        if key not in self._ws:
            n = vdx_workspace_bytes(self.handle(frames, size).ptr, batch)
            self._ws[key] = torch.empty(n, dtype=torch.uint8, device=self.device)
        return self._ws[key]

    def slot(self, name: str, batch: int, frames: int, size: int) -> torch.Tensor:
        """Flat view of a named intermediate of the LAST forward at this geometry (parity/debug tool)."""
        if vdx_get_activation_storage(self.handle(frames, size).ptr):
            raise RuntimeError('slot(): the last forward stored bf16 activations; run with act_bf16 = False to inspect slots')
        n, off = self.handle(frames, size).slot_table()[name]
        ws = self.workspace(batch, frames, size).view(torch.float32)
        return ws[off * batch: off * batch + n * batch]

    # -- forward -------------------------------------------------------------------------------
    def __call__(self, x, time, cond=None, null_cond_prob=0.0, focus_present_mask=None, prob_focus_present=0.0,
                 *, cond_mask=None, out=None):
        """x [B,C,F,H,W], time [B] int -> [B,F,H,W,out_dim] (channel-last, reference unet3d.py:387).

        focus_present_mask / prob_focus_present are accepted and have no effect, exactly as in the reference
        (PreNorm drops them, modules.py:146-148).
        """
        self._require_gpu()
        assert not (self.has_cond and cond is None), 'cond must be passed in if cond_dim specified'   # unet3d.py:271-273
        B, Cc, Fr, H, W = x.shape
        assert Cc == self.channels and H == W, 'expected [B, channels, F, S, S]'
        x = x.to(self.device, torch.float32).contiguous()
        t32 = torch.as_tensor(time).to(self.device, torch.int32).contiguous()
        assert t32.shape == (B,)
        null_all = 0
        cm = None
        if self.has_cond:
            cond = cond.to(self.device, torch.float32).contiguous()
            assert cond.shape == (B, self.cond_dim)
            if cond_mask is not None:
                cm = cond_mask.to(self.device, torch.uint8).contiguous()
            elif null_cond_prob == 1:
                null_all = 1
            elif null_cond_prob == 0:
                null_all = 0
            else:   # prob_mask_like (utils.py:85-101) with the host generator (Q14)
                cm = (torch.rand(B, device=self.device) < null_cond_prob).to(torch.uint8)
        h = self.handle(Fr, H)
        self.apply_activation_storage(h)
        ws = self.workspace(B, Fr, H)
        if out is None:
            out = torch.empty(B, Fr, H, W, self.out_dim, dtype=torch.float32, device=self.device)
        if self.log_dims:
            logging.debug('Unet3D forward: x %s -> %s', tuple(x.shape), tuple(out.shape))
        L.check(vdx_unet_forward(h.ptr, L.ptr(self.flat_params), L.ptr(self.packed()), L.ptr(x), L.ptr(t32),
                                 L.ptr(cond) if self.has_cond else 0, L.ptr(cm), null_all, L.ptr(out), L.ptr(ws), ws.numel(), B,
                                 L.stream_ptr()))
        self._last_fwd = (x, t32, cond if self.has_cond else None, cm, null_all, B, Fr, H)     # inputs kept alive for backward()
        return out

    def forward_with_cond_scale(self, *args, cond_scale=2.0, **kwargs):
        """reference unet3d.py:254-260: eps(c), and with guidance eps(0) + s (eps(c) - eps(0)).

        The two forwards of the guided case run as ONE batch of 2B (first half conditioned, second half on the null embedding,
        selected per sample through cond_mask) -- the same arithmetic per sample, half the launches."""
        if cond_scale == 1 or not self.has_cond:
            return self(*args, null_cond_prob=0.0, **kwargs)
        args = list(args)
        x = args[0] if args else kwargs.pop('x')
        time = args[1] if len(args) > 1 else kwargs.pop('time')
        cond = args[2] if len(args) > 2 else kwargs.pop('cond', None)
        assert cond is not None, 'cond must be passed in if cond_dim specified'
        for k in ('null_cond_prob', 'cond_mask'):
            kwargs.pop(k, None)
        B = x.shape[0]
        x2 = torch.cat((x, x), 0)
        t2 = torch.cat((torch.as_tensor(time), torch.as_tensor(time)), 0)
        c2 = torch.cat((cond, cond), 0)
        mask = torch.zeros(2 * B, dtype=torch.bool, device=x.device)
        mask[B:] = True                                  # second half: null conditioning
        both = self(x2, t2, c2, cond_mask=mask, **kwargs)
        logits, null_logits = both[:B], both[B:]
        return null_logits + (logits - null_logits) * cond_scale
