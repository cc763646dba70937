"""ctypes binding of libvdx.so (the C ABI declared in include/vdx.h).

This is the only place the Python host touches native code.  There is NO fallback: if the shared
library is missing or a symbol cannot be resolved, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VDX_LIB: a variant build of the same sources (tools/ab_variants.sh: A/B timing of kernel changes); the product path is the in-tree library
LIB_PATH = os.environ.get('VDX_LIB') or os.path.join(_HERE, 'libvdx.so')

MODE_F32 = 0
MODE_BF16 = 1
MODE_F16 = 2
MODES = {'f32': MODE_F32, 'fp32': MODE_F32, 'float32': MODE_F32, 'bf16': MODE_BF16, 'bfloat16': MODE_BF16,
         'f16': MODE_F16, 'fp16': MODE_F16, 'float16': MODE_F16}
GN_SLOTS = 32


class VdxError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f'{LIB_PATH} not found: the HIP extension is required (no CPU fallback exists). '
        'Build it with `python -c "import __graft_entry__ as g; g.build()"` or `make -C video_diffusion_nnx_amd/csrc`.')

lib = C.CDLL(LIB_PATH)

c_void_p, c_int, c_size_t, c_float, c_double = C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_double


class ConvDesc(C.Structure):
    _fields_ = [
        ('x0', c_void_p), ('x1', c_void_p), ('c0', c_int), ('c1', c_int),
        ('packed_w', c_void_p), ('bias', c_void_p), ('y', c_void_p), ('cout', c_int),
        ('batch', c_int), ('frames', c_int), ('h', c_int), ('w', c_int),
        ('kind', c_int), ('kh', c_int), ('kw', c_int), ('stride', c_int),
        ('in_stats', c_void_p), ('gamma', c_void_p), ('beta', c_void_p), ('groups', c_int),
        ('scale_shift', c_void_p), ('scale_shift_stride', c_int),
        ('out_stats', c_void_p), ('out_groups', c_int),
        ('x_bf16', c_int), ('y_bf16', c_int),
        ('res', c_void_p), ('res_bf16', c_int),
    ]


def _sig(name, restype, argtypes):
    fn = getattr(lib, name)          # AttributeError if the symbol is missing: fail loudly
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


vdx_last_error = _sig('vdx_last_error', C.c_char_p, [])
vdx_version = _sig('vdx_version', c_int, [])
vdx_packed_conv_bytes = _sig('vdx_packed_conv_bytes', c_size_t, [c_int, c_int, c_int, c_int])
vdx_pack_conv_weights = _sig('vdx_pack_conv_weights', c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p])
vdx_gn_stats_bytes = _sig('vdx_gn_stats_bytes', c_size_t, [c_int, c_int])
vdx_conv_forward = _sig('vdx_conv_forward', c_int, [c_int, C.POINTER(ConvDesc), c_void_p])


def check(status: int) -> None:
    if status != 0:
        raise VdxError(f'libvdx status {status}: {vdx_last_error().decode()}')


def ptr(t) -> int:
    """Device pointer of a torch tensor (must be contiguous) or 0 for None."""
    if t is None:
        return 0
    assert t.is_contiguous(), 'libvdx needs contiguous tensors'
    return t.data_ptr()


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream

c_long = C.c_long
vdx_resblock_tail = _sig('vdx_resblock_tail', c_int, [c_void_p] * 6 + [c_int, c_void_p, c_void_p, c_int, c_int, c_long, c_void_p])
vdx_gn_silu_apply_bf16 = _sig('vdx_gn_silu_apply_bf16', c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_int, c_long, c_void_p])
vdx_resblock_tail_rc_bf16 = _sig('vdx_resblock_tail_rc_bf16', c_int, [c_void_p] * 3 + [c_int, c_int] + [c_void_p] * 6 + [c_int, c_void_p, c_void_p, c_int, c_int, c_long, c_void_p])
vdx_init_conv = _sig('vdx_init_conv', c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p])
vdx_final_conv = _sig('vdx_final_conv', c_int, [c_void_p] * 4 + [c_long, c_int, c_int, c_void_p])
vdx_time_mlp = _sig('vdx_time_mlp', c_int, [c_void_p] * 5 + [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p])
vdx_attention_forward = _sig('vdx_attention_forward', c_int, [c_int] + [c_void_p] * 6 + [c_int] * 7 + [c_void_p])
vdx_attention_forward_ex = _sig('vdx_attention_forward_ex', c_int, [c_int] + [c_void_p] * 6 + [c_int] * 8 + [c_void_p])
vdx_attention_forward_bf16 = _sig('vdx_attention_forward_bf16', c_int, [c_void_p] * 6 + [c_int] * 8 + [c_void_p])
vdx_sla_forward_bf16 = _sig('vdx_sla_forward_bf16', c_int, [c_void_p] * 7 + [c_int] * 6 + [c_void_p])
vdx_sla_workspace_bytes = _sig('vdx_sla_workspace_bytes', c_size_t, [c_int] * 4)
vdx_sla_forward = _sig('vdx_sla_forward', c_int, [c_int] + [c_void_p] * 7 + [c_int] * 6 + [c_void_p])
