"""Why does the level-0 persistent conv (conv64p, prologue form, bf16 storage, B = 64) run ~265 us inside the sampling step
(rocprofv3) and 330-450 us replayed standalone?  Varies the data (random / constant / small-range) and the relative placement
of the input and output buffers."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_nnx_amd import ops, _lib as L
import ctypes as C
B, Fr, s, c = 64, 16, 64, 64
dev = torch.device('cuda:0')
n = B * Fr * s * s * c
big = torch.empty(3 * n + (64 << 20), dtype=torch.bfloat16, device=dev)
w = torch.randn(1, 3, 3, c, c, device=dev) / (9 * c) ** 0.5
pw = ops.pack_conv_weights(w, 'bf16'); bias = torch.zeros(c, device=dev)
si = ops.gn_stats_zeros(B, 8, dev); si.view(B, 32, 8, 2)[:, 0, :, 1] = float(Fr * s * s * c // 8)
so = ops.gn_stats_zeros(B, 8, dev)
gamma = torch.ones(c, device=dev); beta = torch.zeros(c, device=dev)
st = torch.cuda.current_stream(dev)

def launch(x, y):
    d = L.ConvDesc()
    d.x_bf16, d.y_bf16 = 1, 1
    d.x0, d.x1, d.c0, d.c1 = L.ptr(x), None, c, 0
    d.packed_w, d.bias, d.y, d.cout = L.ptr(pw), L.ptr(bias), L.ptr(y), c
    d.batch, d.frames, d.h, d.w = B, Fr, s, s
    d.kind, d.kh, d.kw, d.stride = 0, 3, 3, 1
    d.in_stats, d.gamma, d.beta, d.groups = L.ptr(si), L.ptr(gamma), L.ptr(beta), 8
    d.scale_shift = None; d.scale_shift_stride = 0
    d.out_stats, d.out_groups = L.ptr(so), 8
    L.check(L.vdx_conv_forward(1, C.byref(d), L.stream_ptr()))

def run(x, y, reps=12):
    launch(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for i in range(reps):
        launch(x, y)
    e1.record(st); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def view(off_elems):
    return big[off_elems:off_elems + n].view(B, Fr, s, s, c)

for name, fill in (('randn', lambda t: t.normal_()), ('zeros', lambda t: t.zero_()), ('0.5*randn', lambda t: t.normal_().mul_(0.5))):
    x = view(0); fill(x)
    for yoff in (n, n + (1 << 10), n + (1 << 20) // 2 + 4096, n + (7 << 20) + 12288):
        y = view(yoff)
        print(f'data {name:10s} y offset - x size = {(yoff - n) * 2:9d} B: {run(x, y):7.1f} us', flush=True)
