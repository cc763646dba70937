// extern "C" surface of libvdx.so: argument validation + dispatch to the kernel launchers.
#include <stdio.h>
#include <string.h>
#include "vdx_internal.h"

static thread_local char g_err[512] = "";

int vdx_set_error(int code, const char* msg, const char* file, int line) {
    snprintf(g_err, sizeof(g_err), "%s (%s:%d)", msg ? msg : "error", file ? file : "?", line);
    return code;
}
#define VDX_FAIL(code, msg) return vdx_set_error((code), (msg), __FILE__, __LINE__)
#define VDX_HIP(expr)                                                                   \
    do { hipError_t _e = (expr); if (_e != hipSuccess) return vdx_set_error(VDX_ERR_HIP, hipGetErrorString(_e), __FILE__, __LINE__); } while (0)

extern "C" {

const char* vdx_last_error(void) { return g_err; }
int vdx_version(void) { return 1; }

size_t vdx_packed_conv_bytes(int mode, int taps, int cin, int cout) { return vdx::conv_packed_bytes(mode, taps, cin, cout); }

int vdx_pack_conv_weights(int mode, const float* kernel, void* packed, int taps, int cin, int cout, void* stream) {
    if (!kernel || !packed || taps <= 0 || cin <= 0 || cout <= 0) VDX_FAIL(VDX_ERR_INVALID, "pack_conv_weights: bad argument");
    if (mode != VDX_MODE_F32 && mode != VDX_MODE_BF16) VDX_FAIL(VDX_ERR_INVALID, "bad mode");
    VDX_HIP(vdx::launch_pack_weights(mode, kernel, packed, taps, cin, cout, (hipStream_t)stream));
    return VDX_OK;
}

size_t vdx_gn_stats_bytes(int batch, int groups) { return (size_t)batch * 32 /*GN_SLOTS*/ * groups * 2 * sizeof(double); }

int vdx_conv_forward(int mode, const vdx_conv_desc* d, void* stream) {
    if (!d || !d->x0 || !d->packed_w || !d->y) VDX_FAIL(VDX_ERR_INVALID, "conv: null tensor");
    if (mode != VDX_MODE_F32 && mode != VDX_MODE_BF16) VDX_FAIL(VDX_ERR_INVALID, "bad mode");
    if (d->c0 % 4 || d->c1 % 4 || d->cout % 4) VDX_FAIL(VDX_ERR_INVALID, "conv: channel counts must be multiples of 4");
    if (d->c1 && !d->x1) VDX_FAIL(VDX_ERR_INVALID, "conv: x1 missing");
    if (d->batch <= 0 || d->frames <= 0 || d->h <= 0 || d->w <= 0) VDX_FAIL(VDX_ERR_INVALID, "conv: bad geometry");
    vdx::ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x0 = d->x0; a.x1 = d->x1; a.C0 = d->c0; a.C1 = d->c1;
    a.wp = d->packed_w; a.bias = d->bias; a.y = d->y; a.Cout = d->cout;
    a.NF = d->batch * d->frames; a.F = d->frames; a.H = d->h; a.W = d->w;
    a.kind = d->kind;
    if (d->kind == 0) {
        if (d->kh != d->kw || d->kh < 1 || d->kh > 4 || (d->stride != 1 && d->stride != 2)) VDX_FAIL(VDX_ERR_INVALID, "conv: unsupported kernel/stride");
        if (d->stride == 2 && (d->h % 2 || d->w % 2)) VDX_FAIL(VDX_ERR_INVALID, "conv: stride 2 needs even H, W");
        a.kh = d->kh; a.kw = d->kw; a.stride = d->stride;
        // Flax SAME: total = max((ceil(n/s)-1)*s + k - n, 0), low = total/2 (n even for s=2)
        const int total = d->stride == 1 ? d->kh - 1 : d->kh - 2;
        a.pad = total / 2;
        if (d->stride == 1 && (d->kh % 2) == 0) VDX_FAIL(VDX_ERR_INVALID, "conv: even kernel needs stride 2");
    } else if (d->kind == 1) {
        a.kh = a.kw = 4; a.stride = 1; a.pad = 0;
    } else VDX_FAIL(VDX_ERR_INVALID, "conv: bad kind");
    if (d->in_stats) {
        if (d->c1) VDX_FAIL(VDX_ERR_INVALID, "conv: prologue with concat input is not supported");
        if (!d->gamma || !d->beta || d->groups <= 0 || d->groups > 32 || d->c0 % d->groups) VDX_FAIL(VDX_ERR_INVALID, "conv: bad prologue");
        a.pro = 1; a.in_stats = d->in_stats; a.gamma = d->gamma; a.beta = d->beta; a.groups = d->groups;
        a.ss = d->scale_shift; a.ss_stride = d->scale_shift_stride;
    }
    if (d->out_stats) {
        if (d->out_groups <= 0 || d->cout % d->out_groups) VDX_FAIL(VDX_ERR_INVALID, "conv: bad out_groups");
        a.out_stats = d->out_stats; a.out_groups = d->out_groups;
    }
    VDX_HIP(vdx::launch_conv(mode, a, (hipStream_t)stream));
    return VDX_OK;
}

}  // extern "C"
