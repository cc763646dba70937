// extern "C" surface of libvdx.so: argument validation + dispatch to the kernel launchers.
#include <stdio.h>
#include <string.h>
#include <math.h>
#include "vdx_internal.h"
#include "model.h"
#include "comm.h"

struct vdx_handle {
    vdx::Model model;
    // every argument a captured sampling step bakes in (full pointers: two workspaces / streams never alias one key)
    struct GraphKey { const void* p[12]; unsigned long long seed; int i[4]; size_t ws; float f; };
    // one cached graph per loop kind: 0 = DDPM p_sample_loop, 1 = DDIM
    // `last` = the stream the exec was last launched on: replays may still be running when the graph has to go
    struct GraphSlot {
        hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; GraphKey key; hipStream_t last = nullptr;
        void drop() {
            if (exec && last) (void)hipStreamSynchronize(last);       // earlier hipGraphLaunch calls of this exec have finished
            if (exec) { (void)hipGraphExecDestroy(exec); exec = nullptr; }
            if (graph) { (void)hipGraphDestroy(graph); graph = nullptr; }
            last = nullptr;
        }
    } gs[2];
    void drop_graphs() { for (GraphSlot& g : gs) g.drop(); }
    vdx::BwdState bwd;
    vdx::Comm comm;
};

namespace vdx {
vdx_launch_hook g_launch_hook = nullptr;
void* g_launch_hook_user = nullptr;
}

static thread_local char g_err[512] = "";

int vdx_set_error(int code, const char* msg, const char* file, int line) {
    snprintf(g_err, sizeof(g_err), "%s (%s:%d)", msg ? msg : "error", file ? file : "?", line);
    return code;
}
#define VDX_FAIL(code, msg) return vdx_set_error((code), (msg), __FILE__, __LINE__)
#define VDX_HIP(expr)                                                                   \
    do { hipError_t _e = (expr); if (_e != hipSuccess) return vdx_set_error(VDX_ERR_HIP, hipGetErrorString(_e), __FILE__, __LINE__); } while (0)

// runs `step` nsteps times on `st`: eagerly, or (use_graph) captured once into the handle's graph slot `which` and replayed
template <class Step>
static int run_steps(vdx_handle* h, int which, const vdx_handle::GraphKey& key, int nsteps, int use_graph, hipStream_t st, Step step) {
    if (!use_graph) {
        for (int i = 0; i < nsteps; ++i) { int rc = step(); if (rc != VDX_OK) return rc; }
        return VDX_OK;
    }
    vdx_handle::GraphSlot& g = h->gs[which];
    int done = 0;
    if (!g.exec || memcmp(&key, &g.key, sizeof(key)) != 0) {
        if (nsteps == 0) return VDX_OK;
        int rc = step();                                    // eager first step (also sets kernel attributes before capture)
        if (rc != VDX_OK) return rc;
        done = 1;
        g.drop();
        VDX_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        rc = step();
        hipError_t ce = hipStreamEndCapture(st, &g.graph);
        if (rc != VDX_OK || ce != hipSuccess) {             // a failed capture leaves no half-built graph behind
            if (g.graph) { (void)hipGraphDestroy(g.graph); g.graph = nullptr; }
            if (rc != VDX_OK) return rc;
            return vdx_set_error(VDX_ERR_HIP, hipGetErrorString(ce), __FILE__, __LINE__);
        }
        hipError_t ie = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
        if (ie != hipSuccess) { g.exec = nullptr; g.drop(); return vdx_set_error(VDX_ERR_HIP, hipGetErrorString(ie), __FILE__, __LINE__); }
        g.key = key;
    }
    g.last = st;
    for (int i = done; i < nsteps; ++i) VDX_HIP(hipGraphLaunch(g.exec, st));
    return VDX_OK;
}

extern "C" {

const char* vdx_last_error(void) { return g_err; }
int vdx_version(void) { return 1; }

size_t vdx_packed_conv_bytes(int mode, int taps, int cin, int cout) { return vdx::conv_packed_bytes(mode, taps, cin, cout); }

int vdx_pack_conv_weights(int mode, const float* kernel, void* packed, int taps, int cin, int cout, void* stream) {
    if (!kernel || !packed || taps <= 0 || cin <= 0 || cout <= 0) VDX_FAIL(VDX_ERR_INVALID, "pack_conv_weights: bad argument");
    if (mode != VDX_MODE_F32 && mode != VDX_MODE_BF16 && mode != VDX_MODE_F16) VDX_FAIL(VDX_ERR_INVALID, "bad mode");
    VDX_HIP(vdx::launch_pack_weights(mode, kernel, packed, taps, cin, cout, (hipStream_t)stream));
    return VDX_OK;
}

void vdx_set_launch_hook(vdx_launch_hook hook, void* user) { vdx::g_launch_hook = hook; vdx::g_launch_hook_user = user; }

size_t vdx_gn_stats_bytes(int batch, int groups) { return (size_t)batch * 32 /*GN_SLOTS*/ * groups * 2 * sizeof(double); }

int vdx_conv_forward(int mode, const vdx_conv_desc* d, void* stream) {
    if (!d || !d->x0 || !d->packed_w || !d->y) VDX_FAIL(VDX_ERR_INVALID, "conv: null tensor");
    if (mode != VDX_MODE_F32 && mode != VDX_MODE_BF16 && mode != VDX_MODE_F16) VDX_FAIL(VDX_ERR_INVALID, "bad mode");
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
    if ((d->x_bf16 || d->y_bf16) && mode != VDX_MODE_BF16) VDX_FAIL(VDX_ERR_INVALID, "conv: bf16 tensors need VDX_MODE_BF16");
    if (d->x_bf16 && (d->c0 % 8 || d->c1 % 8)) VDX_FAIL(VDX_ERR_INVALID, "conv: bf16 inputs need channel counts that are multiples of 8");
    a.x0_bf16 = a.x1_bf16 = d->x_bf16 ? 1 : 0; a.y_bf16 = d->y_bf16 ? 1 : 0;
    if (d->res) {
        if (d->out_stats || d->kind != 0 || d->stride != 1) VDX_FAIL(VDX_ERR_INVALID, "conv: res needs a stride-1 conv without statistics");
        if (d->res_bf16 && mode != VDX_MODE_BF16) VDX_FAIL(VDX_ERR_INVALID, "conv: bf16 tensors need VDX_MODE_BF16");
        a.res = (const float*)d->res; a.res_bf16 = d->res_bf16 ? 1 : 0;
    }
    VDX_HIP(vdx::launch_conv(mode, a, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_resblock_tail(const float* y2, const float* r, float* out, const double* stats, const float* gn_gamma,
                      const float* gn_beta, int groups, const float* ln_gamma, const float* ln_beta, int c, int batch,
                      long pix_per_sample, void* stream) {
    if (!y2 || !r || !out || !stats || !gn_gamma || !gn_beta || !ln_gamma || !ln_beta) VDX_FAIL(VDX_ERR_INVALID, "tail: null tensor");
    if (c % 4 || c > 1024 || groups <= 0 || groups > 32 || c % groups) VDX_FAIL(VDX_ERR_INVALID, "tail: bad channels/groups");
    vdx::TailArgs a;
    memset(&a, 0, sizeof(a));
    a.y2 = y2; a.r = r; a.out = out; a.stats = stats; a.gn_gamma = gn_gamma; a.gn_beta = gn_beta; a.groups = groups;
    a.ln_gamma = ln_gamma; a.ln_beta = ln_beta; a.C = c; a.batch = batch; a.pix_per_sample = pix_per_sample;
    VDX_HIP(vdx::launch_resblock_tail(a, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_resblock_tail_rc_bf16(const void* y2, const void* x0, const void* x1, int c0, int c1, const void* rc_w_packed,
                              const float* rc_bias, void* out, const double* stats, const float* gn_gamma, const float* gn_beta,
                              int groups, const float* ln_gamma, const float* ln_beta, int c, int batch, long pix_per_sample,
                              void* stream) {
    if (!y2 || !x0 || !rc_w_packed || !rc_bias || !out || !stats || !gn_gamma || !gn_beta || !ln_gamma || !ln_beta || (c1 && !x1))
        VDX_FAIL(VDX_ERR_INVALID, "tail_rc: null tensor");
    if (c1 < 0 || batch < 1 || groups <= 0 || groups > 32 || c % groups || !vdx::tail_rc16_supported(c0 + c1, c0, c, pix_per_sample))
        VDX_FAIL(VDX_ERR_INVALID, "tail_rc: shape not served");
    vdx::TailArgs a;
    memset(&a, 0, sizeof(a));
    a.y2 = (const float*)y2; a.y2_bf16 = 1; a.out = (float*)out; a.out_bf16 = 1; a.r_bf16 = 1;
    a.stats = stats; a.gn_gamma = gn_gamma; a.gn_beta = gn_beta; a.groups = groups;
    a.ln_gamma = ln_gamma; a.ln_beta = ln_beta; a.C = c; a.batch = batch; a.pix_per_sample = pix_per_sample;
    a.x0 = (const float*)x0; a.x1 = (const float*)x1; a.C0 = c0; a.C1 = c1; a.rc_w = rc_w_packed; a.rc_b = rc_bias;
    VDX_HIP(vdx::launch_resblock_tail(a, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_gn_silu_apply_bf16(void* y, const double* stats, const float* gn_gamma, const float* gn_beta, const float* scale_shift,
                           int ss_stride, int groups, int c, int batch, long pix_per_sample, void* stream) {
    if (!y || !stats || !gn_gamma || !gn_beta) VDX_FAIL(VDX_ERR_INVALID, "gn_silu_apply: null tensor");
    if (batch < 1 || pix_per_sample < 1 || c < 8 || c % 8 || c > 1024 || groups <= 0 || groups > 32 || c % groups || (scale_shift && ss_stride < 2 * c))
        VDX_FAIL(VDX_ERR_INVALID, "gn_silu_apply: shape not served");
    VDX_HIP(vdx::launch_gn_silu_apply16((float*)y, stats, gn_gamma, gn_beta, scale_shift, ss_stride, groups, c, batch, pix_per_sample, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_init_conv(const float* x, const float* kernel, const float* bias, float* y, int batch, int cin, int frames,
                  int h, int w, int cout, int k, void* stream) {
    if (!x || !kernel || !bias || !y) VDX_FAIL(VDX_ERR_INVALID, "init_conv: null tensor");
    if (k < 1 || k > 15 || !(k & 1) || cin < 1 || cin > 8) VDX_FAIL(VDX_ERR_INVALID, "init_conv: bad kernel size / channels");
    VDX_HIP(vdx::launch_init_conv(x, kernel, bias, y, batch, cin, frames, h, w, cout, k, 0, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_final_conv(const float* x, const float* kernel, const float* bias, float* y, long npix, int d, int cout, void* stream) {
    if (!x || !kernel || !bias || !y || d % 4) VDX_FAIL(VDX_ERR_INVALID, "final_conv: bad argument");
    VDX_HIP(vdx::launch_final_conv(x, kernel, bias, y, npix, d, cout, 0, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_time_mlp(const int* time, const float* w1, const float* b1, const float* w2, const float* b2, int dim,
                 const float* cond, const float* null_cond_emb, const unsigned char* cond_mask, int null_all, int cond_dim,
                 float* temb, int batch, void* stream) {
    if (!time || !w1 || !b1 || !w2 || !b2 || !temb || dim < 4 || (dim % 4)) VDX_FAIL(VDX_ERR_INVALID, "time_mlp: bad argument (dim must be a multiple of 4; Unet3D itself needs dim % 8 == 0)");
    if (cond_dim && (!cond || !null_cond_emb)) VDX_FAIL(VDX_ERR_INVALID, "time_mlp: cond missing");
    vdx::TimeMlpArgs a;
    memset(&a, 0, sizeof(a));
    a.time = time; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.dim = dim; a.time_dim = 4 * dim;
    a.cond = cond; a.null_cond_emb = null_cond_emb; a.cond_mask = cond_mask; a.null_all = null_all; a.cond_dim = cond_dim;
    a.temb = temb; a.temb_dim = 4 * dim + cond_dim;
    VDX_HIP(vdx::launch_time_mlp(a, batch, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_attention_forward(int mode, const float* x, float* y, const void* wqkv_packed, const float* bqkv,
                          const void* wo_packed, const float* bo, int batch, int frames, int h, int w, int c, int heads,
                          int temporal, void* stream) {
    return vdx_attention_forward_ex(mode, x, y, wqkv_packed, bqkv, wo_packed, bo, batch, frames, h, w, c, heads, temporal, 0, stream);
}

int vdx_attention_forward_ex(int mode, const float* x, float* y, const void* wqkv_packed, const float* bqkv,
                             const void* wo_packed, const float* bo, int batch, int frames, int h, int w, int c, int heads,
                             int temporal, int fp8_core, void* stream) {
    if (fp8_core && mode != VDX_MODE_BF16) VDX_FAIL(VDX_ERR_INVALID, "attention: the fp8 core needs VDX_MODE_BF16");
    if (!x || !y || !wqkv_packed || !bqkv || !wo_packed || !bo) VDX_FAIL(VDX_ERR_INVALID, "attention: null tensor");
    if (mode != VDX_MODE_F32 && mode != VDX_MODE_BF16 && mode != VDX_MODE_F16) VDX_FAIL(VDX_ERR_INVALID, "bad mode");
    if (c % 4 || c > 1024 || heads < 1) VDX_FAIL(VDX_ERR_INVALID, "attention: C must be a multiple of 4 and <= 1024");
    vdx::AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.wqkv = wqkv_packed; a.bqkv = bqkv; a.wo = wo_packed; a.bo = bo; a.C = c; a.heads = heads;
    a.scale = 1.0f / sqrtf(32.0f);
    const long hw = (long)h * w;
    if (temporal) {
        a.L = frames; a.nseq = (long)batch * hw; a.inner = hw; a.inner_stride = c; a.outer_stride = (long)frames * hw * c; a.tok_stride = hw * c;
    } else {
        a.L = (int)hw; a.nseq = (long)batch * frames; a.inner = 1; a.inner_stride = 0; a.outer_stride = hw * c; a.tok_stride = c;
    }
    if (a.L > 64) VDX_FAIL(VDX_ERR_INVALID, "attention: more than 64 tokens per sequence is not supported");
    a.fp8_core = fp8_core ? 1 : 0;
    VDX_HIP(vdx::launch_attention(mode, a, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_attention_forward_bf16(const void* x_bf16, void* y_bf16, const void* wqkv_packed, const float* bqkv,
                               const void* wo_packed, const float* bo, int batch, int frames, int h, int w, int c, int heads,
                               int temporal, int fp8_core, void* stream) {
    if (!x_bf16 || !y_bf16 || !wqkv_packed || !bqkv || !wo_packed || !bo) VDX_FAIL(VDX_ERR_INVALID, "attention: null tensor");
    if (c % 8 || c > 1024 || heads < 1) VDX_FAIL(VDX_ERR_INVALID, "attention (bf16 tensors): C must be a multiple of 8 and <= 1024");
    vdx::AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = reinterpret_cast<const float*>(x_bf16); a.y = reinterpret_cast<float*>(y_bf16); a.io_bf16 = 1;
    a.wqkv = wqkv_packed; a.bqkv = bqkv; a.wo = wo_packed; a.bo = bo; a.C = c; a.heads = heads;
    a.scale = 1.0f / sqrtf(32.0f);
    const long hw = (long)h * w;
    if (temporal) {
        a.L = frames; a.nseq = (long)batch * hw; a.inner = hw; a.inner_stride = c; a.outer_stride = (long)frames * hw * c; a.tok_stride = hw * c;
    } else {
        a.L = (int)hw; a.nseq = (long)batch * frames; a.inner = 1; a.inner_stride = 0; a.outer_stride = hw * c; a.tok_stride = c;
    }
    if (a.L > 64) VDX_FAIL(VDX_ERR_INVALID, "attention: more than 64 tokens per sequence is not supported");
    a.fp8_core = fp8_core ? 1 : 0;
    VDX_HIP(vdx::launch_attention(VDX_MODE_BF16, a, (hipStream_t)stream));
    return VDX_OK;
}

size_t vdx_sla_workspace_bytes(int mode, int nframes, int npix, int heads) { return vdx::sla_workspace_bytes(mode, nframes, npix, heads); }

int vdx_sla_forward(int mode, const float* x, float* y, const void* wq_packed, const void* wk_packed, const void* wv_packed,
                    const void* wo_packed, void* workspace, int batch, int frames, int h, int w, int c, int heads, void* stream) {
    if (!x || !y || !wq_packed || !wk_packed || !wv_packed || !wo_packed || !workspace) VDX_FAIL(VDX_ERR_INVALID, "sla: null tensor");
    if (mode != VDX_MODE_F32 && mode != VDX_MODE_BF16 && mode != VDX_MODE_F16) VDX_FAIL(VDX_ERR_INVALID, "bad mode");
    if (heads != 8 || c % 4 || c > 1024) VDX_FAIL(VDX_ERR_INVALID, "sla: needs 8 heads, C multiple of 4 and <= 1024");
    vdx::SlaArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.wq = wq_packed; a.wk = wk_packed; a.wv = wv_packed; a.wo = wo_packed; a.workspace = workspace;
    a.C = c; a.heads = heads; a.NF = batch * frames; a.N = h * w;
    VDX_HIP(vdx::launch_sla(mode, a, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_sla_forward_bf16(const void* x_bf16, void* y_bf16, const void* wq_packed, const void* wk_packed, const void* wv_packed,
                         const void* wo_packed, void* workspace, int batch, int frames, int h, int w, int c, int heads, void* stream) {
    if (!x_bf16 || !y_bf16 || !wq_packed || !wk_packed || !wv_packed || !wo_packed || !workspace) VDX_FAIL(VDX_ERR_INVALID, "sla: null tensor");
    if (heads != 8 || c % 8 || c > 1024) VDX_FAIL(VDX_ERR_INVALID, "sla (bf16 tensors): needs 8 heads, C multiple of 8 and <= 1024");
    vdx::SlaArgs a;
    memset(&a, 0, sizeof(a));
    a.x = reinterpret_cast<const float*>(x_bf16); a.y = reinterpret_cast<float*>(y_bf16); a.io_bf16 = 1;
    a.wq = wq_packed; a.wk = wk_packed; a.wv = wv_packed; a.wo = wo_packed; a.workspace = workspace;
    a.C = c; a.heads = heads; a.NF = batch * frames; a.N = h * w;
    VDX_HIP(vdx::launch_sla(VDX_MODE_BF16, a, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_create(const vdx_config* cfg, vdx_handle** out) {
    if (!cfg || !out) VDX_FAIL(VDX_ERR_INVALID, "create: null argument");
    if (cfg->mode != VDX_MODE_F32 && cfg->mode != VDX_MODE_BF16 && cfg->mode != VDX_MODE_F16) VDX_FAIL(VDX_ERR_INVALID, "bad mode");
    vdx_handle* h = new vdx_handle();
    h->model.cfg = *cfg;
    int rc = vdx::model_build(&h->model);
    if (rc != VDX_OK) { delete h; return rc; }
    // the only device memory the handle owns: the small ResnetBlock time-MLP table
    const size_t tb = h->model.ss_layers.size() * sizeof(vdx::SsLayer);
    if (tb) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) {
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&h->model.d_ss_layers), tb);
            if (e == hipSuccess) e = hipMemcpy(h->model.d_ss_layers, h->model.ss_layers.data(), tb, hipMemcpyHostToDevice);
            if (e != hipSuccess) { delete h; return vdx_set_error(VDX_ERR_HIP, hipGetErrorString(e), __FILE__, __LINE__); }
            vdx::model_build_pack_tables(&h->model);
            for (int k = 0; k < 2 && e == hipSuccess; ++k) {
                const std::vector<vdx::PackJob>& v = k ? h->model.pack_t_jobs : h->model.pack_jobs;
                vdx::PackJob** d = k ? &h->model.d_pack_t_jobs : &h->model.d_pack_jobs;
                e = hipMalloc(reinterpret_cast<void**>(d), v.size() * sizeof(vdx::PackJob));
                if (e == hipSuccess) e = hipMemcpy(*d, v.data(), v.size() * sizeof(vdx::PackJob), hipMemcpyHostToDevice);
            }
            if (e != hipSuccess) { vdx_destroy(h); return vdx_set_error(VDX_ERR_HIP, hipGetErrorString(e), __FILE__, __LINE__); }
        }   // without a device the handle still serves the layout queries (CPU-side tests)
    }
    *out = h;
    return VDX_OK;
}

void vdx_destroy(vdx_handle* h) {
    if (!h) return;
    h->drop_graphs();
    vdx::comm_destroy(&h->comm);
    vdx::bwd_state_free(&h->bwd);
    if (h->model.d_ss_layers) (void)hipFree(h->model.d_ss_layers);
    if (h->model.d_pack_jobs) (void)hipFree(h->model.d_pack_jobs);
    if (h->model.d_pack_t_jobs) (void)hipFree(h->model.d_pack_t_jobs);
    delete h;
}

int vdx_set_activation_storage(vdx_handle* h, int bf16) {
    if (!h) VDX_FAIL(VDX_ERR_INVALID, "set_activation_storage: null handle");
    if (bf16 && h->model.mode != VDX_MODE_BF16) VDX_FAIL(VDX_ERR_INVALID, "bf16 activation storage needs a VDX_MODE_BF16 handle");
    if (bf16 < 0 || bf16 > 2) VDX_FAIL(VDX_ERR_INVALID, "set_activation_storage: 0 (fp32), 1 (bf16, inference) or 2 (bf16, training forward)");
    const int v = bf16;
    if (v != h->model.act16) {                       // a cached sampling graph was captured with the other storage
        h->drop_graphs();
        h->model.act16 = v;
    }
    return VDX_OK;
}
int vdx_get_activation_storage(const vdx_handle* h) { return h ? h->model.act16 : 0; }

int vdx_set_attention_fp8(vdx_handle* h, int on) {
    if (!h) VDX_FAIL(VDX_ERR_INVALID, "set_attention_fp8: null handle");
    if (on && h->model.mode != VDX_MODE_BF16) VDX_FAIL(VDX_ERR_INVALID, "fp8 attention needs a VDX_MODE_BF16 handle");
    const int v = on ? 1 : 0;
    if (v != h->model.attn_fp8) { h->drop_graphs(); h->model.attn_fp8 = v; }
    return VDX_OK;
}
int vdx_get_attention_fp8(const vdx_handle* h) { return h ? h->model.attn_fp8 : 0; }

int vdx_param_count(const vdx_handle* h) { return h ? (int)h->model.params.size() : 0; }
long vdx_param_total(const vdx_handle* h) { return h ? h->model.param_total : 0; }

int vdx_param_info(const vdx_handle* h, int index, char* name, int name_cap, int* ndim, long shape[6], long* offset) {
    if (!h || index < 0 || index >= (int)h->model.params.size()) VDX_FAIL(VDX_ERR_INVALID, "param_info: bad index");
    const vdx::ParamInfo& p = h->model.params[index];
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", p.name.c_str());
    if (ndim) *ndim = p.ndim;
    if (shape) for (int i = 0; i < 6; ++i) shape[i] = i < p.ndim ? p.shape[i] : 1;
    if (offset) *offset = p.offset;
    return VDX_OK;
}

size_t vdx_packed_bytes(const vdx_handle* h) { return h ? h->model.packed_bytes : 0; }

int vdx_pack_params(const vdx_handle* h, const float* params, void* packed, void* stream) {
    if (!h || !params || !packed) VDX_FAIL(VDX_ERR_INVALID, "pack_params: null argument");
    VDX_HIP(vdx::model_pack(&h->model, params, packed, (hipStream_t)stream));
    return VDX_OK;
}

size_t vdx_workspace_bytes(const vdx_handle* h, int batch) { return h ? vdx::model_workspace_bytes(&h->model, batch) : 0; }
int vdx_slot_count(const vdx_handle* h) { return h ? (int)h->model.slots.size() : 0; }

int vdx_slot_info(const vdx_handle* h, int index, char* name, int name_cap, long* floats_per_sample, long* float_offset_per_sample) {
    if (!h || index < 0 || index >= (int)h->model.slots.size()) VDX_FAIL(VDX_ERR_INVALID, "slot_info: bad index");
    const vdx::Slot& s = h->model.slots[index];
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", s.name.c_str());
    if (floats_per_sample) *floats_per_sample = s.floats_per_sample;
    if (float_offset_per_sample) *float_offset_per_sample = s.offset_per_sample;
    return VDX_OK;
}

int vdx_unet_forward(const vdx_handle* h, const float* params, const void* packed, const float* x, const int* time,
                     const float* cond, const unsigned char* cond_mask, int null_all, float* out, void* workspace,
                     size_t workspace_bytes, int batch, void* stream) {
    if (!h || !params || !packed || !x || !time || !out || !workspace) VDX_FAIL(VDX_ERR_INVALID, "unet_forward: null argument");
    if (!h->model.d_ss_layers) VDX_FAIL(VDX_ERR_STATE, "unet_forward: handle was created without a GPU");
    return vdx::model_forward(&h->model, params, packed, x, time, cond, cond_mask, null_all, out, workspace, workspace_bytes, batch,
                              (hipStream_t)stream);
}

int vdx_randn(float* out, long n, uint64_t seed, uint64_t offset, const uint64_t* dev_offset, void* stream) {
    if (n == 0) return VDX_OK;
    if (!out || n < 0) VDX_FAIL(VDX_ERR_INVALID, "randn: bad argument");
    VDX_HIP(vdx::launch_randn(out, n, seed, offset, reinterpret_cast<const unsigned long long*>(dev_offset), (hipStream_t)stream));
    return VDX_OK;
}

int vdx_q_sample(const float* x_start, const int* t, const float* noise, float* out, const float* sqrt_ac,
                 const float* sqrt_one_minus_ac, int batch, long per_sample, float pre_scale, float pre_shift, void* stream) {
    if (!x_start || !t || !noise || !out || !sqrt_ac || !sqrt_one_minus_ac || batch < 1 || per_sample < 1) VDX_FAIL(VDX_ERR_INVALID, "q_sample: bad argument");
    if (per_sample % 4) VDX_FAIL(VDX_ERR_INVALID, "q_sample: per_sample must be a multiple of 4");
    VDX_HIP(vdx::launch_q_sample(x_start, t, noise, out, sqrt_ac, sqrt_one_minus_ac, batch, per_sample, pre_scale, pre_shift, (hipStream_t)stream));
    return VDX_OK;
}

static int fill_psample(vdx::PSampleArgs& a, const float* x, const float* eps_hat, float* out, const int* t, const float* tables,
                        int timesteps, const float* noise, uint64_t seed, uint64_t offset, const uint64_t* dev_offset,
                        const float* thres, int clip, int channels, long per_sample) {
    memset(&a, 0, sizeof(a));
    a.x = x; a.eps = eps_hat; a.out = out; a.t = t; a.tables = tables; a.T = timesteps; a.noise = noise;
    a.seed = seed; a.offset = offset; a.dev_offset = reinterpret_cast<const unsigned long long*>(dev_offset);
    a.thres = thres; a.clip = clip; a.C = channels; a.per_sample = per_sample; a.post_scale = 1.f; a.post_shift = 0.f;
    return 0;
}

int vdx_p_sample_step(const float* x, const float* eps_hat, float* out, const int* t, const float* tables, int timesteps,
                      const float* noise, uint64_t seed, uint64_t offset, const uint64_t* dev_offset, const float* thres,
                      int clip_denoised, int batch, int channels, long per_sample, void* stream) {
    if (!x || !eps_hat || !out || !t || !tables || timesteps < 1 || batch < 1 || channels < 1) VDX_FAIL(VDX_ERR_INVALID, "p_sample: bad argument");
    if (per_sample % channels) VDX_FAIL(VDX_ERR_INVALID, "p_sample: per_sample must be a multiple of channels");
    if (!noise && per_sample % 4) VDX_FAIL(VDX_ERR_INVALID, "p_sample: generated noise needs per_sample % 4 == 0");
    vdx::PSampleArgs a;
    fill_psample(a, x, eps_hat, out, t, tables, timesteps, noise, seed, offset, dev_offset, thres, clip_denoised, channels, per_sample);
    VDX_HIP(vdx::launch_p_sample(a, batch, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_loss_sum(const float* eps_hat, const float* noise, double* acc, int batch, int channels, long fhw, int l2, void* stream) {
    if (!eps_hat || !noise || !acc) VDX_FAIL(VDX_ERR_INVALID, "loss: null tensor");
    VDX_HIP(vdx::launch_loss(eps_hat, noise, acc, batch, channels, fhw, l2, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_affine(const float* x, float* y, long n, float a, float b, void* stream) {
    if (!x || !y || n < 0) VDX_FAIL(VDX_ERR_INVALID, "affine: bad argument");
    if (n) VDX_HIP(vdx::launch_affine(x, y, n, a, b, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_dynamic_threshold(const float* x, const float* eps_hat, const int* t, const float* tables, int timesteps, float percentile,
                          float* thres_out, int batch, int channels, long per_sample, void* stream) {
    if (!x || !eps_hat || !t || !tables || !thres_out || batch < 1 || channels < 1 || per_sample < 1) VDX_FAIL(VDX_ERR_INVALID, "dynamic_threshold: bad argument");
    if (!(percentile > 0.f && percentile <= 1.f) || per_sample % channels) VDX_FAIL(VDX_ERR_INVALID, "dynamic_threshold: percentile must be in (0, 1]");
    VDX_HIP(vdx::launch_dyn_thres(x, eps_hat, t, tables, timesteps, percentile, thres_out, batch, channels, per_sample, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_p_sample_loop_dyn(vdx_handle* h, const float* params, const void* packed, float* img, float* eps_buf, int* t_dev,
                          uint64_t* step_dev, const float* tables, int timesteps, int nsteps, const float* cond, uint64_t seed,
                          int clip_denoised, float percentile, float* thres_buf, void* workspace, size_t workspace_bytes, int batch,
                          int use_graph, void* stream) {
    if (!h || !params || !packed || !img || !eps_buf || !t_dev || !step_dev || !tables || !workspace) VDX_FAIL(VDX_ERR_INVALID, "p_sample_loop: null argument");
    if (!h->model.d_ss_layers) VDX_FAIL(VDX_ERR_STATE, "p_sample_loop: handle was created without a GPU");
    if (nsteps < 0 || nsteps > timesteps) VDX_FAIL(VDX_ERR_INVALID, "p_sample_loop: nsteps out of range");
    const bool dyn = percentile > 0.f && clip_denoised;
    if (dyn && (!thres_buf || percentile > 1.f)) VDX_FAIL(VDX_ERR_INVALID, "p_sample_loop: dynamic threshold needs thres_buf and a percentile in (0, 1]");
    const vdx::Model& m = h->model;
    const long per_sample = (long)m.cfg.channels * m.cfg.num_frames * m.cfg.image_size * m.cfg.image_size;
    if (per_sample % 4) VDX_FAIL(VDX_ERR_INVALID, "p_sample_loop: C*F*H*W must be a multiple of 4");
    if (m.out_dim != m.cfg.channels) VDX_FAIL(VDX_ERR_INVALID, "p_sample_loop: out_dim must equal channels");
    hipStream_t st = (hipStream_t)stream;
    auto step = [&]() -> int {
        // reference p_sample_loop never forwards cond (Q10); when a cond tensor is supplied it is used un-masked
        int rc = vdx::model_forward(&m, params, packed, img, t_dev, cond, nullptr, 0, eps_buf, workspace, workspace_bytes, batch, st);
        if (rc != VDX_OK) return rc;
        hipError_t e = hipSuccess;
        if (dyn) e = vdx::launch_dyn_thres(img, eps_buf, t_dev, tables, timesteps, percentile, thres_buf, batch, m.cfg.channels, per_sample, st);
        vdx::PSampleArgs a;
        fill_psample(a, img, eps_buf, img, t_dev, tables, timesteps, nullptr, seed, 1, step_dev, dyn ? thres_buf : nullptr, clip_denoised, m.cfg.channels, per_sample);
        if (e == hipSuccess) e = vdx::launch_p_sample(a, batch, st);
        if (e == hipSuccess) e = vdx::launch_advance(t_dev, batch, reinterpret_cast<unsigned long long*>(step_dev), st);
        if (e != hipSuccess) return vdx_set_error(VDX_ERR_HIP, hipGetErrorString(e), __FILE__, __LINE__);
        return VDX_OK;
    };
    vdx_handle::GraphKey key;
    memset(&key, 0, sizeof(key));
    key.p[0] = params; key.p[1] = packed; key.p[2] = img; key.p[3] = eps_buf; key.p[4] = t_dev; key.p[5] = step_dev;
    key.p[6] = tables; key.p[7] = cond; key.p[8] = workspace; key.p[9] = stream; key.p[10] = dyn ? thres_buf : nullptr;
    key.seed = seed; key.i[0] = timesteps; key.i[1] = clip_denoised | (h->model.act16 << 8); key.i[2] = batch;
    key.ws = workspace_bytes; key.f = dyn ? percentile : 0.f;
    return run_steps(h, 0, key, nsteps, use_graph, st, step);
}

int vdx_p_sample_loop(vdx_handle* h, const float* params, const void* packed, float* img, float* eps_buf, int* t_dev,
                      uint64_t* step_dev, const float* tables, int timesteps, int nsteps, const float* cond, uint64_t seed,
                      int clip_denoised, void* workspace, size_t workspace_bytes, int batch, int use_graph, void* stream) {
    return vdx_p_sample_loop_dyn(h, params, packed, img, eps_buf, t_dev, step_dev, tables, timesteps, nsteps, cond, seed, clip_denoised,
                                 0.f, nullptr, workspace, workspace_bytes, batch, use_graph, stream);
}

int vdx_ddim_step(const float* x, const float* eps_hat, float* out, const float* alphas_cumprod, const int* seq,
                  const uint64_t* step_dev, const float* thres, int clip_denoised, int batch, int channels, long per_sample, void* stream) {
    if (!x || !eps_hat || !out || !alphas_cumprod || !seq || batch < 1 || channels < 1 || per_sample < 1 || per_sample % channels)
        VDX_FAIL(VDX_ERR_INVALID, "ddim_step: bad argument");
    VDX_HIP(vdx::launch_ddim_step(x, eps_hat, out, alphas_cumprod, seq, reinterpret_cast<const unsigned long long*>(step_dev), thres,
                                  clip_denoised, batch, channels, per_sample, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_ddim_sample_loop_dyn(vdx_handle* h, const float* params, const void* packed, float* img, float* eps_buf, int* t_dev,
                             uint64_t* step_dev, const float* alphas_cumprod, const int* seq, int seq_len, int nsteps, const float* cond,
                             int clip_denoised, const float* tables, int timesteps, float percentile, float* thres_buf, void* workspace,
                             size_t workspace_bytes, int batch, int use_graph, void* stream) {
    if (!h || !params || !packed || !img || !eps_buf || !t_dev || !step_dev || !alphas_cumprod || !seq || !workspace) VDX_FAIL(VDX_ERR_INVALID, "ddim_sample_loop: null argument");
    if (!h->model.d_ss_layers) VDX_FAIL(VDX_ERR_STATE, "ddim_sample_loop: handle was created without a GPU");
    if (seq_len < 1 || nsteps < 0 || nsteps > seq_len) VDX_FAIL(VDX_ERR_INVALID, "ddim_sample_loop: nsteps out of range");
    const bool dyn = percentile > 0.f && clip_denoised;
    if (dyn && (!thres_buf || !tables || timesteps < 1 || percentile > 1.f)) VDX_FAIL(VDX_ERR_INVALID, "ddim_sample_loop: dynamic threshold needs tables, thres_buf and a percentile in (0, 1]");
    const vdx::Model& m = h->model;
    const long per_sample = (long)m.cfg.channels * m.cfg.num_frames * m.cfg.image_size * m.cfg.image_size;
    if (m.out_dim != m.cfg.channels) VDX_FAIL(VDX_ERR_INVALID, "ddim_sample_loop: out_dim must equal channels");
    hipStream_t st = (hipStream_t)stream;
    auto step = [&]() -> int {
        int rc = vdx::model_forward(&m, params, packed, img, t_dev, cond, nullptr, 0, eps_buf, workspace, workspace_bytes, batch, st);
        if (rc != VDX_OK) return rc;
        hipError_t e = hipSuccess;
        if (dyn) e = vdx::launch_dyn_thres(img, eps_buf, t_dev, tables, timesteps, percentile, thres_buf, batch, m.cfg.channels, per_sample, st);
        if (e == hipSuccess) e = vdx::launch_ddim_step(img, eps_buf, img, alphas_cumprod, seq, reinterpret_cast<const unsigned long long*>(step_dev),
                                                       dyn ? thres_buf : nullptr, clip_denoised, batch, m.cfg.channels, per_sample, st);
        if (e == hipSuccess) e = vdx::launch_ddim_advance(t_dev, batch, seq, reinterpret_cast<unsigned long long*>(step_dev), st);
        if (e != hipSuccess) return vdx_set_error(VDX_ERR_HIP, hipGetErrorString(e), __FILE__, __LINE__);
        return VDX_OK;
    };
    vdx_handle::GraphKey key;
    memset(&key, 0, sizeof(key));
    key.p[0] = params; key.p[1] = packed; key.p[2] = img; key.p[3] = eps_buf; key.p[4] = t_dev; key.p[5] = step_dev;
    key.p[6] = alphas_cumprod; key.p[7] = cond; key.p[8] = workspace; key.p[9] = stream; key.p[10] = seq; key.p[11] = dyn ? (const void*)thres_buf : nullptr;
    key.i[0] = seq_len; key.i[1] = clip_denoised | (h->model.act16 << 8); key.i[2] = batch; key.i[3] = dyn ? timesteps : 0; key.ws = workspace_bytes;
    key.f = dyn ? percentile : 0.f; key.seed = dyn ? (unsigned long long)(uintptr_t)tables : 0ull;
    return run_steps(h, 1, key, nsteps, use_graph, st, step);
}

int vdx_ddim_sample_loop(vdx_handle* h, const float* params, const void* packed, float* img, float* eps_buf, int* t_dev,
                         uint64_t* step_dev, const float* alphas_cumprod, const int* seq, int seq_len, int nsteps, const float* cond,
                         int clip_denoised, void* workspace, size_t workspace_bytes, int batch, int use_graph, void* stream) {
    return vdx_ddim_sample_loop_dyn(h, params, packed, img, eps_buf, t_dev, step_dev, alphas_cumprod, seq, seq_len, nsteps, cond, clip_denoised,
                                    nullptr, 0, 0.f, nullptr, workspace, workspace_bytes, batch, use_graph, stream);
}

int vdx_pack_conv_weights_t(int mode, const float* kernel, void* packed, int taps, int cin, int cout, void* stream) {
    if (!kernel || !packed || taps <= 0 || cin <= 0 || cout <= 0) VDX_FAIL(VDX_ERR_INVALID, "pack_conv_weights_t: bad argument");
    if (mode != VDX_MODE_F32 && mode != VDX_MODE_BF16 && mode != VDX_MODE_F16) VDX_FAIL(VDX_ERR_INVALID, "bad mode");
    VDX_HIP(vdx::launch_pack_weights_t(mode, kernel, packed, taps, cin, cout, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_conv_backward_weights(const vdx_wgrad_desc* d, void* stream) {
    if (!d || !d->x0 || !d->dy || !d->dw) VDX_FAIL(VDX_ERR_INVALID, "wgrad: null tensor");
    if (d->c0 % 4 || d->c1 % 4 || d->cout % 4 || (d->c1 && !d->x1)) VDX_FAIL(VDX_ERR_INVALID, "wgrad: bad channels");
    vdx::WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x0 = d->x0; a.x1 = d->x1; a.C0 = d->c0; a.C1 = d->c1; a.dy = d->dy; a.Cout = d->cout; a.dW = d->dw;
    a.NF = d->batch * d->frames; a.F = d->frames; a.H = d->h; a.W = d->w;
    a.kind = d->kind; a.kh = d->kind ? 4 : d->kh; a.kw = d->kind ? 4 : d->kw; a.stride = d->kind ? 1 : d->stride;
    if (d->kind == 0 && (d->kh != d->kw || d->kh < 1 || d->kh > 4 || (d->stride != 1 && d->stride != 2))) VDX_FAIL(VDX_ERR_INVALID, "wgrad: unsupported kernel/stride");
    if (d->in_stats) {
        if (d->c1 || !d->gamma || !d->beta || d->groups <= 0 || d->groups > 32) VDX_FAIL(VDX_ERR_INVALID, "wgrad: bad prologue");
        a.pro = 1; a.in_stats = d->in_stats; a.gamma = d->gamma; a.beta = d->beta; a.groups = d->groups; a.ss = d->scale_shift; a.ss_stride = d->scale_shift_stride;
    }
    a.bf16_mma = d->bf16_operands ? 1 : 0;
    VDX_HIP(vdx::launch_conv_wgrad(a, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_norm_act_backward(const float* dact, const float* y, float* dy, const double* stats, const float* gamma, const float* beta,
                          int groups, const float* scale_shift, int scale_shift_stride, float* d_gamma, float* d_beta, float* dss,
                          const float* r, const float* ln_gamma, float* dr, float* d_ln_gamma, float* d_ln_beta, float* scratch,
                          int c, int batch, long pix_per_sample, void* stream) {
    if (!dact || !y || !dy || !stats || !gamma || !beta || !d_gamma || !d_beta || !scratch) VDX_FAIL(VDX_ERR_INVALID, "norm_act_backward: null tensor");
    if (r && (!ln_gamma || !dr || !d_ln_gamma || !d_ln_beta)) VDX_FAIL(VDX_ERR_INVALID, "norm_act_backward: incomplete LayerNorm branch");
    if (c % 4 || c > 1024 || groups < 1 || groups > 32 || c % groups) VDX_FAIL(VDX_ERR_INVALID, "norm_act_backward: bad channels/groups");
    vdx::NormBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.dact = dact; a.y = y; a.dy = dy; a.stats = stats; a.gamma = gamma; a.beta = beta; a.groups = groups;
    a.ss = scale_shift; a.ss_stride = scale_shift_stride; a.d_gamma = d_gamma; a.d_beta = d_beta; a.dss = dss;
    a.r = r; a.ln_gamma = ln_gamma; a.dr = dr; a.d_ln_gamma = d_ln_gamma; a.d_ln_beta = d_ln_beta;
    a.R = scratch; a.G = scratch + (vdx::norm_bwd_scratch_floats(c, batch, pix_per_sample) - (size_t)batch * 64); a.C = c; a.batch = batch; a.pix_per_sample = pix_per_sample;
    VDX_HIP(vdx::launch_norm_bwd(a, (hipStream_t)stream));
    return VDX_OK;
}

size_t vdx_norm_act_backward_scratch_floats(int c, int batch, long pix_per_sample) {
    if (c < 4 || c % 4 || c > 1024 || batch < 1 || pix_per_sample < 1) return 0;
    return vdx::norm_bwd_scratch_floats(c, batch, pix_per_sample);
}

int vdx_attention_core_backward(const float* qkv, const float* d_o, float* o, float* dq, float* dk, float* dv, int batch, int frames,
                                int h, int w, int heads, int temporal, void* stream) {
    return vdx_attention_core_backward_ex(qkv, d_o, o, dq, dk, dv, batch, frames, h, w, heads, temporal, 0, stream);
}

int vdx_attention_core_backward_ex(const float* qkv, const float* d_o, float* o, float* dq, float* dk, float* dv, int batch, int frames,
                                   int h, int w, int heads, int temporal, int bf16_operands, void* stream) {
    if (!qkv || !d_o || !o || !dq || !dk || !dv || heads < 1) VDX_FAIL(VDX_ERR_INVALID, "attention_core_backward: bad argument");
    vdx::AttnBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.qkv = qkv; a.dO = d_o; a.O = o; a.dq = dq; a.dk = dk; a.dv = dv; a.heads = heads; a.scale = 1.0f / sqrtf(32.0f);
    const long hw = (long)h * w;
    if (temporal) { a.L = frames; a.nseq = (long)batch * hw; a.inner = hw; a.outer_p = (long)frames * hw; a.tok_p = hw; }
    else { a.L = (int)hw; a.nseq = (long)batch * frames; a.inner = 1; a.outer_p = hw; a.tok_p = 1; }
    if (a.L > 64) VDX_FAIL(VDX_ERR_INVALID, "attention_core_backward: more than 64 tokens per sequence");
    a.bf16_mma = bf16_operands ? 1 : 0;
    a.dstride = heads * 32;
    VDX_HIP(vdx::launch_attn_core_bwd(a, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_temporal_attention_backward_fused(const float* x, const float* dy, const void* packed_wqkv, const float* bqkv, const void* packed_wo_t,
                                          void* o_bf16, void* dqkv_bf16, float* dx, int batch, int frames, int h, int w, void* stream) {
    if (!x || !dy || !packed_wqkv || !bqkv || !packed_wo_t || !o_bf16 || !dqkv_bf16 || !dx) VDX_FAIL(VDX_ERR_INVALID, "temporal_attention_backward_fused: null argument");
    if (batch < 1 || frames < 1 || frames > 16 || h < 1 || w < 1) VDX_FAIL(VDX_ERR_INVALID, "temporal_attention_backward_fused: 1..16 frames");
    vdx::AttnBwdXArgs a;
    memset(&a, 0, sizeof(a));
    const long hw = (long)h * w;
    a.x = x; a.g = dy; a.wqkv = packed_wqkv; a.bqkv = bqkv; a.woT = packed_wo_t; a.O = o_bf16; a.dqkv = dqkv_bf16; a.dx = dx;
    a.L = frames; a.nseq = (long)batch * hw; a.inner = hw; a.outer_p = (long)frames * hw; a.tok_p = hw; a.scale = 1.0f / sqrtf(32.0f);
    VDX_HIP(vdx::launch_attn_bwd_fused(a, (hipStream_t)stream));
    return VDX_OK;
}

size_t vdx_sla_backward_scratch_floats(int nframes, int heads) { return vdx::sla_bwd_scratch_floats(nframes, heads); }

int vdx_sla_core_backward(const float* q, const float* k, const float* v, const float* d_out, float* o, float* dq, float* dk, float* dv,
                          float* scratch, int nframes, int npix, int heads, void* stream) {
    return vdx_sla_core_backward_ex(q, k, v, d_out, o, dq, dk, dv, scratch, nframes, npix, heads, 0, stream);
}

int vdx_sla_core_backward_ex(const float* q, const float* k, const float* v, const float* d_out, float* o, float* dq, float* dk, float* dv,
                             float* scratch, int nframes, int npix, int heads, int bf16_operands, void* stream) {
    if (!q || !k || !v || !d_out || !o || !dq || !dk || !dv || !scratch || heads != 8) VDX_FAIL(VDX_ERR_INVALID, "sla_core_backward: bad argument");
    vdx::SlaBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.q = q; a.k = k; a.v = v; a.dOut = d_out; a.O = o; a.dq = dq; a.dk = dk; a.dv = dv; a.A = scratch; a.NF = nframes; a.N = npix; a.heads = heads;
    a.bf16_mma = bf16_operands ? 1 : 0;
    a.dstride = 256;
    VDX_HIP(vdx::launch_sla_bwd(a, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_colsum(const float* x, float* out, long rows, int c, void* stream) {
    if (!x || !out || rows < 0 || c < 1 || c % 4) VDX_FAIL(VDX_ERR_INVALID, "colsum: bad argument");
    if (rows) VDX_HIP(vdx::launch_colsum(x, out, rows, c, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_num_stages(const vdx_handle* h) { return h ? 2 * h->model.cfg.n_mults + 3 : 0; }
size_t vdx_packed_bwd_bytes(const vdx_handle* h) { return h ? h->model.packed_t_bytes : 0; }
size_t vdx_bwd_workspace_bytes(const vdx_handle* h, int batch) { return h ? vdx::model_bwd_workspace_bytes(&h->model, batch) : 0; }

int vdx_pack_params_bwd(const vdx_handle* h, const float* params, void* packed_t, void* stream) {
    if (!h || !params || !packed_t) VDX_FAIL(VDX_ERR_INVALID, "pack_params_bwd: null argument");
    VDX_HIP(vdx::model_pack_t(&h->model, params, packed_t, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_unet_backward(vdx_handle* h, const float* params, const void* packed, const void* packed_t, const float* x, const int* time,
                      const float* cond, const unsigned char* cond_mask, int null_all, const float* d_out, void* fwd_workspace,
                      void* bwd_workspace, size_t bwd_workspace_bytes, float* grads, int stage_hi, int stage_lo, int batch, void* stream) {
    if (!h || !params || !packed || !packed_t || !x || !time || !d_out || !fwd_workspace || !bwd_workspace || !grads) VDX_FAIL(VDX_ERR_INVALID, "unet_backward: null argument");
    if (!h->model.d_ss_layers) VDX_FAIL(VDX_ERR_STATE, "unet_backward: handle was created without a GPU");
    if (h->model.act16 == 1) VDX_FAIL(VDX_ERR_STATE, "unet_backward: the forward ran with the INFERENCE form of bf16 activation storage (res_conv folded into the block tail); use vdx_set_activation_storage(h, 2) for a forward that feeds the backward");
    if (h->model.attn_fp8) VDX_FAIL(VDX_ERR_STATE, "unet_backward: fp8 attention is a forward (sampling) option; the backward differentiates the bf16 cores");
    // the fused q|k|v weight gradient (wgrad.hip, split = heads * 32) owns whole 64-wide output tiles per tensor
    if ((h->model.cfg.attn_heads * 32) % 64) VDX_FAIL(VDX_ERR_INVALID, "unet_backward: attn_heads must be even (heads * 32 a multiple of 64); odd head counts are forward / sampling only");
    return vdx::model_backward(&h->model, &h->bwd, params, packed, packed_t, x, time, cond, cond_mask, null_all, d_out, fwd_workspace,
                               bwd_workspace, bwd_workspace_bytes, grads, stage_hi, stage_lo, batch, (hipStream_t)stream);
}

int vdx_loss_grad(const float* eps_hat, const float* noise, float* d_eps_hat, int batch, int channels, long fhw, int l2, void* stream) {
    if (!eps_hat || !noise || !d_eps_hat || batch < 1 || channels < 1 || fhw < 1) VDX_FAIL(VDX_ERR_INVALID, "loss_grad: bad argument");
    VDX_HIP(vdx::launch_loss_grad(eps_hat, noise, d_eps_hat, batch, channels, fhw, l2, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_adam_ema_step(float* params, const float* grads, float* m, float* v, float* ema, long n, float lr, float b1, float b2,
                      float eps, long step_count, float grad_scale, int do_ema, float ema_decay, void* stream) {
    if (!params || !grads || !m || !v || (do_ema && !ema) || n < 1 || step_count < 0) VDX_FAIL(VDX_ERR_INVALID, "adam_ema_step: bad argument");
    VDX_HIP(vdx::launch_adam_ema(params, grads, m, v, ema, n, lr, b1, b2, eps, step_count, grad_scale, do_ema, ema_decay, (hipStream_t)stream));
    return VDX_OK;
}

int vdx_comm_unique_id(void* unique_id_out) {
    if (!unique_id_out) VDX_FAIL(VDX_ERR_INVALID, "comm_unique_id: null argument");
    return vdx::comm_unique_id(unique_id_out);
}

int vdx_comm_init(vdx_handle* h, int rank, int world, const void* unique_id) {
    if (!h || !unique_id || world < 1 || rank < 0 || rank >= world) VDX_FAIL(VDX_ERR_INVALID, "comm_init: bad argument");
    return vdx::comm_init(&h->comm, rank, world, unique_id);
}

int vdx_allreduce_bucket(vdx_handle* h, float* ptr, size_t count, void* stream) {
    if (!h || !ptr) VDX_FAIL(VDX_ERR_INVALID, "allreduce_bucket: null argument");
    if (count == 0) return VDX_OK;
    return vdx::comm_allreduce(&h->comm, ptr, count, (hipStream_t)stream);
}

int vdx_comm_world(const vdx_handle* h) { return h ? h->comm.world : 1; }

int vdx_comm_destroy(vdx_handle* h) {
    if (!h) VDX_FAIL(VDX_ERR_INVALID, "comm_destroy: null handle");
    vdx::comm_destroy(&h->comm);
    return VDX_OK;
}

}  // extern "C"
