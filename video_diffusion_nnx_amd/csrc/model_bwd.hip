// Reverse pass of the Unet3D hot path (host orchestration).  Mirrors model.hip's forward walk backwards
// (reference: jax.value_and_grad over unet3d.py:262-387, trainer.py:361).  Every GEMM-shaped step reuses the forward
// launchers: data gradients = conv_igemm with transposed/flipped packed weights, weight gradients = conv_wgrad,
// projections of the attention blocks = 1x1 convs around the small per-sequence cores (attn_bwd.hip).
#include <math.h>
#include <string.h>
#include <vector>

#include "vdx_common.h"
#include "vdx_internal.h"
#include "model.h"

namespace vdx {

namespace {

struct LevelBufs { float* ga; float* gb; float* t1; float* t2; float* t3; float* t4; float* gskip; };
constexpr int LVL_BUFS = 7;

struct Bwd {
    const Model* m; const float* p; const char* pk; const char* pt; float* grads; int B; hipStream_t st;
    int a16 = 0;                                    // the forward stored its inter-kernel activations as bf16 (Model::act16 == 2): every slot read here is a bf16 tensor
    // forward workspace views
    float* act; float* temb; float* ss; float* ss_lin; double* stats;
    // backward workspace views
    std::vector<LevelBufs> lv; float* gr; float* S; float* Sbuf[2]; int s_cur = 0; float* dss; float* dtemb; float* normscr; float* sla_a;
    float* part_side; float* part_main;             // slots of the deterministic accumulations (WG_PART_FLOATS each): one per stream, used in stream order
    float* slot(int s) const { return act + (size_t)m->slots[s].offset_per_sample * B; }
    double* stat(int i) const { return stats + (size_t)i * B * GN_SLOTS * m->cfg.resnet_groups * 2; }
    long pix(int lvl) const { const long s = m->cfg.image_size >> lvl; return (long)m->cfg.num_frames * s * s; }
    int size(int lvl) const { return m->cfg.image_size >> lvl; }
    hipError_t err = hipSuccess;
    bool ok(hipError_t e) { if (e != hipSuccess && err == hipSuccess) err = e; return e == hipSuccess; }
    // the attention scratch of the next block: two of them, used alternately -- the weight gradients of a block (side stream) read its
    // scratch while the next block's data-gradient chain already fills the other one (one scratch: the main stream idled 60-240 us
    // behind every attention block, ~1.1 ms per train step at the N shape)
    float* next_scratch() { s_cur ^= 1; S = Sbuf[s_cur]; return S; }

    // ---- second stream ------------------------------------------------------------------------------------------------------
    // Weight gradients are leaves of the reverse graph: nothing of the pass reads them, so they run on the handle's side stream
    // beside the data-gradient chain (at batch 4 per GPU most kernels leave CUs or wave slots idle; DESIGN.md, backward).
    // Edges: a side launch waits for everything enqueued on the main stream so far (side()); the main stream waits for the side
    // stream before it overwrites a buffer a pending side kernel reads (writes()), and at the end of every call (join()): the
    // parameter gradients of the stages a call ran are final on `st` when it returns, as the data-parallel reducer assumes.
    BwdState* state = nullptr;
    struct Pending { const void* buf; hipEvent_t done; int seq; };
    std::vector<Pending> pending;                   // buffers read by side kernels the main stream has not waited for
    bool main_dirty = true; int seq = 0;
    hipEvent_t next_event() { hipEvent_t e = state->ev[state->ev_next]; state->ev_next = (state->ev_next + 1) % BWD_EVENTS; return e; }
    // the stream of a leaf kernel; side_done() after its launch names the buffers it reads
    hipStream_t side() {
        if (!state->side) return st;
        if ((int)pending.size() > BWD_EVENTS / 4) join();           // (an event of the ring is never reused while an entry still names it)
        if (main_dirty) { hipEvent_t e = next_event(); ok(hipEventRecord(e, st)); ok(hipStreamWaitEvent(state->side, e, 0)); main_dirty = false; }
        return state->side;
    }
    void side_done(const void* r0, const void* r1 = nullptr) {
        if (!state->side) return;
        hipEvent_t e = next_event(); ok(hipEventRecord(e, state->side)); ++seq;
        if (r0) pending.push_back({r0, e, seq});
        if (r1) pending.push_back({r1, e, seq});
    }
    void wait_side(int upto, hipEvent_t e) {                         // the side stream runs in order: waiting for one kernel covers the earlier ones
        ok(hipStreamWaitEvent(st, e, 0));
        size_t k = 0;
        for (const Pending& p : pending) if (p.seq > upto) pending[k++] = p;
        pending.resize(k);
    }
    void join() { if (!pending.empty()) wait_side(pending.back().seq, pending.back().done); }
    // the main stream is about to launch a kernel that writes w0 / w1
    hipStream_t writes(const void* w0, const void* w1 = nullptr) {
        const Pending* last = nullptr;
        for (const Pending& p : pending) if (p.buf == w0 || (w1 && p.buf == w1)) last = &p;
        if (last) wait_side(last->seq, last->done);
        main_dirty = true;
        return st;
    }
};

// widest tensor a level's backward buffers hold: its own width, or the width of the level that feeds it (the input gradient of
// downs[l].res0 and the output of ups[.].res0 have dims[l] channels, which exceeds dims[l+1] for non-monotone dim_mults; the
// stem width at level 0)
size_t lvl_width(const Model* m, int l) {
    const size_t w = (size_t)m->cfg.dim * m->cfg.dim_mults[l];
    return std::max(w, l == 0 ? (size_t)m->init_dim : (size_t)m->cfg.dim * m->cfg.dim_mults[l - 1]);
}

// data gradient through a conv: out[.., cin_rows] = conv(dy; transposed packing rows [row0, row0+nrows)) (+ res)
void dgrad(Bwd& b, const float* dy, int Cdy, const void* wpt, int rows_total, int row0, int nrows, int lvl_in, int kind_fwd, int k,
           int stride_fwd, const float* res, float* out, int dy_bf16 = 0) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x0 = dy; a.C0 = Cdy; a.x0_bf16 = dy_bf16; a.wp = wpt; a.y = out; a.Cout = nrows; a.wrows = rows_total; a.wrow0 = row0; a.res = res;
    a.NF = b.B * b.m->cfg.num_frames; a.F = b.m->cfg.num_frames;
    // lvl_in = level of the FORWARD conv's input; dy lives at the forward output resolution
    if (kind_fwd == 0 && stride_fwd == 1) { a.H = a.W = b.size(lvl_in); a.kind = 0; a.kh = a.kw = k; a.stride = 1; a.pad = (k - 1) / 2; }
    else if (kind_fwd == 0) { a.H = a.W = b.size(lvl_in + 1); a.kind = 1; a.kh = a.kw = 4; a.stride = 1; }                      // Downsample -> ConvTranspose
    else { a.H = a.W = b.size(lvl_in - 1); a.kind = 0; a.kh = a.kw = 4; a.stride = 2; a.pad = 1; }                               // Upsample -> stride-2 conv
    b.ok(launch_conv(b.m->mode, a, b.writes(out)));
}

void wgrad(Bwd& b, const float* x0, int c0, const float* x1, int c1, const float* dy, int Cout, long w_off, long b_off, int lvl_in, int kind, int k, int stride,
           const double* in_stats = nullptr, const float* gamma = nullptr, const float* beta = nullptr, const float* ss = nullptr, int ss_stride = 0,
           int dy_bf16 = 0) {
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.dy_bf16 = dy_bf16;
    a.x0_bf16 = b.a16;                                             // x0 / x1 are forward slots (block inputs); y1 (prologue form) is bf16 in bf16 mode either way
    a.x0 = x0; a.x1 = x1; a.C0 = c0; a.C1 = c1; a.dy = dy; a.Cout = Cout; a.dW = b.grads + w_off; a.db = b_off >= 0 ? b.grads + b_off : nullptr;
    a.NF = b.B * b.m->cfg.num_frames; a.F = b.m->cfg.num_frames; a.H = a.W = b.size(lvl_in);
    a.kind = kind; a.kh = a.kw = kind ? 4 : k; a.stride = kind ? 1 : stride;
    a.bf16_mma = (b.m->mode == MODE_BF16);
    if (in_stats) { a.x0_bf16 = (b.m->mode == MODE_BF16); a.pro = 1; a.in_stats = in_stats; a.gamma = gamma; a.beta = beta; a.groups = b.m->cfg.resnet_groups; a.ss = ss; a.ss_stride = ss_stride; }
    a.part = b.part_side; a.part_cap = WG_PART_FLOATS;             // (every weight gradient runs on the side stream, in order)
    b.ok(launch_conv_wgrad(a, b.side()));
    b.side_done(dy);
}


// ResnetBlock backward.  g = dL/d(out) [pix][cout]; writes dL/d(x0) to out0 ([pix][c0]) and dL/d(x1) to out1 ([pix][c1], if c1)
void res_bwd(Bwd& b, const ResP& r, const float* g, const float* x0, int c0, const float* x1, int c1, int lvl, float* out0, float* out1) {
    const Model* m = b.m;
    const int G = m->cfg.resnet_groups;
    const long npix = b.pix(lvl) * b.B;
    LevelBufs& L = b.lv[lvl];
    const float* rsrc = r.has_res ? b.slot(r.s_rc) : x0;
    // 1. tail: out = SiLU(GN2(y2)) + LN(r)
    NormBwdArgs t;
    memset(&t, 0, sizeof(t));
    const int d16 = (m->mode == MODE_BF16) ? 1 : 0;      // bf16 mode: dL/d(y2), dL/d(y1) are bf16 tensors (only convolutions read them)
    t.dact = g; t.y = b.slot(r.s_y2); t.y_bf16 = (m->mode == MODE_BF16); t.dy = L.t1; t.dy_bf16 = d16; t.stats = b.stat(r.st2); t.gamma = b.p + r.b2_gs; t.beta = b.p + r.b2_gb; t.groups = G;
    t.d_gamma = b.grads + r.b2_gs; t.d_beta = b.grads + r.b2_gb;
    t.r = rsrc; t.r_bf16 = b.a16; t.ln_gamma = b.p + r.n2_s; t.dr = L.t2; t.d_ln_gamma = b.grads + r.n2_s; t.d_ln_beta = b.grads + r.n2_b;
    // R at a fixed offset of the scratch: zeroed once per backward, the finalize pass leaves it zero again
    t.G = b.normscr; t.R = b.normscr + (size_t)b.B * 64; t.C = r.cout; t.batch = b.B; t.pix_per_sample = b.pix(lvl);
    t.dgp = b.part_main;
    b.ok(launch_norm_bwd(t, b.writes(L.t1, L.t2)));
    // 2. conv2: y2 = conv(SiLU(GN1(y1)*(1+s)+sh)); the weight gradients that only need the tail's outputs go to the side stream now
    const float* ssrow = r.has_mlp ? b.ss + (size_t)m->ss_layers[r.ss_index].out_off * b.B : nullptr;
    wgrad(b, b.slot(r.s_y1), r.cout, nullptr, 0, L.t1, r.cout, r.b2_w, r.b2_b, lvl, 0, 3, 1, b.stat(r.st1), b.p + r.b1_gs, b.p + r.b1_gb, ssrow, 2 * r.cout, d16);
    if (r.has_res) wgrad(b, x0, c0, x1, c1, L.t2, r.cout, r.rc_w, r.rc_b, lvl, 0, 1, 1);
    dgrad(b, L.t1, r.cout, b.pt + r.pt_b2, r.cout, 0, r.cout, lvl, 0, 3, 1, nullptr, L.t3, d16);       // dL/d(act1)
    // 3. prologue: act1 = SiLU((GN1(y1))*(1+s)+sh)
    NormBwdArgs q;
    memset(&q, 0, sizeof(q));
    q.dact = L.t3; q.y = b.slot(r.s_y1); q.y_bf16 = (m->mode == MODE_BF16); q.dy = L.t4; q.dy_bf16 = d16; q.stats = b.stat(r.st1); q.gamma = b.p + r.b1_gs; q.beta = b.p + r.b1_gb; q.groups = G;
    q.ss = ssrow; q.ss_stride = 2 * r.cout; q.d_gamma = b.grads + r.b1_gs; q.d_beta = b.grads + r.b1_gb;
    q.dss = r.has_mlp ? b.dss + (size_t)m->ss_layers[r.ss_index].out_off * b.B : nullptr;
    q.G = b.normscr; q.R = b.normscr + (size_t)b.B * 64; q.C = r.cout; q.batch = b.B; q.pix_per_sample = b.pix(lvl);
    q.dgp = b.part_main;
    b.ok(launch_norm_bwd(q, b.writes(L.t4)));                                                           // t4 = dL/d(y1) (not t1: conv2's weight gradient may still be reading it)
    // 4. conv1 + residual branch
    wgrad(b, x0, c0, x1, c1, L.t4, r.cout, r.b1_w, r.b1_b, lvl, 0, 3, 1, nullptr, nullptr, nullptr, nullptr, 0, d16);
    const int cin = c0 + c1;
    if (r.has_res) {
        dgrad(b, L.t2, r.cout, b.pt + r.pt_rc, cin, 0, c0, lvl, 0, 1, 1, nullptr, L.t3);
        dgrad(b, L.t4, r.cout, b.pt + r.pt_b1, cin, 0, c0, lvl, 0, 3, 1, L.t3, out0, d16);
        if (c1) {
            dgrad(b, L.t2, r.cout, b.pt + r.pt_rc, cin, c0, c1, lvl, 0, 1, 1, nullptr, L.t3);
            dgrad(b, L.t4, r.cout, b.pt + r.pt_b1, cin, c0, c1, lvl, 0, 3, 1, L.t3, out1, d16);
        }
    } else {
        dgrad(b, L.t4, r.cout, b.pt + r.pt_b1, cin, 0, c0, lvl, 0, 3, 1, L.t2, out0, d16);              // identity residual: + dL/d(r)
    }
}

// 1x1 projection  y[pix][cout] = x[pix][cin] . W (+bias) (+res)  with a packed weight
// x_bf16 / y_bf16: the tensor is a bf16 backward intermediate (bf16 mode); a bf16 y excludes res
void proj(Bwd& b, const float* x, int cin, const void* wp, const float* bias, int cout, int lvl, const float* res, float* y,
          int x_bf16 = 0, int y_bf16 = 0) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x0 = x; a.C0 = cin; a.wp = wp; a.bias = bias; a.y = y; a.Cout = cout; a.res = res; a.x0_bf16 = x_bf16; a.y_bf16 = y_bf16;
    a.NF = b.B * b.m->cfg.num_frames; a.F = b.m->cfg.num_frames; a.H = a.W = b.size(lvl);
    a.kind = 0; a.kh = a.kw = 1; a.stride = 1; a.pad = 0;
    b.ok(launch_conv(b.m->mode, a, b.writes(y)));
}

void wgrad1x1(Bwd& b, const float* x, int cin, const float* dy, int cout, long w_off, long b_off, int lvl, int x_bf16 = 0) {
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x0_bf16 = x_bf16;
    a.x0 = x; a.C0 = cin; a.dy = dy; a.Cout = cout; a.dW = b.grads + w_off; a.db = b_off >= 0 ? b.grads + b_off : nullptr;
    a.NF = b.B * b.m->cfg.num_frames; a.F = b.m->cfg.num_frames; a.H = a.W = b.size(lvl);
    a.kind = 0; a.kh = a.kw = 1; a.stride = 1;
    a.bf16_mma = (b.m->mode == MODE_BF16);
    a.part = b.part_side; a.part_cap = WG_PART_FLOATS;             // (every weight gradient runs on the side stream, in order)
    b.ok(launch_conv_wgrad(a, b.side()));
    b.side_done(b.S, dy);                                          // x = O lives in the attention scratch
}

// the q, k and v projection weight gradients of one block in ONE launch: dy = [rows][dq | dk | dv], x read once
void wgrad1x1_qkv(Bwd& b, const float* x, int cin, const float* dqkv, int hd, const long (&w_off)[3], const long* b_off, int lvl, int dy_bf16 = 0, int x_bf16 = 0) {
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.dy_bf16 = dy_bf16; a.x0_bf16 = x_bf16;
    a.x0 = x; a.C0 = cin; a.dy = dqkv; a.Cout = 3 * hd; a.split = hd;
    a.dW = b.grads + w_off[0]; a.dW1 = b.grads + w_off[1]; a.dW2 = b.grads + w_off[2];
    if (b_off) { a.db = b.grads + b_off[0]; a.db1 = b.grads + b_off[1]; a.db2 = b.grads + b_off[2]; }
    a.NF = b.B * b.m->cfg.num_frames; a.F = b.m->cfg.num_frames; a.H = a.W = b.size(lvl);
    a.kind = 0; a.kh = a.kw = 1; a.stride = 1;
    a.bf16_mma = (b.m->mode == MODE_BF16);
    a.part = b.part_side; a.part_cap = WG_PART_FLOATS;             // (every weight gradient runs on the side stream, in order)
    b.ok(launch_conv_wgrad(a, b.side()));
    b.side_done(b.S);                                              // dq | dk | dv live in the attention scratch
}

// y = MHA(x) + x  backward: g = dL/dy -> out = dL/dx
void attn_bwd(Bwd& b, const AttnP& ap, const float* g, const float* x, int lvl, bool temporal, float* out) {
    const Model* m = b.m;
    const int C = ap.C, H = m->cfg.attn_heads, HD = H * 32;
    const long npix = b.pix(lvl) * b.B;
    float* qkv = b.next_scratch(); float* dO = qkv + npix * 3 * HD; float* O = dO + npix * HD; float* dq = O + npix * HD;      // [npix][dq | dk | dv]
    // bf16 mode, <= 16 tokens: qkv, dO, O and dq|dk|dv are bf16 tensors (the MFMA core is bound by this traffic); the buffers keep
    // their fp32-sized places in the scratch
    const long hw = (long)b.size(lvl) * b.size(lvl), Fr = m->cfg.num_frames;
    const int io16 = (m->mode == MODE_BF16 && (temporal ? Fr : hw) <= 16) ? 1 : 0;
    b.writes(b.S);                                                 // the scratch is rewritten from here on
    if (m->mode == MODE_BF16 && temporal && C == 64 && H == 8 && Fr <= 16) {
        // widest level: one fused kernel (attn_bwd16x_kernel) instead of recompute / dO projection / core / dx projection
        AttnBwdXArgs f;
        memset(&f, 0, sizeof(f));
        f.x = x; f.x_bf16 = b.a16; f.g = g; f.wqkv = b.pk + ap.pk_qkv; f.bqkv = reinterpret_cast<const float*>(b.pk + ap.pk_bqkv); f.woT = b.pt + ap.pt_o;
        f.O = O; f.dqkv = dq; f.dx = out; f.L = (int)Fr; f.nseq = b.B * hw; f.inner = hw; f.outer_p = Fr * hw; f.tok_p = hw;
        f.scale = 1.0f / sqrtf((float)m->cfg.attn_dim_head);
        b.writes(out);
        b.ok(launch_attn_bwd_fused(f, b.st));
        wgrad1x1(b, O, HD, g, C, ap.o_w, ap.o_b, lvl, 1);
        wgrad1x1_qkv(b, x, C, dq, HD, ap.w, ap.b, lvl, 1, b.a16);
        return;
    }
    proj(b, x, C, b.pk + ap.pk_qkv, reinterpret_cast<const float*>(b.pk + ap.pk_bqkv), 3 * HD, lvl, nullptr, qkv, b.a16, io16);
    proj(b, g, C, b.pt + ap.pt_o, nullptr, HD, lvl, nullptr, dO, 0, io16);                           // dO = g . Wo^T
    AttnBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.qkv = qkv; a.dO = dO; a.O = O; a.dq = dq; a.heads = H; a.scale = 1.0f / sqrtf((float)m->cfg.attn_dim_head);
    a.dstride = 3 * HD; a.io_bf16 = io16;                                                            // one [rows][dq|dk|dv] buffer
    if (io16) { a.dk = reinterpret_cast<float*>(reinterpret_cast<char*>(dq) + HD * 2); a.dv = reinterpret_cast<float*>(reinterpret_cast<char*>(dq) + 2 * HD * 2); }
    else { a.dk = dq + HD; a.dv = dq + 2 * HD; }
    if (temporal) { a.L = (int)Fr; a.nseq = b.B * hw; a.inner = hw; a.outer_p = Fr * hw; a.tok_p = hw; }
    else { a.L = (int)hw; a.nseq = b.B * Fr; a.inner = 1; a.outer_p = hw; a.tok_p = 1; }
    a.bf16_mma = (m->mode == MODE_BF16);
    b.ok(launch_attn_core_bwd(a, b.writes(b.S)));
    wgrad1x1(b, O, HD, g, C, ap.o_w, ap.o_b, lvl, io16);
    wgrad1x1_qkv(b, x, C, dq, HD, ap.w, ap.b, lvl, io16, b.a16);
    proj(b, dq, 3 * HD, b.pt + ap.pt_qkv, nullptr, C, lvl, g, out, io16, 0);                          // dx = g + [dq|dk|dv] . [Wq;Wk;Wv]^T
}

void sla_bwd(Bwd& b, const SlaP& sp, const float* g, const float* x, int lvl, float* out) {
    const Model* m = b.m;
    const int C = sp.C, HD = m->cfg.attn_heads * 32;
    const long npix = b.pix(lvl) * b.B;
    float* q = b.next_scratch(); float* k = q + npix * HD; float* v = k + npix * HD; float* dOut = v + npix * HD; float* O = dOut + npix * HD;
    float* dq = O + npix * HD;      // [npix][dq | dk | dv]
    const int io16 = (m->mode == MODE_BF16) ? 1 : 0;      // bf16 mode: q, k, v, dOut, O and dq|dk|dv are bf16 tensors in fp32-sized scratch places
    b.writes(b.S);
    proj(b, x, C, b.pk + sp.pk[0], nullptr, HD, lvl, nullptr, q, b.a16, io16);
    proj(b, x, C, b.pk + sp.pk[1], nullptr, HD, lvl, nullptr, k, b.a16, io16);
    proj(b, x, C, b.pk + sp.pk[2], nullptr, HD, lvl, nullptr, v, b.a16, io16);
    proj(b, g, C, b.pt + sp.pt_o, nullptr, HD, lvl, nullptr, dOut, 0, io16);
    SlaBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.q = q; a.k = k; a.v = v; a.dOut = dOut; a.O = O; a.dq = dq; a.dstride = 3 * HD; a.io_bf16 = io16; a.A = b.sla_a;
    if (io16) { a.dk = reinterpret_cast<float*>(reinterpret_cast<char*>(dq) + HD * 2); a.dv = reinterpret_cast<float*>(reinterpret_cast<char*>(dq) + 2 * HD * 2); }
    else { a.dk = dq + HD; a.dv = dq + 2 * HD; }
    a.NF = b.B * m->cfg.num_frames; a.N = b.size(lvl) * b.size(lvl); a.heads = m->cfg.attn_heads;
    a.bf16_mma = (m->mode == MODE_BF16);
    b.ok(launch_sla_bwd(a, b.writes(b.S)));
    wgrad1x1(b, O, HD, g, C, sp.o_w, -1, lvl, io16);
    wgrad1x1_qkv(b, x, C, dq, HD, sp.w, nullptr, lvl, io16, b.a16);
    proj(b, dq, 3 * HD, b.pt + sp.pt_qkv, nullptr, C, lvl, g, out, io16, 0);
}

float* other(const LevelBufs& L, const float* cur) { return cur == L.ga ? L.gb : L.ga; }

// time-MLP (Linear + LayerNorm) backward of the ResnetBlocks [first, first + count) of Model::ss_layers.  Called at the end of the
// stage that owns the blocks, so every parameter gradient of a stage is final when the stage returns (the data-parallel reducer
// all-reduces a bucket as soon as its last stage has been enqueued); d(temb) accumulates across stages and is consumed by the stem.
void ss_bwd(Bwd& b, int first, int count) {
    if (first < 0 || count <= 0) return;
    // leaves of the reverse graph (parameter gradients and d(temb), which only the stem's time-MLP backward reads -- on the same stream):
    // side stream, in order behind the weight gradients of the stage (main stream: 3 launches of 2-4 workgroups per stage, ~0.4 ms per step)
    b.ok(launch_resblock_ss_bwd(b.p, b.grads, b.temb, b.m->d_ss_layers + first, count, b.ss_lin, b.dss, b.dtemb, b.m->temb_dim, b.B, b.side(), b.part_side, WG_PART_FLOATS));
    b.side_done(b.dss);
}

}  // namespace

void bwd_state_free(BwdState* s) {
    if (s->side) { (void)hipStreamSynchronize(s->side); (void)hipStreamDestroy(s->side); s->side = nullptr; }
    for (int i = 0; i < BWD_EVENTS; ++i) if (s->ev[i]) { (void)hipEventDestroy(s->ev[i]); s->ev[i] = nullptr; }
}

static size_t al(size_t floats) { return (floats + 63) / 64 * 64; }

// norm backward scratch: G [B][64] + the largest per-workgroup partials block of any level (norm_bwd.hip)
static size_t norm_scratch_floats(const Model* m, int B) {
    size_t mx = 0;
    for (int l = 0; l < m->cfg.n_mults; ++l) {
        const long s = m->cfg.image_size >> l, pix = (long)m->cfg.num_frames * s * s;
        const int w0 = m->cfg.dim * m->cfg.dim_mults[l], w1 = l ? m->cfg.dim * m->cfg.dim_mults[l - 1] : m->init_dim;
        for (int C : {w0, w1, m->cfg.dim}) mx = std::max(mx, norm_bwd_scratch_floats(C, B, pix));
    }
    return (size_t)B * 64 + mx;
}

size_t model_bwd_workspace_bytes(const Model* m, int B) {
    const int nl = m->cfg.n_mults;
    size_t fl = 0;
    for (int l = 0; l < nl; ++l) {
        const long s = m->cfg.image_size >> l;
        fl += LVL_BUFS * al((size_t)B * m->cfg.num_frames * s * s * lvl_width(m, l));
    }
    const size_t pix0 = (size_t)B * m->cfg.num_frames * m->cfg.image_size * m->cfg.image_size;
    fl += al(pix0 * m->init_dim);                                  // gr
    fl += 2 * al(pix0 * (size_t)(m->cfg.attn_heads * 32) * 8);     // S (two, alternating per attention block)
    fl += al((size_t)m->ss_floats_per_sample * B);                 // dss
    fl += al((size_t)m->temb_dim * B);                             // dtemb
    fl += al(norm_scratch_floats(m, B));                           // norm scratch
    fl += al(sla_bwd_scratch_floats(B * m->cfg.num_frames, m->cfg.attn_heads));
    fl += 2 * al(WG_PART_FLOATS);                                   // deterministic accumulation slots: side stream, main stream
    return fl * 4;
}

hipError_t model_pack_t(const Model* m, const float* p, void* packed_t, hipStream_t st) {
    if (!m->d_pack_t_jobs) return hipErrorInvalidValue;
    return launch_pack_jobs(m->mode, p, packed_t, m->d_pack_t_jobs, (int)m->pack_t_jobs.size(), st);
}

int model_backward(const Model* m, BwdState* state, const float* params, const void* packed, const void* packed_t, const float* x,
                   const int* time, const float* cond, const unsigned char* cond_mask, int null_all, const float* d_out,
                   void* fwd_workspace, void* bwd_workspace, size_t bwd_workspace_bytes, float* grads, int stage_hi, int stage_lo,
                   int B, hipStream_t st) {
    const vdx_config& c = m->cfg;
    const int nl = c.n_mults, top = 2 * nl + 2;
    if (stage_hi > top || stage_lo < 0 || stage_lo > stage_hi) return vdx_set_error(VDX_ERR_INVALID, "backward: bad stage range", __FILE__, __LINE__);
    if (bwd_workspace_bytes < model_bwd_workspace_bytes(m, B)) return vdx_set_error(VDX_ERR_NOMEM, "backward: workspace too small", __FILE__, __LINE__);
    {
        const long sb = c.image_size >> (c.n_mults - 1);
        if (sb * sb > 64 || c.num_frames > 64)
            return vdx_set_error(VDX_ERR_INVALID, "backward: attention over more than 64 tokens (frames larger than 64 x 64, or more than 64 frames) is forward-only", __FILE__, __LINE__);
    }
    if (stage_hi != top && state->next_stage != stage_hi) return vdx_set_error(VDX_ERR_STATE, "backward: stages must be run in descending order from the head", __FILE__, __LINE__);
    Bwd b;
    b.m = m; b.p = params; b.pk = reinterpret_cast<const char*>(packed); b.pt = reinterpret_cast<const char*>(packed_t);
    b.grads = grads; b.B = B; b.st = st; b.state = state;
    b.a16 = (m->act16 == 2 && m->mode == MODE_BF16) ? 1 : 0;
    {   // forward workspace carve (must match model_forward)
        char* w = reinterpret_cast<char*>(fwd_workspace);
        b.act = reinterpret_cast<float*>(w); w += ((size_t)m->act_floats_per_sample * B * 4 + 255) / 256 * 256;
        b.temb = reinterpret_cast<float*>(w); w += ((size_t)m->temb_dim * B * 4 + 255) / 256 * 256;
        b.ss = reinterpret_cast<float*>(w); w += ((size_t)m->ss_floats_per_sample * B * 4 + 255) / 256 * 256;
        b.ss_lin = reinterpret_cast<float*>(w); w += ((size_t)m->ss_floats_per_sample * B * 4 + 255) / 256 * 256;
        b.stats = reinterpret_cast<double*>(w);
    }
    {   // backward workspace carve
        float* w = reinterpret_cast<float*>(bwd_workspace);
        b.lv.resize(nl);
        for (int l = 0; l < nl; ++l) {
            const long s = c.image_size >> l;
            const size_t n = al((size_t)B * c.num_frames * s * s * lvl_width(m, l));
            b.lv[l].ga = w; w += n; b.lv[l].gb = w; w += n; b.lv[l].t1 = w; w += n; b.lv[l].t2 = w; w += n; b.lv[l].t3 = w; w += n; b.lv[l].t4 = w; w += n; b.lv[l].gskip = w; w += n;
        }
        const size_t pix0 = (size_t)B * c.num_frames * c.image_size * c.image_size;
        b.gr = w; w += al(pix0 * m->init_dim);
        for (int i = 0; i < 2; ++i) { b.Sbuf[i] = w; w += al(pix0 * (size_t)(c.attn_heads * 32) * 8); }
        b.S = b.Sbuf[0];
        b.dss = w; w += al((size_t)m->ss_floats_per_sample * B);
        b.dtemb = w; w += al((size_t)m->temb_dim * B);
        b.normscr = w; w += al(norm_scratch_floats(m, B));
        b.sla_a = w; w += al(sla_bwd_scratch_floats(B * c.num_frames, c.attn_heads));
        b.part_side = w; w += al(WG_PART_FLOATS);
        b.part_main = w;
    }
    hipError_t e;
    if (!state->side) {
        if (hipStreamCreateWithFlags(&state->side, hipStreamNonBlocking) != hipSuccess) return vdx_set_error(VDX_ERR_HIP, "backward: side stream", __FILE__, __LINE__);
        for (int i = 0; i < BWD_EVENTS; ++i)
            if (hipEventCreateWithFlags(&state->ev[i], hipEventDisableTiming) != hipSuccess) return vdx_set_error(VDX_ERR_HIP, "backward: events", __FILE__, __LINE__);
    }
#define VDX_E(x) do { e = (x); if (e != hipSuccess) return vdx_set_error(VDX_ERR_HIP, hipGetErrorString(e), __FILE__, __LINE__); } while (0)
    const long pix0 = b.pix(0) * B;
    float* g = state->g;
    for (int stage = stage_hi; stage >= stage_lo; --stage) {
        if (stage == top) {
            // fresh pass: zero every accumulator
            VDX_E(hipMemsetAsync(grads, 0, (size_t)m->param_total * 4, st));
            VDX_E(hipMemsetAsync(b.dss, 0, (size_t)m->ss_floats_per_sample * B * 4, st));
            VDX_E(hipMemsetAsync(b.dtemb, 0, (size_t)m->temb_dim * B * 4, st));
            // head: out = conv1x1(fin(concat(x_up, r)))
            const Level& U = m->ups[nl - 1];
            VDX_E(launch_final_conv_bwd(b.slot(m->fin.s_out), d_out, params + m->fin_w, b.lv[0].ga, grads + m->fin_w, grads + m->fin_b, pix0, c.dim, m->out_dim, b.a16, b.writes(b.lv[0].ga), b.part_main, WG_PART_FLOATS));
            res_bwd(b, m->fin, b.lv[0].ga, b.slot(U.s_attn), U.cout, b.slot(m->s_init_attn), m->init_dim, 0, b.lv[0].gb, b.gr);
            g = b.lv[0].gb;
        } else if (stage >= nl + 2) {
            const int i = stage - (nl + 2);
            const Level& L = m->ups[i];
            const Level& D = m->downs[nl - 1 - i];
            const int lvl = L.lvl;
            LevelBufs& LB = b.lv[lvl];
            if (L.has_resample) {                                  // g is at level lvl-1: Upsample backward
                wgrad(b, b.slot(L.s_attn), L.cout, nullptr, 0, g, L.cout, L.rs_w, L.rs_b, lvl, 1, 4, 1);
                dgrad(b, g, L.cout, b.pt + L.pt_rs, L.cout, 0, L.cout, lvl, 1, 4, 1, nullptr, LB.ga);
                g = LB.ga;
            }
            const float* attn_in = L.has_sla ? b.slot(L.s_sla) : b.slot(L.res1.s_out);
            float* o = other(LB, g);
            attn_bwd(b, L.attn, g, attn_in, lvl, true, o); g = o;
            if (L.has_sla) { o = other(LB, g); sla_bwd(b, L.sla, g, b.slot(L.res1.s_out), lvl, o); g = o; }
            o = other(LB, g);
            res_bwd(b, L.res1, g, b.slot(L.res0.s_out), L.cout, nullptr, 0, lvl, o, nullptr); g = o;
            const float* xin = (i == 0) ? b.slot(m->mid2.s_out) : b.slot(m->ups[i - 1].s_rs);
            const int cx = (i == 0) ? m->mid2.cout : m->ups[i - 1].cout;
            o = other(LB, g);
            res_bwd(b, L.res0, g, xin, cx, b.slot(D.s_attn), D.cout, lvl, o, LB.gskip); g = o;
            ss_bwd(b, L.res0.ss_index, 2);
        } else if (stage == nl + 1) {
            const int lvl = nl - 1;
            LevelBufs& LB = b.lv[lvl];
            float* o = other(LB, g);
            res_bwd(b, m->mid2, g, b.slot(m->s_mid_tattn), m->mid2.cin, nullptr, 0, lvl, o, nullptr); g = o;
            o = other(LB, g); attn_bwd(b, m->mid_tattn, g, b.slot(m->s_mid_sattn), lvl, true, o); g = o;
            o = other(LB, g); attn_bwd(b, m->mid_sattn, g, b.slot(m->mid1.s_out), lvl, false, o); g = o;
            const Level& D = m->downs[nl - 1];
            o = other(LB, g);
            res_bwd(b, m->mid1, g, b.slot(D.s_attn), D.cout, nullptr, 0, lvl, o, nullptr); g = o;
            ss_bwd(b, m->mid1.ss_index, 2);
        } else if (stage >= 1) {
            const int i = stage - 1;
            const Level& L = m->downs[i];
            LevelBufs& LB = b.lv[i];
            if (L.has_resample) {                                  // g is at level i+1: Downsample backward
                wgrad(b, b.slot(L.s_attn), L.cout, nullptr, 0, g, L.cout, L.rs_w, L.rs_b, i, 0, 4, 2);
                dgrad(b, g, L.cout, b.pt + L.pt_rs, L.cout, 0, L.cout, i, 0, 4, 2, LB.gskip, LB.ga);      // + skip gradient
                g = LB.ga;
            } else {
                VDX_E(launch_add_inplace(g, LB.gskip, b.pix(i) * B * L.cout, b.writes(g)));
            }
            const float* attn_in = L.has_sla ? b.slot(L.s_sla) : b.slot(L.res1.s_out);
            float* o = other(LB, g);
            attn_bwd(b, L.attn, g, attn_in, i, true, o); g = o;
            if (L.has_sla) { o = other(LB, g); sla_bwd(b, L.sla, g, b.slot(L.res1.s_out), i, o); g = o; }
            o = other(LB, g);
            res_bwd(b, L.res1, g, b.slot(L.res0.s_out), L.cout, nullptr, 0, i, o, nullptr); g = o;
            const float* xin = (i == 0) ? b.slot(m->s_init_attn) : b.slot(m->downs[i - 1].s_rs);
            o = other(LB, g);
            res_bwd(b, L.res0, g, xin, L.cin, nullptr, 0, i, o, nullptr); g = o;
            ss_bwd(b, L.res0.ss_index, 2);
        } else {
            // stem: r gradient joins, init temporal attention, init conv, time-embedding MLPs
            LevelBufs& LB = b.lv[0];
            VDX_E(launch_add_inplace(g, b.gr, pix0 * m->init_dim, b.writes(g)));
            // d(temb) is final (every ss_bwd ran on the side stream): the time-MLP backward (a few workgroups, ~240 us) goes there too,
            // beside the init attention's backward; the init conv's weight gradient runs on the main stream behind that attention
            // while the side stream takes the attention's weight gradients (the pass used to end with ~380 us of one stream idle)
            TimeMlpArgs t;
            memset(&t, 0, sizeof(t));
            t.time = time; t.w1 = params + m->t_w1; t.b1 = params + m->t_b1; t.w2 = params + m->t_w2; t.b2 = params + m->t_b2;
            t.dim = c.dim; t.time_dim = m->time_dim; t.cond = cond; t.cond_mask = cond_mask; t.null_all = null_all; t.cond_dim = c.cond_dim; t.temb_dim = m->temb_dim;
            VDX_E(launch_time_mlp_bwd(t, b.dtemb, grads + m->t_w1, grads + m->t_b1, grads + m->t_w2, grads + m->t_b2,
                                      c.cond_dim ? grads + m->null_cond : nullptr, B, b.side()));
            b.side_done(b.dtemb);
            float* o = other(LB, g);
            attn_bwd(b, m->init_attn, g, b.slot(m->s_init), 0, true, o); g = o;
            VDX_E(launch_init_conv_wgrad(x, g, grads + m->init_w, grads + m->init_b, B, c.channels, c.num_frames, c.image_size, c.image_size,
                                         m->init_dim, c.init_kernel_size, b.writes(nullptr), b.part_main, WG_PART_FLOATS));
        }
        if (b.err != hipSuccess) return vdx_set_error(VDX_ERR_HIP, hipGetErrorString(b.err), __FILE__, __LINE__);
    }
    b.join();                                                      // every gradient of the stages of this call is final on `st`
    if (b.err != hipSuccess) return vdx_set_error(VDX_ERR_HIP, hipGetErrorString(b.err), __FILE__, __LINE__);
#undef VDX_E
    state->g = g;
    state->next_stage = stage_lo - 1;
    return VDX_OK;
}

}  // namespace vdx
