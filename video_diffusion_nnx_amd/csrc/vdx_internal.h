// Internal (non-ABI) declarations shared by the kernel translation units and the host runtime.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <algorithm>
#include "../../include/vdx.h"

int vdx_set_error(int code, const char* msg, const char* file, int line);

namespace vdx {

// Instrumentation hook (vdx.h: vdx_set_launch_hook).  A LaunchScope brackets ONE kernel launch: its constructor calls the hook with
// phase 0, its destructor with phase 1.  Costs one load + branch when no hook is installed.
extern vdx_launch_hook g_launch_hook;
extern void* g_launch_hook_user;
struct LaunchScope {
    vdx_launch_info info; char desc[192]; hipStream_t st; bool on;
    LaunchScope(hipStream_t s, const char* kernel, double flops, double bytes, const char* fmt, ...) __attribute__((format(printf, 6, 7))) : st(s), on(g_launch_hook != nullptr) {
        if (!on) return;
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(desc, sizeof(desc), fmt, ap);
        va_end(ap);
        info.kernel = kernel; info.shape = desc; info.flops = flops; info.bytes = bytes;
        g_launch_hook(g_launch_hook_user, 0, &info, st);
    }
    ~LaunchScope() { if (on && g_launch_hook) g_launch_hook(g_launch_hook_user, 1, &info, st); }
    LaunchScope(const LaunchScope&) = delete;
    LaunchScope& operator=(const LaunchScope&) = delete;
};

// Arguments of conv_igemm_kernel (POD, passed by value).  Caller fills the first block; launch_conv
// completes the geometry.
struct ConvArgs {
    // tensors (channel-last fp32): input = concat(x0[.., C0], x1[.., C1]) on the channel axis
    const float* x0; const float* x1; int C0, C1;
    const void* wp;                 // packed weights [taps][Cout][CinPad] (fp32 or bf16 per mode)
    const float* bias;              // [Cout] or null
    float* y; int Cout;
    int NF, F;                      // frames total (B*F), frames per sample
    int H, W;                       // input spatial size
    int kind;                       // 0: conv (kh, kw, stride, pad)   1: ConvTranspose 4x4 / stride 2 (4 phases)
    int kh, kw, stride, pad;
    // prologue on the input: 0 none, 1 GroupNorm-apply (+scale/shift) + SiLU
    int pro;
    const double* in_stats; const float* gamma; const float* beta; int groups;
    const float* ss; int ss_stride; // per-sample [scale(Cin) | shift(Cin)] rows, or null
    // epilogue: GroupNorm partial statistics of the output (or null)
    double* out_stats; int out_groups;
    int x0_bf16, y_bf16;            // x0 / y are stored as bf16 (the intra-ResnetBlock tensors in bf16 mode); y_bf16 excludes res
    int x1_bf16;                    // x1 stored as bf16 (bf16 activation storage)
    const float* res;               // optional residual added to the output: y = conv + bias + res  ([.., Cout] like y)
    int res_bf16;                   // res stored as bf16
    int wrows, wrow0;               // packed weight rows per tap / first row (0,0 = Cout rows from 0): slices a wider packing
    // completed by launch_conv
    int Ho, Wo, Hy, Wy, CinPad, PH, PW, NP, tiles_y, tiles_x;
    unsigned x0_bytes, x1_bytes, w_bytes;          // buffer-descriptor extents
    unsigned m_ihiw, m_iw, m_phpw, m_pw;           // floor(2^32 / d) + 1 for d = IH*IW, IW, PH*PW, PW (div_magic)
};

// out = SiLU(GroupNorm(y2; stats, gn_gamma, gn_beta)) + LayerNorm_C(r; ln_gamma, ln_beta), channel-last [B][pix][C]
struct TailArgs {
    const float* y2; const float* r; float* out;
    int y2_bf16;                    // y2 stored as bf16
    int r_bf16, out_bf16;           // r / out stored as bf16 (bf16 activation storage)
    const double* stats; const float* gn_gamma; const float* gn_beta; int groups;
    const float* ln_gamma; const float* ln_beta;
    int C; int batch; long pix_per_sample;
    int lpp;                        // completed by the launcher
    // fused 1x1 res_conv (bf16 tensors, bf16 mode): r = concat(x0[.., C0], x1[.., C1]) . rc_w + rc_b is computed on the MFMA inside
    // the tail instead of being read from `r` (resblock_tail_rc16_kernel; `r` is ignored when rc_w is set)
    const float* x0; const float* x1; int C0, C1;
    const void* rc_w;               // packed [C rows][C0 + C1] bf16 (conv_packed_bytes(MODE_BF16, 1, C0 + C1, C)), or null
    const float* rc_b;              // [C]
    // fused head (sampling forward, the network's last block): fin_out[pix][0] = out[pix][:] . fin_w[:][0] + fin_b[0] is written INSTEAD of
    // `out` (final_conv of unet3d.py:251 with one output channel; `out` may be null then)
    const float* fin_w; const float* fin_b; float* fin_out;
};
// shapes resblock_tail_rc16_kernel is instantiated for
bool tail_rc16_supported(int cin, int c0, int cout, long pix_per_sample);

struct TimeMlpArgs {
    const int* time; int t_is_device_scalar;     // time[b] (or time[0] for every sample)
    const float* w1; const float* b1; const float* w2; const float* b2;   // Flax [in][out]
    int dim, time_dim;
    const float* cond; const float* null_cond_emb; const unsigned char* cond_mask; int null_all; int cond_dim;
    float* temb; int temb_dim;      // temb_dim = time_dim + cond_dim
};

// one ResnetBlock time-MLP: offsets (in floats) into the flat parameter buffer / the scale-shift workspace
struct SsLayer { long w_off, b_off, g_off, be_off, out_off; int n; int pad_; };

hipError_t launch_resblock_tail(TailArgs a, hipStream_t st);
hipError_t launch_init_conv(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int F, int H, int W,
                            int Cout, int K, int y_bf16, hipStream_t st);
hipError_t launch_init_conv_mode(int mode, const float* x, const float* w, const float* bias, float* y, int B, int Cin, int F, int H, int W,
                                 int Cout, int K, int y_bf16, hipStream_t st);     // bf16 mode: MFMA im2col kernel when Cin == 1
hipError_t launch_final_conv(const float* x, const float* w, const float* bias, float* y, long npix, int D, int Cout, int x_bf16, hipStream_t st);
hipError_t launch_time_mlp(const TimeMlpArgs& a, int B, hipStream_t st);
hipError_t launch_resblock_ss(const float* params, const float* temb, const SsLayer* layers, int nlayers, float* ss_base,
                              float* lin_base, int temb_dim, int B, int max_n, hipStream_t st);   // max_n = widest layer (2 * cout)

// y = MHA(x) + x over sequences of L tokens; token address = (s / inner) * outer_stride + (s % inner) * inner_stride + tok * tok_stride
struct AttnArgs {
    const float* x; float* y;
    const void* wqkv; const float* bqkv;      // packed [3*heads*32][CPad]; bias [3][heads][32]
    const void* wo; const float* bo;          // packed [C][HDPad]; bias [C]
    int C, heads, L;
    long nseq, inner, inner_stride, outer_stride, tok_stride;
    float scale;
    int io_bf16;                              // x and y stored as bf16 (bf16 activation storage); strides stay in elements
    int fp8_core;                             // bf16 mode, <= 16 tokens: QK^T and PV on fp8 (e4m3) MFMA operands (vdx_set_attention_fp8)
    void* oscratch;                           // [rows][heads*32] bf16: per-head attention output of launch_attention_heads
    int CPad, HDPad;                          // completed by the launcher
};
hipError_t launch_attention(int mode, AttnArgs a, hipStream_t st);
// bf16 mode, <= 16 tokens, 8 heads: q/k/v projection + attention core per head (one workgroup per head: its weights resident in LDS,
// every wave streams its own sequences, no barriers) -> a.oscratch; the out-projection + residual is a 1x1 conv_igemm by the caller
hipError_t launch_attention_heads(AttnArgs a, hipStream_t st);
// sequences of more than 64 tokens (token-major rows): softmax(q k^T / sqrt(d)) v per (sequence, head) from a materialised
// qkv [rows][3 * heads * 32] fp32 into o [rows][heads * 32] fp32; the projections are 1x1 convs by the caller
hipError_t launch_attention_long_core(const float* qkv, float* o, long nseq, int L, int heads, float scale, hipStream_t st);

// y = SpatialLinearAttention(x) + x, x/y channel-last [NF][N][C]; 8 heads x 32
struct SlaArgs {
    const float* x; float* y;
    const void* wq; const void* wk; const void* wv;   // packed [256][CPad] each
    const void* wo;                                    // packed [C][256]
    void* workspace;                                   // sla_workspace_bytes()
    int C, heads, NF, N;
    int io_bf16;                                       // x and y stored as bf16 (bf16 activation storage)
    // completed by the launcher
    int CPad, nsub, nchunk; float* part; void* ctxT;
};
size_t sla_workspace_bytes(int mode, int NF, int N, int heads);
hipError_t launch_sla(int mode, SlaArgs a, hipStream_t st);
// bf16 mode, 8 heads, N % 16 == 0: per-head kernel (weights resident in LDS, one wave per frame) -> O [NF * N rows][heads * 32] bf16;
// to_out + residual is a 1x1 conv_igemm by the caller
hipError_t launch_sla_heads(SlaArgs a, void* O, hipStream_t st);

struct PSampleArgs {
    const float* x; const float* eps; float* out;       // x/out [B,C,F,H,W] (may alias); eps channel-last [B,F,H,W,C]
    const int* t; const float* tables; int T;           // device t[B]; tables [5][T]
    const float* noise;                                 // explicit z [B,C,F,H,W], or null -> Philox(seed, offset + *dev_offset)
    unsigned long long seed, offset; const unsigned long long* dev_offset;
    const float* thres;                                 // per-sample dynamic threshold s[B] or null (s = 1)
    int clip; int C; long per_sample;
    float post_scale, post_shift;
};
hipError_t launch_randn(float* out, long n, unsigned long long seed, unsigned long long offset, const unsigned long long* dev_offset, hipStream_t st);
hipError_t launch_q_sample(const float* x0, const int* t, const float* noise, float* out, const float* sqrt_ac, const float* sqrt_1mac,
                           int B, long per_sample, float pre_scale, float pre_shift, hipStream_t st);
hipError_t launch_p_sample(const PSampleArgs& a, int B, hipStream_t st);
hipError_t launch_advance(int* t, int B, unsigned long long* dev_offset, hipStream_t st);
hipError_t launch_ddim_step(const float* x, const float* eps, float* out, const float* ac, const int* seq, const unsigned long long* step_dev,
                            const float* thres, int clip, int B, int C, long per_sample, hipStream_t st);
hipError_t launch_ddim_advance(int* t, int B, const int* seq, unsigned long long* step_dev, hipStream_t st);
hipError_t launch_dyn_thres(const float* x, const float* eps, const int* t, const float* tables, int T, float q, float* out, int B, int C,
                            long per_sample, hipStream_t st);
hipError_t launch_loss(const float* eps_hat, const float* noise, double* acc, int B, int Cc, long fhw, int l2, hipStream_t st);
hipError_t launch_affine(const float* x, float* y, long n, float a, float b, hipStream_t st);

hipError_t launch_loss_grad(const float* eps_hat, const float* noise, float* d_eps, int B, int Cc, long fhw, int l2, hipStream_t st);
hipError_t launch_adam_ema(float* p, const float* g, float* m, float* v, float* ema, long n, float lr, float b1, float b2, float eps,
                           long step_count, float grad_scale, int do_ema, float decay, hipStream_t st);

// Weight gradient of a conv (kind 0: (kh,kw)/stride SAME; kind 1: ConvTranspose 4x4/2): dW += Xhat^T (*) dY
struct WgradArgs {
    const float* x0; const float* x1; int C0, C1;          // conv input (channel-last), optional concat
    const float* dy; int Cout;                             // output gradient [NF, Hy, Wy, Cout]
    float* dW;                                             // Flax layout [taps][Cin][Cout], accumulated with atomics
    float* db;                                             // optional: db[co] += sum_pixels dY (bias gradient), or null
    int NF, F, H, W;                                       // input geometry
    int kind, kh, kw, stride;
    int pro; const double* in_stats; const float* gamma; const float* beta; int groups; const float* ss; int ss_stride;
    int x0_bf16;                                           // x0 (and x1) stored as bf16
    int dy_bf16;                                           // dy stored as bf16 (backward intermediates of bf16 mode)
    int split; float* dW1; float* dW2; float* db1; float* db2;   // split > 0: dY column block co / split goes to (dW, dW1, dW2)[co / split], each [taps][Cin][split]
    int bf16_mma;                                          // bf16 MFMA operands (bf16 mode) instead of exact f32
    // deterministic accumulation (round 3): when `part` is set and large enough, every workgroup STORES its partial tile into its own slot
    // part[slot][taps][Cin][Cout] (bias sums: behind them, [slot][Cout]) and a finalize pass adds the slots in a fixed order -- no atomics,
    // so two runs give bit-identical gradients.  Without scratch (the block-level entry points) the kernels add with float atomics.
    float* part; size_t part_cap;                          // scratch and its capacity in floats, or null / 0
    float* part_b; long part_E;                            // completed by the launcher: bias slots, floats per dW slot
    // completed by the launcher
    int taps, sa, sb, ext, halo, Hm, Wm, Hy, Wy, PH, PW, co_tiles;
    unsigned m_iw, m_bw;                                   // magic multipliers (div_magic) of the staged window widths
    int pwl;                                               // log2(PW) (bf16 form)
};
hipError_t launch_conv_wgrad(WgradArgs a, hipStream_t st);
struct PackJob;
hipError_t launch_pack_jobs(int mode, const float* params, void* dst_base, const PackJob* d_jobs, int njobs, hipStream_t st);
hipError_t launch_colsum(const float* x, float* out, long rows, int C, hipStream_t st, float* part = nullptr, size_t part_cap = 0);
// dst[e] (+= the split targets: column block co / split of a [rows][Cout] layout goes to d0 / d1 / d2) += sum over slots k < nslots, in
// order, of part[k * slot_stride + e], e < E: the second half of every deterministic accumulation of the backward
hipError_t launch_slot_sum(const float* part, int nslots, size_t slot_stride, long E, int Cout, int split, float* d0, float* d1, float* d2, hipStream_t st);
constexpr size_t WG_PART_FLOATS = (size_t)12 << 20;        // scratch per stream for the slots (48 MB; a launch that needs more falls back to atomics)
hipError_t launch_add_inplace(float* y, const float* x, long n, hipStream_t st);

// Backward of  act = SiLU((gamma*GN(y)+beta)*(1+s)+sh)  [+ LayerNorm_C(r) branch of the block tail]; channel-last [B][pix][C]
struct NormBwdArgs {
    const float* dact; const float* y; float* dy;                 // dL/dact (or dL/dout), saved pre-norm tensor, result dL/dy
    int y_bf16;                                                   // y stored as bf16
    int dy_bf16;                                                  // dy written as bf16 (its consumers, the conv data / weight gradients, round it to bf16 anyway)
    int r_bf16;                                                   // r stored as bf16 (bf16 activation storage of the training forward)
    const double* stats; const float* gamma; const float* beta; int groups;
    const float* ss; int ss_stride;                               // forward scale/shift rows or null
    float* d_gamma; float* d_beta;                                // accumulated (atomics)
    float* dss;                                                   // [B][2C] (ds | dsh) written, or null
    const float* r; const float* ln_gamma; float* dr; float* d_ln_gamma; float* d_ln_beta;   // LN branch (tail) or nulls
    float* R; float* G;                                           // scratch: per-workgroup partial sums [B][nwg][4][C] (written, never accumulated), [B][groups][2]
    int nwg;                                                      // workgroups per sample of the reduce pass (completed by the launcher)
    float* dgp;                                                   // deterministic mode: scratch [batch][4][C] for the per-sample parameter-gradient rows (or null: float atomics)
    int C, batch; long pix_per_sample;
    int lpp;
};
hipError_t launch_norm_bwd(NormBwdArgs a, hipStream_t st);
size_t norm_bwd_scratch_floats(int C, int batch, long pix_per_sample);     // floats behind NormBwdArgs::R + G (G = R + that - 64 * batch)

// attention core backward: qkv [npix][3*heads*32] (biased, q unscaled), dO [npix][heads*32] -> O, dq, dk, dv [npix][heads*32]
struct AttnBwdArgs {
    const float* qkv; const float* dO; float* O; float* dq; float* dk; float* dv;
    int heads, L; long nseq, inner, outer_p, tok_p; float scale;
    int dstride;                                                     // row stride (elements) of dq / dk / dv: heads*32, or 3*heads*32 for one [rows][dq|dk|dv] buffer
    int io_bf16;                                                     // qkv, dO, O, dq, dk, dv are bf16 tensors (bf16 MFMA form only)
    int bf16_mma;                                                    // bf16 MFMA form (bf16 mode, L <= 16) instead of the exact fp32 VALU form
};
hipError_t launch_attn_core_bwd(const AttnBwdArgs& a, hipStream_t st);
// fused attention backward of the widest level (bf16 mode, C == 64, 8 heads x 32, <= 16 tokens): q/k/v recompute, dO = g Wo^T, the
// core and dx = g + dq|dk|dv . Wqkv^T in one kernel; O and dq|dk|dv are written (bf16) for the weight-gradient kernels
struct AttnBwdXArgs {
    const float* x; const float* g;            // block input, dL/d(block output): fp32 [rows][64]
    int x_bf16;                                // x stored as bf16 (bf16 activation storage of the training forward)
    const void* wqkv; const float* bqkv;       // forward packing [3 * 256][64] bf16, biases [3 * 256]
    const void* woT;                           // transposed packing of the out-projection [256][64] bf16
    void* O; void* dqkv;                       // bf16 [rows][256], [rows][768]
    float* dx;                                 // fp32 [rows][64]
    int L; long nseq, inner, outer_p, tok_p; float scale;
};
hipError_t launch_attn_bwd_fused(const AttnBwdXArgs& a, hipStream_t st);
// SLA core backward: q,k,v,dOut [NF*N][256] -> O (forward, pre to_out), dq, dk, dv ; A = scratch (sla_bwd_scratch_floats)
struct SlaBwdArgs {
    const float* q; const float* k; const float* v; const float* dOut; float* O; float* dq; float* dk; float* dv; float* A;
    int NF, N, heads;
    int dstride;                                  // row stride (elements) of dq / dk / dv: 256, or 768 for one interleaved buffer
    int io_bf16;                                  // q, k, v, dOut, O, dq, dk, dv are bf16 tensors (bf16 MFMA forms only)
    int bf16_mma;                                 // pass A on bf16 MFMA (bf16 mode) instead of the exact fp32 VALU form
};
size_t sla_bwd_scratch_floats(int NF, int heads);
hipError_t launch_sla_bwd(const SlaBwdArgs& a, hipStream_t st);

hipError_t launch_final_conv_bwd(const float* x, const float* dout, const float* w, float* dx, float* dW, float* db, long npix, int D, int Cout, int x_bf16, hipStream_t st, float* part = nullptr, size_t part_cap = 0);
hipError_t launch_init_conv_wgrad(const float* x, const float* dy, float* dW, float* db, int B, int Cin, int F, int H, int W, int Cout, int K, hipStream_t st, float* part = nullptr, size_t part_cap = 0);
hipError_t launch_resblock_ss_bwd(const float* params, float* grads, const float* temb, const SsLayer* layers, int nlayers, const float* lin_base,
                                  float* dss_base, float* dtemb, int temb_dim, int B, hipStream_t st, float* part = nullptr, size_t part_cap = 0);
hipError_t launch_time_mlp_bwd(const TimeMlpArgs& a, const float* dtemb, float* dw1, float* db1, float* dw2, float* db2, float* dnull, int B, hipStream_t st);

size_t conv_packed_bytes(int mode, int taps, int Cin, int Cout);
int conv_cin_pad(int mode, int Cin);
hipError_t launch_pack_weights(int mode, const float* src, void* dst, int taps, int Cin, int Cout, hipStream_t st);
hipError_t launch_conv(int mode, ConvArgs a, hipStream_t st);
int conv3x3_ws_geo(const ConvArgs& a);
bool conv4x4_ws_eligible(int mode, const ConvArgs& a);     // Downsample / Upsample of the wide levels on the weight-streaming machinery
hipError_t launch_conv4x4_ws(const ConvArgs& a, hipStream_t st);       // 0 / 8 / 16: the GEO template argument launch_conv3x3_ws will use
// y <- SiLU(GroupNorm(y) * (scale + 1) + shift) in place on a bf16 tensor [batch][pix][C] (the Block prologue as its own pass; elementwise.hip)
hipError_t launch_gn_silu_apply16(float* y_bf16, const double* stats, const float* gamma, const float* beta, const float* ss, int ss_stride, int groups,
                                  int C, int batch, long pix_per_sample, hipStream_t st);
bool resample32_eligible(int mode, const ConvArgs& a);      // Downsample / Upsample of 32-channel bf16 tensors (dim-32 networks): everything in registers (conv_rs.hip)
hipError_t launch_resample32(const ConvArgs& a, hipStream_t st);
bool conv1x1_pw_eligible(int mode, const ConvArgs& a);     // 1x1 convs of the wide levels on bf16 tensors: weight rows resident in LDS, x straight into fragments (conv_pw.hip)
hipError_t launch_conv1x1_pw(const ConvArgs& a, hipStream_t st);
int conv1x1_pw_rows(const ConvArgs& a);                    // 64 / 128: the ROWS template argument launch_conv1x1_pw will use
// persistent weight-streaming 3x3 kernel for Cout % 128 == 0 with bf16 tensors (conv_ws.hip); `a` geometry-completed by launch_conv
bool conv3x3_ws_eligible(int mode, const ConvArgs& a);
hipError_t launch_conv3x3_ws(const ConvArgs& a, hipStream_t st);
hipError_t launch_pack_weights_t(int mode, const float* src, void* dst, int taps, int Cin, int Cout, hipStream_t st);

}  // namespace vdx
