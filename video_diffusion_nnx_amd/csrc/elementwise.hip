// HBM-bound elementwise / small-reduction kernels of the UNet (gfx950), all fp32, float4-vectorised.
//
//   resblock_tail ...... out = SiLU(GroupNorm(y2)) + LayerNorm_C(r)      modules.py:173-179 (Block 2) + :240-243
//   init_conv .......... nnx.Conv(C -> D, (1,7,7)) SAME, direct form     unet3d.py:110-115,282
//   final_conv ......... nnx.Conv(D -> out_dim, 1) pointwise             unet3d.py:251
//   time_mlp ........... SinusoidalPosEmb -> Linear -> GELU(tanh) -> Linear (+ cond mix)   modules.py:30-45, unet3d.py:128-133,288-298
//   resblock_ss ........ LayerNorm(Linear(SiLU(t)))  for every ResnetBlock in one launch    modules.py:202-208,233-238
#include "vdx_common.h"
#include <stdlib.h>
#include "vdx_internal.h"

namespace vdx {

// ------------------------------------------------------------------------------------------------
// ResnetBlock tail.  One pixel is handled by LPP lanes (LPP a power of two <= 64), each lane VPL float4.
// ------------------------------------------------------------------------------------------------
template <int VPL>
__global__ __launch_bounds__(256) void resblock_tail_kernel(TailArgs P) {
    __shared__ float coefA[1024], coefD[1024];
    __shared__ float gm[64];
    const int tid = threadIdx.x;
    const int C = P.C;
    // a workgroup never spans samples: blockIdx.y = sample
    const int b = blockIdx.y;
    gn_mean_rstd_wg(P.stats, b, P.groups, (double)P.pix_per_sample * (C / P.groups), gm, tid, 256);
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int g = c / (C / P.groups);
        const float a = gm[2 * g + 1] * P.gn_gamma[c];
        coefA[c] = a;
        coefD[c] = P.gn_beta[c] - gm[2 * g] * a;
    }
    __syncthreads();
    const int LPP = P.lpp;                                   // lanes per pixel
    const int ppb = 256 / LPP;                               // pixels per workgroup pass
    const int sub = tid % LPP, pl = tid / LPP;
    const float invC = 1.0f / (float)C;
    for (long pix = (long)blockIdx.x * ppb + pl; pix < P.pix_per_sample; pix += (long)gridDim.x * ppb) {
        const size_t base = ((size_t)b * P.pix_per_sample + pix) * C;
        float4 r[VPL];
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int c = (v * LPP + sub) * 4;
            r[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < C) {
                r[v] = load4_f32_or_bf16(P.r, base + c, P.r_bf16);
                s += r[v].x + r[v].y + r[v].z + r[v].w;
                ss += r[v].x * r[v].x + r[v].y * r[v].y + r[v].z * r[v].z + r[v].w * r[v].w;
            }
        }
        for (int o = 1; o < LPP; o <<= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
        const float mean = s * invC;
        const float var = fmaxf(ss * invC - mean * mean, 0.f);
        const float rstd = rsqrtf(var + NORM_EPS);
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int c = (v * LPP + sub) * 4;
            if (c < C) {
                const float4 y = load4_f32_or_bf16(P.y2, base + c, P.y2_bf16);
                const float4 a = *reinterpret_cast<const float4*>(coefA + c);
                const float4 d = *reinterpret_cast<const float4*>(coefD + c);
                const float4 lg = *reinterpret_cast<const float4*>(P.ln_gamma + c);
                const float4 lb = *reinterpret_cast<const float4*>(P.ln_beta + c);
                float4 o;
                o.x = silu_f(fmaf(y.x, a.x, d.x)) + fmaf((r[v].x - mean) * rstd, lg.x, lb.x);
                o.y = silu_f(fmaf(y.y, a.y, d.y)) + fmaf((r[v].y - mean) * rstd, lg.y, lb.y);
                o.z = silu_f(fmaf(y.z, a.z, d.z)) + fmaf((r[v].z - mean) * rstd, lg.z, lb.z);
                o.w = silu_f(fmaf(y.w, a.w, d.w)) + fmaf((r[v].w - mean) * rstd, lg.w, lb.w);
                store4_f32_or_bf16(P.out, base + c, o, P.out_bf16);
            }
        }
    }
}

// bf16 activation storage: y2, r and out are all bf16 tensors -> one lane handles 8 channels = 16 bytes per tensor
// (half the memory instructions of the 4-channel form; the arithmetic is unchanged, fp32).
__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xFFFF0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xFFFF0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xFFFF0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xFFFF0000u);
}

template <int VPL>
__global__ __launch_bounds__(256) void resblock_tail16_kernel(TailArgs P) {
    __shared__ float coefA[1024], coefD[1024];
    __shared__ float gm[64];
    const int tid = threadIdx.x;
    const int C = P.C;
    const int b = blockIdx.y;
    gn_mean_rstd_wg(P.stats, b, P.groups, (double)P.pix_per_sample * (C / P.groups), gm, tid, 256);
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int g = c / (C / P.groups);
        const float a = gm[2 * g + 1] * P.gn_gamma[c];
        coefA[c] = a;
        coefD[c] = P.gn_beta[c] - gm[2 * g] * a;
    }
    __syncthreads();
    const int LPP = P.lpp;
    const int ppb = 256 / LPP;
    const int sub = tid % LPP, pl = tid / LPP;
    const float invC = 1.0f / (float)C;
    const char* rb = reinterpret_cast<const char*>(P.r);
    const char* yb = reinterpret_cast<const char*>(P.y2);
    char* ob = reinterpret_cast<char*>(P.out);
    for (long pix = (long)blockIdx.x * ppb + pl; pix < P.pix_per_sample; pix += (long)gridDim.x * ppb) {
        const size_t base = ((size_t)b * P.pix_per_sample + pix) * C;
        float r[VPL][8];
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int c = (v * LPP + sub) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) r[v][k] = 0.f;
            if (c < C) {
                unpack8(*reinterpret_cast<const uint4*>(rb + (base + c) * 2), r[v]);
#pragma unroll
                for (int k = 0; k < 8; ++k) { s += r[v][k]; ss += r[v][k] * r[v][k]; }
            }
        }
        for (int o = 1; o < LPP; o <<= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
        const float mean = s * invC;
        const float var = fmaxf(ss * invC - mean * mean, 0.f);
        const float rstd = rsqrtf(var + NORM_EPS);
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int c = (v * LPP + sub) * 8;
            if (c < C) {
                float y[8], o[8];
                unpack8(*reinterpret_cast<const uint4*>(yb + (base + c) * 2), y);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    o[k] = silu_f(fmaf(y[k], coefA[c + k], coefD[c + k])) + fmaf((r[v][k] - mean) * rstd, P.ln_gamma[c + k], P.ln_beta[c + k]);
                *reinterpret_cast<uint4*>(ob + (base + c) * 2) =
                    make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
            }
        }
    }
}

// ---- the Block prologue as its own pass: y <- SiLU(GroupNorm(y) * (scale + 1) + shift), in place on a bf16 tensor -------------------------
// conv3x3_ws_kernel's fused-prologue forms re-do this arithmetic in each of the Cout / 128 workgroups of a pixel range, inside a kernel that is
// bound by instruction issue: at 512 channels x 8 x 8 pixels the prologue costs 57 us per launch (313 vs 256 us at B = 64), the tensor is 67 MB
// -- this pass moves 134 MB in ~25 us.  Used by the sampling forward (inference storage: nothing else reads y1 afterwards) where that pays
// (model.hip run_resblock); identical values: the fused form rounds the activation to bf16 before the MFMA too.
__global__ __launch_bounds__(256) void gn_silu_apply16_kernel(unsigned* __restrict__ y, const double* __restrict__ stats, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ ss, int ss_stride, int groups,
                                                              int C, long pix_per_sample) {
    __shared__ float coefA[1024], coefD[1024];
    __shared__ float gm[64];
    const int tid = threadIdx.x, b = blockIdx.y;
    gn_mean_rstd_wg(stats, b, groups, (double)pix_per_sample * (C / groups), gm, tid, 256);
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int g = c / (C / groups);
        const float m = gm[2 * g], rs = gm[2 * g + 1];
        const float ga = gamma[c], be = beta[c];
        float sc = 1.f, sh = 0.f;
        if (ss) { sc = ss[(size_t)b * ss_stride + c] + 1.f; sh = ss[(size_t)b * ss_stride + C + c]; }
        coefA[c] = rs * ga * sc;                              // (the expressions of conv3x3_ws_kernel's make_coef: bit-identical coefficients)
        coefD[c] = (be - m * rs * ga) * sc + sh;
    }
    __syncthreads();
    const int octs = C >> 3;
    const long n = pix_per_sample * octs;                     // 16-byte pieces of this sample
    uint4* base = reinterpret_cast<uint4*>(y) + (size_t)b * n;
    for (long i0 = (long)blockIdx.x * 1024 + tid; i0 < n; i0 += (long)gridDim.x * 1024) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i0 + u * 256 < n) ? base[i0 + u * 256] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * 256 >= n) continue;
            const int c = (int)((i0 + u * 256) % octs) * 8;
            float f[8], o[8];
            unpack8(v[u], f);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = silu_f(fmaf(f[k], coefA[c + k], coefD[c + k]));
            base[i0 + u * 256] = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
        }
    }
}

hipError_t launch_gn_silu_apply16(float* y_bf16, const double* stats, const float* gamma, const float* beta, const float* ss, int ss_stride, int groups,
                                  int C, int batch, long pix_per_sample, hipStream_t st) {
    if (C % 8 || C > 1024 || groups <= 0 || groups > 32 || C % groups) return hipErrorInvalidValue;
    const long n = pix_per_sample * (C / 8);
    const int gx = (int)std::max<long>(1, std::min<long>((n + 1023) / 1024, 2048));
    LaunchScope ls(st, "gn_silu_apply16_kernel", 8.0 * batch * pix_per_sample * C, 4.0 * batch * pix_per_sample * C, "C%d px%ld x %d", C, pix_per_sample, batch);
    hipLaunchKernelGGL(gn_silu_apply16_kernel, dim3(gx, batch), dim3(256), 0, st, reinterpret_cast<unsigned*>(y_bf16), stats, gamma, beta, ss, ss_stride, groups, C, pix_per_sample);
    return hipGetLastError();
}

// ---- tail with the 1x1 res_conv inside (bf16 tensors, bf16 MFMA operands) ---------------------------------------------------------
// out = SiLU(GroupNorm(y2)) + LayerNorm_C(concat(x0, x1) . W_rc + b_rc)   modules.py:219-222 (res_conv) + :240-243
// The separate 1x1 conv read the block input (2E at level 0, a concat) and wrote r (E) for the tail to read again; here a wave takes
// 16 pixels, fetches their x rows straight into MFMA B fragments (lane (pixel, q) = 16 bytes of a 64-byte K chunk) and multiplies
// with the weight image kept in LDS.  The A-tile rows are PERMUTED so that the accumulators of lane (pixel, q) are the channels
// {32j + 8q .. 32j + 8q + 7}: whole 16-byte pieces of the bf16 rows of y2 / out, i.e. the tail's own access pattern.  r never
// exists in HBM and is not rounded to bf16 before the LayerNorm.  Loads of the next 16-pixel group are in flight during the
// arithmetic of the current one (two register sets, ping-pong).  Addresses are per-sample buffer descriptors + one 32-bit row
// offset per tensor + immediates.  CAT: the input is a concat of two CIN/2-channel tensors.
typedef unsigned tu32x4 __attribute__((ext_vector_type(4)));

// FIN: the network's 1x1 head (one output channel) instead of the store of `out`: the last block's output never exists in HBM (537 MB
// written and read again at the N shape, B = 64) and `final_conv16_kernel` is not launched; the dot product takes the fp32 values.
template <int CIN, int COUT, bool CAT, bool FIN = false>
__global__ __launch_bounds__(256, (CIN == 128 && COUT == 64) ? 4 : (COUT <= 128 ? 3 : 2)) void resblock_tail_rc16_kernel(TailArgs P) {
    using M = Mma<MODE_BF16>;
    constexpr int NTM = COUT / 16, NKS = CIN / 32, NH = COUT / 32;
    constexpr int C0 = CAT ? CIN / 2 : CIN, NK0 = C0 / 32;
    constexpr int WRS = CIN * 2 + 16;                       // LDS row stride of the weight image (bytes)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Wl = smem;                                        // [COUT rows, A-tile order][WRS]
    float* coefA = reinterpret_cast<float*>(Wl + COUT * WRS);
    float* coefD = coefA + COUT;
    float* lng = coefD + COUT;
    float* lnb = lng + COUT;
    float* rcb = lnb + COUT;
    float* gm = rcb + COUT;                                 // [32][mean, rstd]
    float* finw = gm + 64;                                  // [COUT] (FIN)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lp = lane & 15, q = lane >> 4;
    const int b = blockIdx.y;
    for (int i = tid; i < COUT * (CIN / 8); i += 256) {
        const int row = i / (CIN / 8), c = i % (CIN / 8);
        const int tm = row >> 4, rr = row & 15;
        const int co = (tm >> 1) * 32 + (rr >> 2) * 8 + (tm & 1) * 4 + (rr & 3);
        *reinterpret_cast<uint4*>(Wl + row * WRS + c * 16) =
            *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(P.rc_w) + ((size_t)co * ((CIN + 63) / 64 * 64) + c * 8) * 2);      // (packed rows are padded to 64 input channels)
    }
    gn_mean_rstd_wg(P.stats, b, P.groups, (double)P.pix_per_sample * (COUT / P.groups), gm, tid, 256);
    __syncthreads();
    for (int c = tid; c < COUT; c += 256) {
        const int g = c / (COUT / P.groups);
        const float a = gm[2 * g + 1] * P.gn_gamma[c];
        coefA[c] = a;
        coefD[c] = P.gn_beta[c] - gm[2 * g] * a;
        lng[c] = P.ln_gamma[c]; lnb[c] = P.ln_beta[c]; rcb[c] = P.rc_b[c];
        if (FIN) finw[c] = P.fin_w[c];
    }
    __syncthreads();

    const unsigned pps = (unsigned)P.pix_per_sample;
    const char* x0p = reinterpret_cast<const char*>(P.x0) + (size_t)b * pps * C0 * 2;
    const char* x1p = CAT ? reinterpret_cast<const char*>(P.x1) + (size_t)b * pps * C0 * 2 : x0p;
    const char* y2p = reinterpret_cast<const char*>(P.y2) + (size_t)b * pps * COUT * 2;
    char* outp = FIN ? const_cast<char*>(y2p) : reinterpret_cast<char*>(P.out) + (size_t)b * pps * COUT * 2;      // (FIN: no store through `ro`)
    float* const finp = FIN ? P.fin_out + (size_t)b * pps : nullptr;
    const float finb = FIN ? P.fin_b[0] : 0.f;
    const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(x0p), 0, pps * C0 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(x1p), 0, pps * C0 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(y2p), 0, pps * COUT * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(outp, 0, pps * COUT * 2, 0x00020000);
    const int ngroups = (int)(pps >> 4);
    const int gstride = (int)gridDim.x * 4;
    const unsigned lane_x = lp * (C0 * 2) + q * 16, lane_y = lp * (COUT * 2) + q * 16;

    auto issue = [&](int g, tu32x4 (&xf)[NKS], tu32x4 (&yf)[NH]) {
        const unsigned ox = (unsigned)g * (16 * C0 * 2) + lane_x, oy = (unsigned)g * (16 * COUT * 2) + lane_y;
#pragma unroll
        for (int s = 0; s < NKS; ++s)
            xf[s] = (s < NK0) ? __builtin_amdgcn_raw_buffer_load_b128(rx0, ox + s * 64, 0, 0)
                              : __builtin_amdgcn_raw_buffer_load_b128(rx1, ox + (s - NK0) * 64, 0, 0);
#pragma unroll
        for (int j = 0; j < NH; ++j) yf[j] = __builtin_amdgcn_raw_buffer_load_b128(ry, oy + j * 64, 0, 0);
    };
    auto compute = [&](int g, const tu32x4 (&xf)[NKS], const tu32x4 (&yf)[NH]) {
        // the weight fragments and coefficients are re-read from LDS for every group: hoisted out of the loop they take ~150 registers
        // and leave one wave per SIMD, far too few loads in flight for a streaming kernel
        __asm__ volatile("" ::: "memory");
        f32x4 acc[NTM];
#pragma unroll
        for (int tm = 0; tm < NTM; ++tm) {
            const float4 b4 = *reinterpret_cast<const float4*>(rcb + (tm >> 1) * 32 + q * 8 + (tm & 1) * 4);
            acc[tm] = f32x4{b4.x, b4.y, b4.z, b4.w};
        }
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            const uint4 bx = make_uint4(xf[s].x, xf[s].y, xf[s].z, xf[s].w);
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm)
                M::mma(acc[tm], *reinterpret_cast<const uint4*>(Wl + (tm * 16 + lp) * WRS + s * 64 + q * 16), bx);
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1 += acc[tm][e]; s2 += acc[tm][e] * acc[tm][e]; }
        s1 = reduce_q(s1); s2 = reduce_q(s2);
        const float mean = s1 * (1.0f / COUT);
        const float var = fmaxf(s2 * (1.0f / COUT) - mean * mean, 0.f);
        const float rstd = rsqrtf(var + NORM_EPS);
        const unsigned oy = (unsigned)g * (16 * COUT * 2) + lane_y;
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < NH; ++j) {
            float y[8], o[8];
            unpack8(make_uint4(yf[j].x, yf[j].y, yf[j].z, yf[j].w), y);
            const int c0 = j * 32 + q * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 a4 = *reinterpret_cast<const float4*>(coefA + c0 + 4 * h), d4 = *reinterpret_cast<const float4*>(coefD + c0 + 4 * h);
                const float4 g4 = *reinterpret_cast<const float4*>(lng + c0 + 4 * h), e4 = *reinterpret_cast<const float4*>(lnb + c0 + 4 * h);
                const f32x4 r = acc[2 * j + h];
                o[4 * h + 0] = silu_f(fmaf(y[4 * h + 0], a4.x, d4.x)) + fmaf((r[0] - mean) * rstd, g4.x, e4.x);
                o[4 * h + 1] = silu_f(fmaf(y[4 * h + 1], a4.y, d4.y)) + fmaf((r[1] - mean) * rstd, g4.y, e4.y);
                o[4 * h + 2] = silu_f(fmaf(y[4 * h + 2], a4.z, d4.z)) + fmaf((r[2] - mean) * rstd, g4.z, e4.z);
                o[4 * h + 3] = silu_f(fmaf(y[4 * h + 3], a4.w, d4.w)) + fmaf((r[3] - mean) * rstd, g4.w, e4.w);
            }
            if constexpr (FIN) {
                const float4 w0 = *reinterpret_cast<const float4*>(finw + c0), w1 = *reinterpret_cast<const float4*>(finw + c0 + 4);
                dot = fmaf(o[0], w0.x, dot); dot = fmaf(o[1], w0.y, dot); dot = fmaf(o[2], w0.z, dot); dot = fmaf(o[3], w0.w, dot);
                dot = fmaf(o[4], w1.x, dot); dot = fmaf(o[5], w1.y, dot); dot = fmaf(o[6], w1.z, dot); dot = fmaf(o[7], w1.w, dot);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(tu32x4{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])},
                                                       ro, oy + j * 64, 0, 0);
            }
        }
        if constexpr (FIN) {
            dot = reduce_q(dot);                                 // the 4 lanes (q) of a pixel
            if (q == 0) finp[(size_t)g * 16 + lp] = dot + finb;
        }
    };

    tu32x4 xa[NKS], ya[NH], xb[NKS], yb[NH];
    int g = (int)blockIdx.x * 4 + wave;
    if (g < ngroups) issue(g, xa, ya);
    while (g < ngroups) {
        const int g1 = g + gstride;
        if (g1 < ngroups) issue(g1, xb, yb);
        compute(g, xa, ya);
        if (g1 >= ngroups) break;
        g = g1 + gstride;
        if (g < ngroups) issue(g, xa, ya);
        compute(g1, xb, yb);
    }
}

bool tail_rc16_supported(int cin, int c0, int cout, long pix_per_sample) {
    const bool shape = (cin == 128 && cout == 64) || (cin == 64 && cout == 128) || (cin == 256 && cout == 64) || (cin == 128 && cout == 256) ||
                       (cin == 64 && cout == 32) || (cin == 32 && cout == 64) || (cin == 128 && cout == 32);      // (dim-32 networks: the YAML-literal config_v2_2)
    return shape && (c0 == cin || 2 * c0 == cin) && pix_per_sample % 16 == 0 && pix_per_sample * std::max(cin, cout) * 2 < (1L << 31);
}

template <int CIN, int COUT>
static hipError_t launch_tail_rc16(const TailArgs& a, hipStream_t st) {
    const size_t lds = (size_t)COUT * (CIN * 2 + 16) + (6 * COUT + 64) * sizeof(float);
    // every workgroup starts with the weight image + the statistics -> coefficient chain: ~8 passes of 4 x 16 pixels each, but at
    // least ~2048 workgroups in the launch
    const long need = (a.pix_per_sample / 16 + 3) / 4;
    const long gx = std::min<long>(need, std::max<long>(std::max<long>(1, need / 8), (2048 + a.batch - 1) / std::max(1, a.batch)));
    auto go = [&](auto kfn) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        const double px = (double)a.batch * a.pix_per_sample;       // y2, x (concat) in + out, bf16; the 1x1 res_conv on the MFMA
        LaunchScope ls(st, "resblock_tail_rc16_kernel", 2.0 * px * CIN * COUT + (a.fin_w ? 2.0 * px * COUT : 0.0),
                       px * (CIN + (a.fin_w ? 1.0 : 2.0) * COUT) * 2 + (a.fin_w ? px * 4 : 0.0) + 2.0 * CIN * COUT, "<%d, %d, %s> px%ld x %d", CIN, COUT,
                       a.fin_w ? "true, head" : (a.C1 ? "true" : "false"), a.pix_per_sample, a.batch);
        hipLaunchKernelGGL(kfn, dim3((unsigned)gx, a.batch), dim3(256), lds, st, a);
        return hipGetLastError();
    };
    if constexpr ((CIN == 128 && COUT == 64) || (CIN == 64 && COUT == 32)) {      // (the last block of a dim-64 / dim-32 network: a concat)
        if (a.fin_w) return a.C1 ? go(resblock_tail_rc16_kernel<CIN, COUT, true, true>) : hipErrorInvalidValue;
    } else if (a.fin_w) return hipErrorInvalidValue;
    return a.C1 ? go(resblock_tail_rc16_kernel<CIN, COUT, true>) : go(resblock_tail_rc16_kernel<CIN, COUT, false>);
}

hipError_t launch_resblock_tail(TailArgs a, hipStream_t st) {
    if (a.rc_w) {
        const int cin = a.C0 + a.C1;
        if (!(a.y2_bf16 && a.out_bf16) || !a.x0 || (a.C1 && !a.x1) || !a.rc_b || !tail_rc16_supported(cin, a.C0, a.C, a.pix_per_sample))
            return hipErrorInvalidValue;
        if (a.groups <= 0 || a.groups > 32 || a.C % a.groups) return hipErrorInvalidValue;
        if (cin == 128 && a.C == 64) return launch_tail_rc16<128, 64>(a, st);
        if (cin == 64 && a.C == 128) return launch_tail_rc16<64, 128>(a, st);
        if (cin == 256 && a.C == 64) return launch_tail_rc16<256, 64>(a, st);
        if (cin == 64 && a.C == 32) return launch_tail_rc16<64, 32>(a, st);
        if (cin == 32 && a.C == 64) return launch_tail_rc16<32, 64>(a, st);
        if (cin == 128 && a.C == 32) return launch_tail_rc16<128, 32>(a, st);
        return launch_tail_rc16<128, 256>(a, st);
    }
    if (a.y2_bf16 && a.r_bf16 && a.out_bf16 && a.C % 8 == 0) {
        const int octs = a.C / 8;
        int lpp = 1;
        while (lpp < octs && lpp < 64) lpp <<= 1;
        a.lpp = lpp;
        const int vpl = (octs + lpp - 1) / lpp;
        const int ppb = 256 / lpp;
        // every workgroup starts with the statistics -> coefficient chain (a few dependent L2 round trips): give it
        // ~8 pixel passes of work when the launch has enough workgroups to fill the chip anyway
        // (~8 passes per workgroup, at least 2048 workgroups in the launch: 8192 at B = 32, 16384 at B = 64 measured best)
        const int tail_wgs = 0;
        const long need = (a.pix_per_sample + ppb - 1) / ppb;
        const long total = tail_wgs > 0 ? tail_wgs : std::max<long>(2048, need * a.batch / 8);
        const int gx = (int)std::min<long>(need, std::max<long>(1, std::min<long>(2048, total / std::max(1, a.batch))));
        dim3 grid(gx, a.batch);
        const double el = (double)a.batch * a.pix_per_sample * a.C;
        LaunchScope ls(st, "resblock_tail16_kernel", 10.0 * el, 3.0 * el * 2, "<%d> C%d px%ld x %d", vpl, a.C, a.pix_per_sample, a.batch);
        switch (vpl) {
            case 1: hipLaunchKernelGGL(resblock_tail16_kernel<1>, grid, dim3(256), 0, st, a); break;
            case 2: hipLaunchKernelGGL(resblock_tail16_kernel<2>, grid, dim3(256), 0, st, a); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    const int quads = a.C / 4;
    int lpp = 1;
    while (lpp < quads && lpp < 64) lpp <<= 1;
    a.lpp = lpp;
    const int vpl = (quads + lpp - 1) / lpp;
    const int ppb = 256 / lpp;
    const int tail_wgs32 = 0;
    const long need32 = (a.pix_per_sample + ppb - 1) / ppb;
    const long total32 = tail_wgs32 > 0 ? tail_wgs32 : std::max<long>(2048, need32 * a.batch / 8);
    int gx = (int)std::min<long>(need32, std::max<long>(1, std::min<long>(2048, total32 / std::max(1, a.batch))));
    dim3 grid(gx, a.batch);
    const double el = (double)a.batch * a.pix_per_sample * a.C;
    LaunchScope ls(st, "resblock_tail_kernel", 10.0 * el, el * ((a.y2_bf16 ? 2.0 : 4.0) + (a.r_bf16 ? 2.0 : 4.0) + (a.out_bf16 ? 2.0 : 4.0)), "<%d> C%d px%ld x %d y2_16 %d", vpl == 3 ? 4 : vpl,
                   a.C, a.pix_per_sample, a.batch, a.y2_bf16);
    switch (vpl) {
        case 1: hipLaunchKernelGGL(resblock_tail_kernel<1>, grid, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(resblock_tail_kernel<2>, grid, dim3(256), 0, st, a); break;
        case 3: case 4: hipLaunchKernelGGL(resblock_tail_kernel<4>, grid, dim3(256), 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// init conv: x is the EXTERNAL layout [B, Cin, F, H, W]; y is channel-last [B, F, H, W, Cout].
// One thread = one output pixel x 16 output channels; weights are wave-uniform (scalar loads).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void init_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        int B, int Cin, int F, int H, int W, int Cout, int K, int y_bf16) {
    extern __shared__ float tile[];                       // [Cin][16+K-1][16+K-1]
    const int pad = K / 2, TW = 16 + K - 1;
    const int tx = blockIdx.x % ((W + 15) / 16), ty = blockIdx.x / ((W + 15) / 16);
    const int f = blockIdx.y % F, b = blockIdx.y / F;
    const int cg = blockIdx.z;                            // group of 16 output channels
    for (int i = threadIdx.x; i < Cin * TW * TW; i += 256) {
        const int c = i / (TW * TW), r = i % (TW * TW);
        const int gy = ty * 16 + r / TW - pad, gx = tx * 16 + r % TW - pad;
        float v = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = x[((((size_t)b * Cin + c) * F + f) * H + gy) * W + gx];
        tile[i] = v;
    }
    __syncthreads();
    const int px = threadIdx.x & 15, py = threadIdx.x >> 4;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    const int co0 = cg * 16;
    for (int c = 0; c < Cin; ++c)
        for (int ky = 0; ky < K; ++ky)
            for (int kx = 0; kx < K; ++kx) {
                const float v = tile[(c * TW + py + ky) * TW + px + kx];
                const float* wr = w + ((size_t)(ky * K + kx) * Cin + c) * Cout + co0;   // Flax (kh,kw,Cin,Cout)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = fmaf(v, (co0 + j < Cout) ? wr[j] : 0.f, acc[j]);
            }
    const int oy = ty * 16 + py, ox = tx * 16 + px;
    if (oy < H && ox < W) {
        const size_t ob = ((((size_t)b * F + f) * H + oy) * W + ox) * Cout + co0;
        if (co0 + 16 <= Cout) {
#pragma unroll
            for (int j = 0; j < 16; j += 4)
                store4_f32_or_bf16(y, ob + j, make_float4(acc[j] + bias[co0 + j], acc[j + 1] + bias[co0 + j + 1],
                                                          acc[j + 2] + bias[co0 + j + 2], acc[j + 3] + bias[co0 + j + 3]), y_bf16);
        } else {
            for (int j = 0; j < 16 && co0 + j < Cout; ++j) {
                if (y_bf16) reinterpret_cast<__bf16*>(y)[ob + j] = (__bf16)(acc[j] + bias[co0 + j]);
                else y[ob + j] = acc[j] + bias[co0 + j];
            }
        }
    }
}

// bf16-mode stem for Cin == 1 on MFMA: im2col of the 16x16 pixel tile in LDS (row = pixel, K = ky * 8 + kx with the kx = 7
// slot zero: one 16-byte LDS write per kernel row), weights repacked to [Cout][K] bf16 per workgroup, 64 output channels.
// Same arithmetic class as every other conv of bf16 mode (bf16 operands, fp32 accumulate).
__global__ __launch_bounds__(256) void init_conv_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             int B, int F, int H, int W, int Cout, int K, int y_bf16) {
    using M = Mma<MODE_BF16>;
    constexpr int KP = 64, RSK = KP * 2 + 16;                 // K (padded) and LDS row stride in bytes
    constexpr int TWM = 16 + 8 - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = reinterpret_cast<float*>(smem);             // [TW][TW] input halo tile, TW = 16 + K - 1
    char* wl = smem + ((TWM * TWM * 4 + 15) / 16) * 16;       // [64 co][RSK]
    char* col = wl + 64 * RSK;                                // [256 px][RSK]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lp = lane & 15, q = lane >> 4;
    const int pad = K / 2, TW = 16 + K - 1;
    const int tx = blockIdx.x % ((W + 15) / 16), ty = blockIdx.x / ((W + 15) / 16);
    const int f = blockIdx.y % F, b = blockIdx.y / F;
    for (int i = tid; i < TW * TW; i += 256) {
        const int gy = ty * 16 + i / TW - pad, gx = tx * 16 + i % TW - pad;
        tile[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? x[(((size_t)b * F + f) * H + gy) * W + gx] : 0.f;
    }
    for (int i = tid; i < 64 * RSK / 4; i += 256) reinterpret_cast<unsigned*>(wl)[i] = 0u;
    __syncthreads();
    {   // Flax (kh, kw, 1, Cout): <= 4096 weights, all of a thread's loads issued before the first LDS store
        float wr[16];
        const int nw = K * K * Cout;
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int i = tid + u * 256; wr[u] = w[i < nw ? i : 0]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = tid + u * 256;
            if (i < nw) {
                const int co = i % Cout, t = i / Cout;
                if (co < 64) M::store1(wl + co * RSK, (t / K) * 8 + (t % K), wr[u]);
            }
        }
    }
    {   // this thread's pixel row of the im2col matrix
        const int py = tid >> 4, px = tid & 15;
        char* row = col + tid * RSK;
#pragma unroll
        for (int ky = 0; ky < 8; ++ky) {
            float v[8];
#pragma unroll
            for (int kx = 0; kx < 8; ++kx) v[kx] = (ky < K && kx < K) ? tile[(py + ky) * TW + px + kx] : 0.f;
            *reinterpret_cast<uint4*>(row + ky * 16) =
                make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        }
    }
    __syncthreads();
    f32x4 acc[4][4];                                          // [co tile][pixel tile of this wave]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        uint4 af[4], bf[4];
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) af[tm] = *reinterpret_cast<const uint4*>(wl + (tm * 16 + lp) * RSK + ch * 64 + q * 16);
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) bf[tn] = *reinterpret_cast<const uint4*>(col + ((wv * 4 + tn) * 16 + lp) * RSK + ch * 64 + q * 16);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) M::mma(acc[tm][tn], af[tm], bf[tn]);
    }
    if (y_bf16 && Cout == 64) {
        // bf16 output: meet in LDS ([pixel][64 channels] = 128-byte rows over the im2col buffer) so that the global stores are
        // 16 bytes per lane and a tile row of 16 pixels is one contiguous 2 KB run (the accumulator layout would write 32-byte runs)
        __syncthreads();                                       // every wave is done reading col
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                const float4 bi = *reinterpret_cast<const float4*>(bias + tm * 16 + 4 * q);
                *reinterpret_cast<uint2*>(col + ((wv * 4 + tn) * 16 + lp) * RSK + (tm * 16 + 4 * q) * 2) =
                    make_uint2(pack_bf16x2(acc[tm][tn][0] + bi.x, acc[tm][tn][1] + bi.y), pack_bf16x2(acc[tm][tn][2] + bi.z, acc[tm][tn][3] + bi.w));
            }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * 256, p = i >> 3, pc = i & 7;
            const int oy = ty * 16 + (p >> 4), ox = tx * 16 + (p & 15);
            if (oy < H && ox < W)
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(y) + (((((size_t)b * F + f) * H + oy) * W + ox) * 64) * 2 + pc * 16) =
                    *reinterpret_cast<const uint4*>(col + p * RSK + pc * 16);
        }
        return;
    }
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
        const int p = (wv * 4 + tn) * 16 + lp;
        const int oy = ty * 16 + (p >> 4), ox = tx * 16 + (p & 15);
        if (oy >= H || ox >= W) continue;
        const size_t ob = ((((size_t)b * F + f) * H + oy) * W + ox) * Cout;
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            const int co = tm * 16 + 4 * q;
            if (co >= Cout) continue;
            const float4 bi = *reinterpret_cast<const float4*>(bias + co);
            store4_f32_or_bf16(y, ob + co, make_float4(acc[tm][tn][0] + bi.x, acc[tm][tn][1] + bi.y, acc[tm][tn][2] + bi.z, acc[tm][tn][3] + bi.w), y_bf16);
        }
    }
}

hipError_t launch_init_conv(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int F, int H, int W,
                            int Cout, int K, int y_bf16, hipStream_t st) {
    return launch_init_conv_mode(MODE_F32, x, w, bias, y, B, Cin, F, H, W, Cout, K, y_bf16, st);
}

hipError_t launch_init_conv_mode(int mode, const float* x, const float* w, const float* bias, float* y, int B, int Cin, int F, int H, int W,
                                 int Cout, int K, int y_bf16, hipStream_t st) {
    if (mode == MODE_BF16 && Cin == 1 && K <= 8 && Cout <= 64 && Cout % 4 == 0) {
        const size_t lds = ((23 * 23 * 4 + 15) / 16) * 16 + (size_t)(64 + 256) * (64 * 2 + 16);
        dim3 grid(((W + 15) / 16) * ((H + 15) / 16), B * F);
        const double px = (double)B * F * H * W;
        LaunchScope ls(st, "init_conv_mfma_kernel", 2.0 * px * K * K * Cout, px * (4.0 + Cout * (y_bf16 ? 2.0 : 4.0)), "k%d 1->%d %dx%dx%d", K, Cout, B * F, H, W);
        hipLaunchKernelGGL(init_conv_mfma_kernel, grid, dim3(256), lds, st, x, w, bias, y, B, F, H, W, Cout, K, y_bf16);
        return hipGetLastError();
    }
    const int TW = 16 + K - 1;
    dim3 grid(((W + 15) / 16) * ((H + 15) / 16), B * F, (Cout + 15) / 16);
    const double px = (double)B * F * H * W;
    LaunchScope ls(st, "init_conv_kernel", 2.0 * px * K * K * Cin * Cout, px * (4.0 * Cin + Cout * (y_bf16 ? 2.0 : 4.0)), "k%d %d->%d %dx%dx%d", K, Cin, Cout, B * F, H, W);
    hipLaunchKernelGGL(init_conv_kernel, grid, dim3(256), (size_t)Cin * TW * TW * 4, st, x, w, bias, y, B, Cin, F, H, W, Cout, K, y_bf16);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// final 1x1 conv D -> Cout (Cout tiny: 1 or 3).  x channel-last [npix, D]; y channel-last [npix, Cout]
// (for Cout == 1 this is bit-identical memory to the external [B,1,F,H,W]).  LPP lanes per pixel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void final_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         long npix, int D, int Cout, int lpp, int x_bf16) {
    const int sub = threadIdx.x % lpp, pl = threadIdx.x / lpp, ppb = 256 / lpp;
    for (long pix = (long)blockIdx.x * ppb + pl; pix < npix; pix += (long)gridDim.x * ppb) {
        for (int co = 0; co < Cout; ++co) {
            float s = 0.f;
            if (x_bf16) {                                 // bf16 activation storage: 8 channels = 16 bytes per lane
                for (int c = sub * 8; c < D; c += lpp * 8) {
                    float v[8];
                    unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(x) + ((size_t)pix * D + c) * 2), v);
#pragma unroll
                    for (int k = 0; k < 8; ++k) s = fmaf(v[k], w[(size_t)(c + k) * Cout + co], s);
                }
            } else {
                for (int c = sub * 4; c < D; c += lpp * 4) {
                    const float4 v = *reinterpret_cast<const float4*>(x + (size_t)pix * D + c);
                    s += v.x * w[(size_t)c * Cout + co] + v.y * w[(size_t)(c + 1) * Cout + co]
                       + v.z * w[(size_t)(c + 2) * Cout + co] + v.w * w[(size_t)(c + 3) * Cout + co];
                }
            }
            for (int o = 1; o < lpp; o <<= 1) s += __shfl_xor(s, o);
            if (sub == 0) y[(size_t)pix * Cout + co] = s + bias[co];
        }
    }
}

// bf16 input whose D channels are one 8-channel piece per lane of a pixel's lane group (D = 8 * lpp): the lane's weights live in
// registers and four pixels are in flight per lane group (the generic form has one 16-byte load in flight per lane and re-reads
// its weights from L1 for every pixel)
template <int COUT>
__global__ __launch_bounds__(256) void final_conv16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, long npix, int D, int lpp) {
    constexpr int U = 4;
    const int sub = threadIdx.x % lpp, pl = threadIdx.x / lpp, ppb = 256 / lpp;
    float wr[COUT][8], br[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        br[co] = bias[co];
#pragma unroll
        for (int k = 0; k < 8; ++k) wr[co][k] = w[(size_t)(sub * 8 + k) * COUT + co];
    }
    for (long g0 = (long)blockIdx.x * U; g0 * ppb < npix; g0 += (long)gridDim.x * U) {
        uint4 raw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long pix = (g0 + u) * ppb + pl;
            raw[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(x) + ((size_t)(pix < npix ? pix : 0) * D + sub * 8) * 2);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long pix = (g0 + u) * ppb + pl;
            float v[8];
            unpack8(raw[u], v);
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) s = fmaf(v[k], wr[co][k], s);
                for (int o = 1; o < lpp; o <<= 1) s += __shfl_xor(s, o);
                if (sub == 0 && pix < npix) y[(size_t)pix * COUT + co] = s + br[co];
            }
        }
    }
}

hipError_t launch_final_conv(const float* x, const float* w, const float* bias, float* y, long npix, int D, int Cout, int x_bf16, hipStream_t st) {
    if (x_bf16 && D % 8 == 0 && D <= 128 && ((D / 8) & (D / 8 - 1)) == 0 && Cout >= 1 && Cout <= 4) {
        const int lpp = D / 8, ppb = 256 / lpp;
        const int blocks = (int)std::min<long>((npix + 4L * ppb - 1) / (4L * ppb), 4096);
        LaunchScope ls(st, "final_conv16_kernel", 2.0 * npix * D * Cout, (double)npix * (D * 2.0 + Cout * 4.0), "<%d> D%d px%ld", Cout, D, npix);
        switch (Cout) {
            case 1: hipLaunchKernelGGL(final_conv16_kernel<1>, dim3(blocks), dim3(256), 0, st, x, w, bias, y, npix, D, lpp); break;
            case 2: hipLaunchKernelGGL(final_conv16_kernel<2>, dim3(blocks), dim3(256), 0, st, x, w, bias, y, npix, D, lpp); break;
            case 3: hipLaunchKernelGGL(final_conv16_kernel<3>, dim3(blocks), dim3(256), 0, st, x, w, bias, y, npix, D, lpp); break;
            default: hipLaunchKernelGGL(final_conv16_kernel<4>, dim3(blocks), dim3(256), 0, st, x, w, bias, y, npix, D, lpp); break;
        }
        return hipGetLastError();
    }
    const int per = (x_bf16 && D % 8 == 0) ? 8 : 4;       // channels per lane and pass
    if (per == 4) x_bf16 = x_bf16 ? -1 : 0;
    if (x_bf16 < 0) return hipErrorInvalidValue;          // bf16 tensors have D % 8 == 0 (config check)
    int lpp = 1;
    while (lpp * per < D && lpp < 16) lpp <<= 1;
    const int ppb = 256 / lpp;
    const int blocks = (int)std::min<long>((npix + ppb - 1) / ppb, 4096);
    LaunchScope ls(st, "final_conv_kernel", 2.0 * npix * D * Cout, (double)npix * (D * (x_bf16 ? 2.0 : 4.0) + Cout * 4.0), "D%d->%d px%ld", D, Cout, npix);
    hipLaunchKernelGGL(final_conv_kernel, dim3(blocks), dim3(256), 0, st, x, w, bias, y, npix, D, Cout, lpp, x_bf16);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// time MLP: one workgroup per sample.  temb[b] = [Linear2(GELU(Linear1(sinusoid(t_b)))) | cond-mix]
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_tanh_f(float x) {
    return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}

__global__ __launch_bounds__(256) void time_mlp_kernel(TimeMlpArgs P) {
    extern __shared__ float sm[];                  // emb[dim] | h[time_dim]
    float* emb = sm;
    float* h = sm + P.dim;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int half = P.dim / 2;
    const float tval = P.t_is_device_scalar ? (float)P.time[0] : (float)P.time[b];
    for (int i = tid; i < half; i += 256) {
        const float fr = expf((float)i * -(logf(10000.0f) / (float)(half - 1)));
        const float arg = tval * fr;
        emb[i] = sinf(arg);
        emb[half + i] = cosf(arg);
    }
    __syncthreads();
    for (int n = tid; n < P.time_dim; n += 256) {
        // four independent partial sums, eight weight loads in flight: the kernel is one dependent chain per output otherwise
        // (36 us per step for 0.3 MFLOP at B = 64)
        float a0 = P.b1[n], a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 2
        for (int k = 0; k < P.dim; k += 4) {
            const float w0 = P.w1[(size_t)k * P.time_dim + n], w1 = P.w1[(size_t)(k + 1) * P.time_dim + n];
            const float w2 = P.w1[(size_t)(k + 2) * P.time_dim + n], w3 = P.w1[(size_t)(k + 3) * P.time_dim + n];
            a0 = fmaf(emb[k], w0, a0); a1 = fmaf(emb[k + 1], w1, a1); a2 = fmaf(emb[k + 2], w2, a2); a3 = fmaf(emb[k + 3], w3, a3);
        }
        h[n] = gelu_tanh_f((a0 + a1) + (a2 + a3));
    }
    __syncthreads();
    float* out = P.temb + (size_t)b * P.temb_dim;
    for (int n = tid; n < P.time_dim; n += 256) {
        float a0 = P.b2[n], a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 2
        for (int k = 0; k < P.time_dim; k += 4) {
            const float w0 = P.w2[(size_t)k * P.time_dim + n], w1 = P.w2[(size_t)(k + 1) * P.time_dim + n];
            const float w2 = P.w2[(size_t)(k + 2) * P.time_dim + n], w3 = P.w2[(size_t)(k + 3) * P.time_dim + n];
            a0 = fmaf(h[k], w0, a0); a1 = fmaf(h[k + 1], w1, a1); a2 = fmaf(h[k + 2], w2, a2); a3 = fmaf(h[k + 3], w3, a3);
        }
        out[n] = (a0 + a1) + (a2 + a3);
    }
    if (P.cond_dim) {
        const bool use_null = P.cond_mask ? (P.cond_mask[b] != 0) : (P.null_all != 0);
        for (int n = tid; n < P.cond_dim; n += 256)
            out[P.time_dim + n] = use_null ? P.null_cond_emb[n] : P.cond[(size_t)b * P.cond_dim + n];
    }
}

hipError_t launch_time_mlp(const TimeMlpArgs& a, int B, hipStream_t st) {
    LaunchScope ls(st, "time_mlp_kernel", 2.0 * B * (a.dim * a.time_dim + (double)a.time_dim * a.time_dim), 4.0 * (a.dim * a.time_dim + (double)a.time_dim * a.time_dim), "dim%d B%d", a.dim, B);
    hipLaunchKernelGGL(time_mlp_kernel, dim3(B), dim3(256), (size_t)(a.dim + a.time_dim) * 4, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// per-ResnetBlock scale/shift: ss[l][b] = LayerNorm_{2c}(Linear(SiLU(temb[b])))
// Two launches: (1) the Linear, parallel over (layer, 64-column group, sample group): thread = (column, k quarter), 8 samples
// per pass, the weight row read once for all of them, coalesced; (2) the LayerNorm per (sample, layer).
// ------------------------------------------------------------------------------------------------
constexpr int SS_BG = 8;                            // samples per workgroup of the Linear
__global__ __launch_bounds__(256) void resblock_ss_lin_kernel(const float* __restrict__ params, const float* __restrict__ temb,
                                                              const SsLayer* __restrict__ layers, float* __restrict__ lin_base,
                                                              int temb_dim, int B) {
    extern __shared__ float sm[];                  // act[SS_BG][temb_dim] | part[4][SS_BG][64]
    float* act = sm;
    float* part = sm + SS_BG * temb_dim;
    const SsLayer L = layers[blockIdx.x];
    const int N = L.n, n0 = blockIdx.y * 64;
    if (n0 >= N) return;                            // uniform: layers differ in width
    const int b0 = blockIdx.z * SS_BG, nb = min(SS_BG, B - b0);
    const int tid = threadIdx.x, col = tid & 63, kq = tid >> 6;
    for (int i = tid; i < nb * temb_dim; i += 256) act[i] = silu_f(temb[(size_t)b0 * temb_dim + i]);
    __syncthreads();
    const int n = n0 + col;
    const float* W = params + L.w_off;
    float acc[SS_BG];
#pragma unroll
    for (int j = 0; j < SS_BG; ++j) acc[j] = 0.f;
    const int kper = (temb_dim + 3) / 4, k0 = kq * kper, k1 = min(temb_dim, k0 + kper);
    if (n < N)
#pragma unroll 8
        for (int k = k0; k < k1; ++k) {              // (unrolled: eight weight loads in flight instead of one dependent trip per k)
            const float w = W[(size_t)k * N + n];
#pragma unroll
            for (int j = 0; j < SS_BG; ++j) acc[j] = fmaf(act[j * temb_dim + k], w, acc[j]);   // rows past nb hold stale LDS: never stored
        }
#pragma unroll
    for (int j = 0; j < SS_BG; ++j) part[(kq * SS_BG + j) * 64 + col] = acc[j];
    __syncthreads();
    if (kq == 0 && n < N) {
        const float bias = params[L.b_off + n];
        for (int j = 0; j < nb; ++j)
            lin_base[(size_t)L.out_off * B + (size_t)(b0 + j) * N + n] =
                bias + part[(0 * SS_BG + j) * 64 + col] + part[(1 * SS_BG + j) * 64 + col] + part[(2 * SS_BG + j) * 64 + col] + part[(3 * SS_BG + j) * 64 + col];
    }
}

__global__ __launch_bounds__(256) void resblock_ss_norm_kernel(const float* __restrict__ params, const SsLayer* __restrict__ layers,
                                                               const float* __restrict__ lin_base, float* __restrict__ ss_base, int B) {
    __shared__ float red[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const SsLayer L = layers[blockIdx.y];
    const int N = L.n;                              // 2 * cout  (<= 2048)
    const float* lin = lin_base + (size_t)L.out_off * B + (size_t)b * N;
    float v[8];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = tid + 256 * j;
        v[j] = (n < N) ? lin[n] : 0.f;
        s += v[j]; ss += v[j] * v[j];
    }
    for (int o = 1; o < 64; o <<= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
    if ((tid & 63) == 0) { red[tid >> 6] = s; red[4 + (tid >> 6)] = ss; }
    __syncthreads();
    s = red[0] + red[1] + red[2] + red[3];
    ss = red[4] + red[5] + red[6] + red[7];
    const float mean = s / (float)N;
    const float var = fmaxf(ss / (float)N - mean * mean, 0.f);
    const float rstd = rsqrtf(var + NORM_EPS);
    const float* g = params + L.g_off;
    const float* be = params + L.be_off;
    float* out = ss_base + (size_t)L.out_off * B + (size_t)b * N;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = tid + 256 * j;
        if (n < N) out[n] = fmaf((v[j] - mean) * rstd, g[n], be[n]);
    }
}

hipError_t launch_resblock_ss(const float* params, const float* temb, const SsLayer* layers, int nlayers, float* ss_base,
                              float* lin_base, int temb_dim, int B, int max_n, hipStream_t st) {
    // lin_base (the pre-LayerNorm values, also what the backward reads) is required scratch
    if (!lin_base || nlayers <= 0) return hipErrorInvalidValue;
    const size_t lds = ((size_t)SS_BG * temb_dim + 4 * SS_BG * 64) * 4;
    hipError_t e;
    {
        LaunchScope ls(st, "resblock_ss_lin_kernel", 0.0, 0.0, "layers%d B%d", nlayers, B);
        hipLaunchKernelGGL(resblock_ss_lin_kernel, dim3(nlayers, (max_n + 63) / 64, (B + SS_BG - 1) / SS_BG), dim3(256), lds, st, params, temb, layers,
                           lin_base, temb_dim, B);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    LaunchScope ls(st, "resblock_ss_norm_kernel", 0.0, 0.0, "layers%d B%d", nlayers, B);
    hipLaunchKernelGGL(resblock_ss_norm_kernel, dim3(B, nlayers), dim3(256), 0, st, params, layers, lin_base, ss_base, B);
    return hipGetLastError();
}

}  // namespace vdx
