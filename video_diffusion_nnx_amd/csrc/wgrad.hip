// Weight gradients of the (1,kh,kw) convolutions and 1x1 projections on exact-f32 MFMA (gfx950), and column sums.
//
//   dW[tap][ci][co] = sum_{pixels p} Xhat[p (+) tap][ci] * dY[p][co]        (autodiff of nnx.Conv / nnx.ConvTranspose,
//                                                                            reference trainer.py:361 jax.value_and_grad)
// Xhat is the conv's EFFECTIVE input: the channel-concat of two tensors and/or the fused prologue
// SiLU(GroupNorm(x)*(scale+1)+shift) of conv_igemm -- recomputed here while staging, never stored.
// GEMM view: M = ci, N = co, K = pixels.  Activations are channel-last, i.e. K-strided, which bf16 MFMA operands
// cannot read without a transpose; v_mfma_f32_16x16x4_f32 takes ONE f32 per lane per operand (A[r][k=q], B[k=q][r]),
// so the natural [pixel][channel] LDS image serves both operands with ds_read_b32.  All taps accumulate at once from a
// single staged halo tile (<= 9 accumulator sets of 2x2 tiles per wave).  Split-K over pixel chunks; results are
// added to the fp32 gradient buffer with float atomics (dW is tiny next to the activations).
#include "vdx_common.h"
#include <type_traits>
#include "vdx_internal.h"
#include <stdlib.h>

namespace vdx {

constexpr int WG_LD = 80;          // floats per LDS pixel row (64 channels + pad: q*80 % 32 = q*16 -> conflict-free halves)

template <int NT>                   // taps handled by one workgroup (<= 9)
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int wi = w & 1, wo = w >> 1;                       // 32-channel sub-tiles of the 64x64 (ci, co) tile
    const int ci0 = blockIdx.y * 64, co0 = (blockIdx.z % P.co_tiles) * 64;
    const int tap0 = (blockIdx.z / P.co_tiles) * NT;
    const int Cin = P.C0 + P.C1;
    const int IH = (P.PH - 1) * P.sa + P.ext, IW = (P.PW - 1) * P.sa + P.ext;     // staged input window
    const int BH = P.PH * P.sb, BW = P.PW * P.sb;                                 // staged dY window
    const int HPX = IH * IW, BPX = BH * BW;

    float* As = reinterpret_cast<float*>(smem);              // [HPX][WG_LD]
    float* Bs = As + (size_t)HPX * WG_LD;                    // [BPX][WG_LD]
    float* coefA = Bs + (size_t)BPX * WG_LD;                 // [64] (prologue)
    float* coefD = coefA + 64;
    float* gm = coefD + 64;                                  // [64]
    int* tapA = reinterpret_cast<int*>(gm + 64);             // [NT] LDS offsets (floats) of the tap shift in As / Bs
    int* tapB = tapA + 16;

    if (tid < NT) {
        const int t = tap0 + tid;
        int ay, ax, by = 0, bx = 0;
        if (P.kind == 0) { ay = t / P.kw; ax = t % P.kw; }
        else { const int jy = t >> 2, jx = t & 3; ay = (jy >> 1) + (jy & 1); ax = (jx >> 1) + (jx & 1); by = jy & 1; bx = jx & 1; }
        tapA[tid] = (ay * IW + ax) * WG_LD;
        tapB[tid] = (by * BW + bx) * WG_LD;
    }

    f32x4 acc[NT][2][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) { acc[t][i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int tiles_x = (P.Wm + P.PW - 1) / P.PW, tiles_y = (P.Hm + P.PH - 1) / P.PH;
    const int patches_per_frame = tiles_x * tiles_y;
    const long total_patches = (long)P.NF * patches_per_frame;
    int last_b = -1;
    const bool do_bias = (blockIdx.y == 0) && (blockIdx.z / P.co_tiles == 0);
    float bias_acc = 0.f;
    for (long pid = blockIdx.x; pid < total_patches; pid += gridDim.x) {
        const int f = (int)(pid / patches_per_frame);
        const int pr = (int)(pid % patches_per_frame);
        const int ty = pr / tiles_x, tx = pr % tiles_x;
        const int my0 = ty * P.PH, mx0 = tx * P.PW;           // patch origin in "m" coordinates (conv: output pixels)
        const int iy0 = my0 * P.sa - P.halo, ix0 = mx0 * P.sa - P.halo;
        const int b = f / P.F;
        __syncthreads();                                      // previous patch fully consumed
        if (P.pro && b != last_b) {                           // per-sample prologue coefficients for this channel tile
            gn_mean_rstd_wg(P.in_stats, b, P.groups, (double)P.F * P.H * P.W * (Cin / P.groups), gm, tid, 256);
            __syncthreads();
            if (tid < 64) {
                const int c = ci0 + tid;
                float a = 0.f, d = 0.f;
                if (c < Cin) {
                    const int g = c / (Cin / P.groups);
                    const float ga = P.gamma[c], be = P.beta[c];
                    float sc = 1.f, sh = 0.f;
                    if (P.ss) { sc = P.ss[(size_t)b * P.ss_stride + c] + 1.f; sh = P.ss[(size_t)b * P.ss_stride + Cin + c]; }
                    a = gm[2 * g + 1] * ga * sc;
                    d = (be - gm[2 * g] * gm[2 * g + 1] * ga) * sc + sh;
                }
                coefA[tid] = a; coefD[tid] = d;
            }
            last_b = b;
            __syncthreads();
        }
        // ---- stage the input window (64 channels) and the dY window (64 channels) ----
        for (int i = tid; i < HPX * 16; i += 256) {
            const int hp = i >> 4, pc = i & 15;
            const int iy = hp / IW, ix = hp - iy * IW;
            const int gy = iy0 + iy, gx = ix0 + ix;
            const int c = ci0 + pc * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < P.H && gx >= 0 && gx < P.W && c < Cin) {
                const size_t pix = ((size_t)f * P.H + gy) * P.W + gx;
                v = (c < P.C0) ? load4_f32_or_bf16(P.x0, pix * P.C0 + c, P.x0_bf16)
                               : load4_f32_or_bf16(P.x1, pix * P.C1 + (c - P.C0), P.x0_bf16);
                if (P.pro) {
                    const float4 a = *reinterpret_cast<const float4*>(coefA + pc * 4);
                    const float4 d = *reinterpret_cast<const float4*>(coefD + pc * 4);
                    v.x = silu_f(fmaf(v.x, a.x, d.x)); v.y = silu_f(fmaf(v.y, a.y, d.y));
                    v.z = silu_f(fmaf(v.z, a.z, d.z)); v.w = silu_f(fmaf(v.w, a.w, d.w));
                }
            }
            *reinterpret_cast<float4*>(As + (size_t)hp * WG_LD + pc * 4) = v;
        }
        for (int i = tid; i < BPX * 16; i += 256) {
            const int bp = i >> 4, pc = i & 15;
            const int yy = bp / BW, xx = bp - yy * BW;
            const int gy = my0 * P.sb + yy, gx = mx0 * P.sb + xx;
            const int c = co0 + pc * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy < P.Hy && gx < P.Wy && c < P.Cout) v = load4_f32_or_bf16(P.dy, (((size_t)f * P.Hy + gy) * P.Wy + gx) * P.Cout + c, P.dy_bf16);
            *reinterpret_cast<float4*>(Bs + (size_t)bp * WG_LD + pc * 4) = v;
        }
        __syncthreads();
        if (P.db && do_bias && tid < 64) {                     // column sums of this dY window (every dY pixel is staged exactly once per co tile)
            float t = 0.f;
            for (int bp = 0; bp < BPX; ++bp) t += Bs[(size_t)bp * WG_LD + tid];
            bias_acc += t;
        }
        // ---- K loop over the patch's "m" positions, 4 per MFMA step (positions beyond Hm/Wm hold zero dY) ----
        const int npos = P.PH * P.PW;
        for (int k0 = 0; k0 < npos; k0 += 4) {
            const int p = k0 + q;
            const int py = p / P.PW, px = p - py * P.PW;
            const float* arow = As + (size_t)((py * P.sa) * IW + px * P.sa) * WG_LD + wi * 32 + r;
            const float* brow = Bs + (size_t)((py * P.sb) * BW + px * P.sb) * WG_LD + wo * 32 + r;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float a0 = arow[tapA[t]], a1 = arow[tapA[t] + 16];
                const float b0 = brow[tapB[t]], b1 = brow[tapB[t] + 16];
                acc[t][0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[t][0][0], 0, 0, 0);
                acc[t][0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[t][0][1], 0, 0, 0);
                acc[t][1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[t][1][0], 0, 0, 0);
                acc[t][1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[t][1][1], 0, 0, 0);
            }
        }
    }
    // split > 0: the dY columns of this tile (never straddling: split % 64 == 0) belong to one of up to three weight tensors
    const int sel = P.split ? co0 / P.split : 0;
    const int cbase = P.split ? sel * P.split : 0, cw = P.split ? P.split : P.Cout;
    float* const dWt = sel == 0 ? P.dW : (sel == 1 ? P.dW1 : P.dW2);
    float* const dbt = sel == 0 ? P.db : (sel == 1 ? P.db1 : P.db2);
    float* const pslot = P.part ? P.part + (size_t)blockIdx.x * P.part_E : nullptr;       // deterministic mode: this chunk's slot
    if (dbt && do_bias && tid < 64 && co0 + tid < P.Cout) {
        if (pslot) P.part_b[(size_t)blockIdx.x * P.Cout + co0 + tid] = bias_acc;
        else atomicAdd(dbt + co0 - cbase + tid, bias_acc);
    }
    // ---- accumulate into dW (Flax layout [taps][Cin][Cout]) ----
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tap = tap0 + t;
        if (tap >= P.taps) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = co0 + wo * 32 + j * 16 + r;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ci = ci0 + wi * 32 + i * 16 + 4 * q + e;
                    if (ci < Cin && co < P.Cout) {
                        if (pslot) pslot[((size_t)tap * Cin + ci) * P.Cout + co] = acc[t][i][j][e];
                        else atomicAdd(dWt + ((size_t)tap * Cin + ci) * cw + co - cbase, acc[t][i][j][e]);
                    }
                }
            }
    }
}

// ---- bf16 form (bf16 mode) -------------------------------------------------------------------------------------------
// Same tiling, but both operands are rounded to bf16 while staged (like every other contraction of bf16 mode) into natural
// [pixel][64 channels] bf16 LDS images, and the K-strided fragments come from gfx950's transposing LDS read:
// ds_read_b64_tr_b16 gives lane i of a 16-lane group column i of a 4-row x 16-column block, every lane supplying its own
// row address -- so the 4 rows are 4 consecutive patch positions (not contiguous in the halo tile) and the result is the
// 4 consecutive-K elements of channel i that v_mfma_f32_16x16x32_bf16 wants (two reads per 32-deep operand).
// One MFMA covers 32 positions where the f32 form needs 8 (16x16x4, 4 per step at twice the cycles): the kernel becomes
// staging-bound.  Bias column sums are taken from the fp32 values while staging (exact).
typedef short s16x4t __attribute__((ext_vector_type(4)));
constexpr int WG_RSB = 64 * 2 + 16;         // bytes per LDS pixel row of the bf16 images

__device__ __forceinline__ bf16x8 tr_frag(const char* row0, const char* row1) {
    const s16x4t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4t*)(uintptr_t)(unsigned)(uintptr_t)row0);
    const s16x4t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4t*)(uintptr_t)(unsigned)(uintptr_t)row1);
    typedef short s16x8t __attribute__((ext_vector_type(8)));
    const s16x8t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// NT taps per workgroup, split over NG groups of 4 waves (NTW = taps per wave): fewer live accumulators, and NG x 256 threads
// share the staging of the same patch.
template <int NT, int NG, bool PF, bool X16, bool DY16>
__global__ __launch_bounds__(256 * NG, NG == 1 ? 5 : 1) void conv_wgrad16_kernel(const WgradArgs P) {
    constexpr int NTH = 256 * NG;
    constexpr int NTW = (NT + NG - 1) / NG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) & 3, wg = tid >> 8;
    const int r = lane & 15, q = lane >> 4;
    const int wi = w & 1, wo = w >> 1;
    const int tw0 = wg * NTW;                                // first tap (within the workgroup's NT) of this wave
    const int ci0 = blockIdx.y * 64, co0 = (blockIdx.z % P.co_tiles) * 64;
    const int tap0 = (blockIdx.z / P.co_tiles) * NT;
    const int Cin = P.C0 + P.C1;
    const int IH = (P.PH - 1) * P.sa + P.ext, IW = (P.PW - 1) * P.sa + P.ext;
    const int BH = P.PH * P.sb, BW = P.PW * P.sb;
    const int HPX = IH * IW, BPX = BH * BW;

    char* As = smem;                                         // [HPX][WG_RSB] bf16
    char* Bs = As + (size_t)HPX * WG_RSB;                    // [BPX][WG_RSB] bf16
    float* coefA = reinterpret_cast<float*>(Bs + (size_t)BPX * WG_RSB);
    float* coefD = coefA + 64;
    float* gm = coefD + 64;
    float* bsum = gm + 64;                                   // [64] bias column sums of this workgroup
    int* tapA = reinterpret_cast<int*>(bsum + 64);           // [NT] byte offsets of the tap shift in As / Bs
    int* tapB = tapA + 16;

    if (tid < NT) {
        const int t = tap0 + tid;
        int ay, ax, by = 0, bx = 0;
        if (P.kind == 0) { ay = t / P.kw; ax = t % P.kw; }
        else { const int jy = t >> 2, jx = t & 3; ay = (jy >> 1) + (jy & 1); ax = (jx >> 1) + (jx & 1); by = jy & 1; bx = jx & 1; }
        tapA[tid] = (ay * IW + ax) * WG_RSB;
        tapB[tid] = (by * BW + bx) * WG_RSB;
    }
    if (tid < 64) bsum[tid] = 0.f;

    f32x4 acc[NTW][2][2];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) { acc[t][i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int tiles_x = (P.Wm + P.PW - 1) / P.PW, tiles_y = (P.Hm + P.PH - 1) / P.PH;
    const int patches_per_frame = tiles_x * tiles_y;
    const long total_patches = (long)P.NF * patches_per_frame;
    int last_b = -1;
    const bool do_bias = P.db && (blockIdx.y == 0) && (blockIdx.z / P.co_tiles == 0);
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);         // this thread's 4 channels (i & 15 is constant per thread)
    // transposing-read lane roles: group g = q supplies positions 8g..8g+7 of a 32-position K block; inside the group lane
    // 4*qr + pc supplies row qr (and qr + 4 for the second read), 8-byte column chunk pc
    const int qr = r >> 2, pcz = r & 3;
    // Staging is software-pipelined: the global loads of the NEXT patch are issued into registers (ra / rb) right after the barrier
    // that publishes the current patch, so they are in flight during the MFMA loop (one workgroup per CU: nothing else would hide them).
    constexpr int XA = 6, XB = 4;                             // float4 pieces per thread (launcher checks HPX * 16 <= XA * NTH, BPX * 16 <= XB * NTH)
    float4 ra[XA], rb[XB];
    unsigned va = 0;                                          // in-bounds mask of ra (the prologue applies to in-bounds pixels only)
    // Every piece is loaded unconditionally (out-of-range pieces read the tensor base and are masked when stored) and kept as raw
    // bits: no branch and no conversion between a load and the next one, so all XA + XB loads of a thread are in flight together.
    unsigned vb = 0;
    auto load_patch = [&](long pid) {
        const int f = (int)(pid / patches_per_frame);
        const int pr = (int)(pid % patches_per_frame);
        const int ty = pr / tiles_x, tx = pr % tiles_x;
        const int my0 = ty * P.PH, mx0 = tx * P.PW;
        const int iy0 = my0 * P.sa - P.halo, ix0 = mx0 * P.sa - P.halo;
        va = 0; vb = 0;
#pragma unroll
        for (int u = 0; u < XA; ++u) {
            const int i = tid + u * NTH;
            const int hp = i >> 4, pc = i & 15;
            const int iy = div_magic(hp, P.m_iw), ix = hp - iy * IW;
            const int gy = iy0 + iy, gx = ix0 + ix;
            const int c = ci0 + pc * 4;
            const bool ok = i < HPX * 16 && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W && c < Cin;
            const size_t pix = ok ? ((size_t)f * P.H + gy) * P.W + gx : 0;
            if (X16) {
                const char* src = (!ok || c < P.C0) ? reinterpret_cast<const char*>(P.x0) + (ok ? (pix * P.C0 + c) * 2 : 0)
                                                    : reinterpret_cast<const char*>(P.x1) + (pix * P.C1 + (c - P.C0)) * 2;
                const uint2 t = *reinterpret_cast<const uint2*>(src);
                ra[u] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), 0.f, 0.f);
            } else {
                const float* src = (!ok || c < P.C0) ? P.x0 + (ok ? pix * P.C0 + c : 0) : P.x1 + pix * P.C1 + (c - P.C0);
                ra[u] = *reinterpret_cast<const float4*>(src);
            }
            va |= (ok ? 1u : 0u) << u;
        }
#pragma unroll
        for (int u = 0; u < XB; ++u) {
            const int i = tid + u * NTH;
            const int bp = i >> 4, pc = i & 15;
            const int yy = div_magic(bp, P.m_bw), xx = bp - yy * BW;
            const int gy = my0 * P.sb + yy, gx = mx0 * P.sb + xx;
            const int c = co0 + pc * 4;
            const bool ok = i < BPX * 16 && gy < P.Hy && gx < P.Wy && c < P.Cout;
            const size_t e = ok ? (((size_t)f * P.Hy + gy) * P.Wy + gx) * P.Cout + c : 0;
            if (DY16) {
                const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(P.dy) + e * 2);
                rb[u] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), 0.f, 0.f);
            } else rb[u] = *reinterpret_cast<const float4*>(P.dy + e);
            vb |= (ok ? 1u : 0u) << u;
        }
    };
    auto widen = [](const float4 v) {                         // 4 bf16 held as raw bits in v.x, v.y -> float4
        const unsigned a = __float_as_uint(v.x), b = __float_as_uint(v.y);
        return make_float4(__uint_as_float(a << 16), __uint_as_float(a & 0xFFFF0000u), __uint_as_float(b << 16), __uint_as_float(b & 0xFFFF0000u));
    };
    // consecutive patches per workgroup: the sample index (and with it the GroupNorm prologue coefficients) changes at most
    // (patches per workgroup / patches per sample) + 1 times, and neighbouring patches share their halo rows in L2
    const long p_per = (total_patches + gridDim.x - 1) / gridDim.x;
    const long p_begin = (long)blockIdx.x * p_per, p_end = min(total_patches, p_begin + p_per);
    if (PF && p_begin < p_end) load_patch(p_begin);
    for (long pid = p_begin; pid < p_end; ++pid) {
        const int b = (int)(pid / patches_per_frame) / P.F;
        __syncthreads();                                      // previous patch fully consumed
        if (P.pro && b != last_b) {
            gn_mean_rstd_wg(P.in_stats, b, P.groups, (double)P.F * P.H * P.W * (Cin / P.groups), gm, tid, NTH);
            __syncthreads();
            if (tid < 64) {
                const int c = ci0 + tid;
                float a = 0.f, d = 0.f;
                if (c < Cin) {
                    const int g = c / (Cin / P.groups);
                    const float ga = P.gamma[c], be = P.beta[c];
                    float sc = 1.f, sh = 0.f;
                    if (P.ss) { sc = P.ss[(size_t)b * P.ss_stride + c] + 1.f; sh = P.ss[(size_t)b * P.ss_stride + Cin + c]; }
                    a = gm[2 * g + 1] * ga * sc;
                    d = (be - gm[2 * g] * gm[2 * g + 1] * ga) * sc + sh;
                }
                coefA[tid] = a; coefD[tid] = d;
            }
            last_b = b;
            __syncthreads();
        }
        if constexpr (PF) {
#pragma unroll
            for (int u = 0; u < XA; ++u) {
                const int i = tid + u * NTH;
                if (i < HPX * 16) {
                    const int hp = i >> 4, pc = i & 15;
                    float4 v = X16 ? widen(ra[u]) : ra[u];
                    if (!((va >> u) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (P.pro && ((va >> u) & 1u)) {
                        const float4 a = *reinterpret_cast<const float4*>(coefA + pc * 4);
                        const float4 d = *reinterpret_cast<const float4*>(coefD + pc * 4);
                        v.x = silu_f(fmaf(v.x, a.x, d.x)); v.y = silu_f(fmaf(v.y, a.y, d.y));
                        v.z = silu_f(fmaf(v.z, a.z, d.z)); v.w = silu_f(fmaf(v.w, a.w, d.w));
                    }
                    *reinterpret_cast<uint2*>(As + (size_t)hp * WG_RSB + pc * 8) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                }
            }
#pragma unroll
            for (int u = 0; u < XB; ++u) {
                const int i = tid + u * NTH;
                if (i < BPX * 16) {
                    const int bp = i >> 4, pc = i & 15;
                    float4 v = DY16 ? widen(rb[u]) : rb[u];
                    if (!((vb >> u) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    bias4.x += v.x; bias4.y += v.y; bias4.z += v.z; bias4.w += v.w;
                    *reinterpret_cast<uint2*>(Bs + (size_t)bp * WG_RSB + pc * 8) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                }
            }
        } else {                                              // single-tap tiles: many small workgroups per CU hide the loads; stage in place
            const int f = (int)(pid / patches_per_frame);
            const int pr = (int)(pid % patches_per_frame);
            const int ty = pr / tiles_x, tx = pr % tiles_x;
            const int my0 = ty * P.PH, mx0 = tx * P.PW;
            const int iy0 = my0 * P.sa - P.halo, ix0 = mx0 * P.sa - P.halo;
            for (int i = tid; i < HPX * 16; i += NTH) {
                const int hp = i >> 4, pc = i & 15;
                const int iy = div_magic(hp, P.m_iw), ix = hp - iy * IW;
                const int gy = iy0 + iy, gx = ix0 + ix;
                const int c = ci0 + pc * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gy >= 0 && gy < P.H && gx >= 0 && gx < P.W && c < Cin) {
                    const size_t pix = ((size_t)f * P.H + gy) * P.W + gx;
                    v = (c < P.C0) ? load4_f32_or_bf16(P.x0, pix * P.C0 + c, P.x0_bf16)
                                   : load4_f32_or_bf16(P.x1, pix * P.C1 + (c - P.C0), P.x0_bf16);
                    if (P.pro) {
                        const float4 a = *reinterpret_cast<const float4*>(coefA + pc * 4);
                        const float4 d = *reinterpret_cast<const float4*>(coefD + pc * 4);
                        v.x = silu_f(fmaf(v.x, a.x, d.x)); v.y = silu_f(fmaf(v.y, a.y, d.y));
                        v.z = silu_f(fmaf(v.z, a.z, d.z)); v.w = silu_f(fmaf(v.w, a.w, d.w));
                    }
                }
                *reinterpret_cast<uint2*>(As + (size_t)hp * WG_RSB + pc * 8) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
            }
            for (int i = tid; i < BPX * 16; i += NTH) {
                const int bp = i >> 4, pc = i & 15;
                const int yy = div_magic(bp, P.m_bw), xx = bp - yy * BW;
                const int gy = my0 * P.sb + yy, gx = mx0 * P.sb + xx;
                const int c = co0 + pc * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gy < P.Hy && gx < P.Wy && c < P.Cout) v = load4_f32_or_bf16(P.dy, (((size_t)f * P.Hy + gy) * P.Wy + gx) * P.Cout + c, P.dy_bf16);
                bias4.x += v.x; bias4.y += v.y; bias4.z += v.z; bias4.w += v.w;
                *reinterpret_cast<uint2*>(Bs + (size_t)bp * WG_RSB + pc * 8) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
            }
        }
        __syncthreads();
        if (PF && pid + 1 < p_end) load_patch(pid + 1);
        // ---- K loop: 32 patch positions per MFMA (a lane group's 8 positions are consecutive in one patch row) ----
        const int npos = P.PH * P.PW;
        for (int k0 = 0; k0 < npos; k0 += 32) {
            const int p0 = k0 + 8 * q + qr, p1 = p0 + 4;
            const int py0 = p0 >> P.pwl, px0 = p0 & (P.PW - 1), py1 = p1 >> P.pwl, px1 = p1 & (P.PW - 1);   // PW = 8 or 16
            const char* a0 = As + (size_t)((py0 * P.sa) * IW + px0 * P.sa) * WG_RSB + wi * 64 + pcz * 8;
            const char* a1 = As + (size_t)((py1 * P.sa) * IW + px1 * P.sa) * WG_RSB + wi * 64 + pcz * 8;
            const char* b0 = Bs + (size_t)((py0 * P.sb) * BW + px0 * P.sb) * WG_RSB + wo * 64 + pcz * 8;
            const char* b1 = Bs + (size_t)((py1 * P.sb) * BW + px1 * P.sb) * WG_RSB + wo * 64 + pcz * 8;
            // kind 0: dy is not shifted by the tap, so its fragments are read once per K block and reused by every tap of the wave
            bf16x8 bf0, bf1;
            if (P.kind == 0) { bf0 = tr_frag(b0, b1); bf1 = tr_frag(b0 + 32, b1 + 32); }
#pragma unroll
            for (int tt = 0; tt < NTW; ++tt) {
                const int t = min(tw0 + tt, NT - 1);          // a wave past the last tap repeats it (EXEC stays full); never stored
                const bf16x8 af0 = tr_frag(a0 + tapA[t], a1 + tapA[t]);
                const bf16x8 af1 = tr_frag(a0 + tapA[t] + 32, a1 + tapA[t] + 32);
                if (P.kind != 0) { bf0 = tr_frag(b0 + tapB[t], b1 + tapB[t]); bf1 = tr_frag(b0 + tapB[t] + 32, b1 + tapB[t] + 32); }
                acc[tt][0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0, bf0, acc[tt][0][0], 0, 0, 0);
                acc[tt][0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0, bf1, acc[tt][0][1], 0, 0, 0);
                acc[tt][1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1, bf0, acc[tt][1][0], 0, 0, 0);
                acc[tt][1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1, bf1, acc[tt][1][1], 0, 0, 0);
            }
        }
    }
    const int sel = P.split ? co0 / P.split : 0;
    const int cbase = P.split ? sel * P.split : 0, cw = P.split ? P.split : P.Cout;
    float* const dWt = sel == 0 ? P.dW : (sel == 1 ? P.dW1 : P.dW2);
    float* const dbt = sel == 0 ? P.db : (sel == 1 ? P.db1 : P.db2);
    float* const pslot = P.part ? P.part + (size_t)blockIdx.x * P.part_E : nullptr;       // deterministic mode: this chunk's slot
    if (do_bias) {                                             // (uniform per workgroup)
        const int pc = tid & 15;
        if (pslot) {
            // fixed-order sum of the threads that share a channel quadruple (pc): through the staging area, which nobody reads any more
            __syncthreads();
            float4* bred = reinterpret_cast<float4*>(As);
            bred[tid] = bias4;
            __syncthreads();
            if (tid < 64 && co0 + tid < P.Cout) {
                float t = 0.f;
                for (int k = 0; k < 256 * NG / 16; ++k) { const float4 v = bred[k * 16 + (tid >> 2)]; t += (tid & 3) == 0 ? v.x : (tid & 3) == 1 ? v.y : (tid & 3) == 2 ? v.z : v.w; }
                if (dbt) P.part_b[(size_t)blockIdx.x * P.Cout + co0 + tid] = t;
            }
        } else {
            atomicAdd(&bsum[pc * 4 + 0], bias4.x); atomicAdd(&bsum[pc * 4 + 1], bias4.y);
            atomicAdd(&bsum[pc * 4 + 2], bias4.z); atomicAdd(&bsum[pc * 4 + 3], bias4.w);
            __syncthreads();
            if (dbt && tid < 64 && co0 + tid < P.Cout) atomicAdd(dbt + co0 - cbase + tid, bsum[tid]);
        }
    }
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) {
        const int tap = tap0 + tw0 + tt;
        if (tw0 + tt >= NT || tap >= P.taps) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = co0 + wo * 32 + j * 16 + r;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ci = ci0 + wi * 32 + i * 16 + 4 * q + e;
                    if (ci < Cin && co < P.Cout) {
                        if (pslot) pslot[((size_t)tap * Cin + ci) * P.Cout + co] = acc[tt][i][j][e];
                        else atomicAdd(dWt + ((size_t)tap * Cin + ci) * cw + co - cbase, acc[tt][i][j][e]);
                    }
                }
            }
    }
}

// ---- 1x1 convolutions (projections) in bf16 mode: dW[Cin][Cout] += X^T dY as a split-K GEMM over the pixel rows -----------------------
// Workgroup tile = 64 input channels x 64 * NCO output channels over a contiguous range of 64-row K tiles (wave (wi, wo): 32 ci x
// NCO x 32 co).  With the q|k|v gradients in one [rows][768] tensor, NCO = 4 stages x three times instead of twelve.  All global
// loads of a K tile are issued unconditionally as raw bits (out-of-range pieces read the tensor base and are zeroed when stored),
// one tile ahead of the MFMA loop that consumes them.
template <bool X16, bool DY16, int NCO>
__global__ __launch_bounds__(256, NCO == 4 ? 2 : 4) void wgrad1x1_kernel(const WgradArgs P, const long rows, const int tiles_per_wg) {
    constexpr int RSA = WG_RSB, RSBB = NCO * 128 + 16;
    constexpr int XA = 4, XB = 4 * NCO;                           // 16-byte pieces per thread per K tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                                              // [64 rows][RSA]
    char* Bs = As + 64 * RSA;                                     // [64 rows][RSBB]
    float* bsum = reinterpret_cast<float*>(Bs + 64 * RSBB);      // [64 * NCO]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, q = lane >> 4, qr = r >> 2, pcz = r & 3;
    const int wi = w & 1, wo = w >> 1;
    const int ci0 = blockIdx.y * 64, co0 = blockIdx.z * 64 * NCO;
    const int Cin = P.C0 + P.C1;
    const bool do_bias = P.db && blockIdx.y == 0;
    for (int i = tid; i < 64 * NCO; i += 256) bsum[i] = 0.f;

    f32x4 acc[NCO][2][2];
#pragma unroll
    for (int s = 0; s < NCO; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) { acc[s][i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[s][i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);              // this thread's 4 output channels: (tid % (16 * NCO)) * 4

    // raw pieces: 4 fp32 (float4) or 4 bf16 (uint2)
    typename std::conditional<X16, uint2, float4>::type ra[XA];
    typename std::conditional<DY16, uint2, float4>::type rb[XB];
    unsigned va = 0, vb = 0;
    // a thread's pieces sit at the same (row in tile, channel) for every tile: element offsets relative to the tile's first row are
    // computed once; per tile only the (uniform) tile base and the row-tail mask change
    unsigned offA[XA], offB[XB]; unsigned cokA = 0, cokB = 0, isx1 = 0;
#pragma unroll
    for (int u = 0; u < XA; ++u) {
        const int i = tid + u * 256, rr = i >> 4, c = ci0 + (i & 15) * 4;
        if (c < Cin) cokA |= 1u << u;
        if (c >= P.C0) { isx1 |= 1u << u; offA[u] = (unsigned)(rr * P.C1 + (c - P.C0)); } else offA[u] = (unsigned)(rr * P.C0 + c);
    }
#pragma unroll
    for (int u = 0; u < XB; ++u) {
        const int i = tid + u * 256, rr = i / (16 * NCO), c = co0 + (i % (16 * NCO)) * 4;
        if (c < P.Cout) cokB |= 1u << u;
        offB[u] = (unsigned)(rr * P.Cout + c);
    }
    auto load_tile = [&](long t) {
        const long row0 = t * 64;
        const int left = (int)min((long)64, rows - row0);         // rows of this tile (uniform)
        const char* xb0 = reinterpret_cast<const char*>(P.x0) + (size_t)row0 * P.C0 * (X16 ? 2 : 4);
        const char* xb1 = reinterpret_cast<const char*>(P.x1) + (size_t)row0 * P.C1 * (X16 ? 2 : 4);
        const char* yb = reinterpret_cast<const char*>(P.dy) + (size_t)row0 * P.Cout * (DY16 ? 2 : 4);
        va = 0; vb = 0;
#pragma unroll
        for (int u = 0; u < XA; ++u) {
            const bool ok = ((tid + u * 256) >> 4) < left && ((cokA >> u) & 1u);
            const size_t o = ok ? (size_t)offA[u] : 0;
            if constexpr (X16) ra[u] = *reinterpret_cast<const uint2*>((ok && ((isx1 >> u) & 1u) ? xb1 : xb0) + o * 2);
            else ra[u] = *reinterpret_cast<const float4*>((ok && ((isx1 >> u) & 1u) ? xb1 : xb0) + o * 4);
            va |= (ok ? 1u : 0u) << u;
        }
#pragma unroll
        for (int u = 0; u < XB; ++u) {
            const bool ok = ((tid + u * 256) / (16 * NCO)) < left && ((cokB >> u) & 1u);
            const size_t o = ok ? (size_t)offB[u] : 0;
            if constexpr (DY16) rb[u] = *reinterpret_cast<const uint2*>(yb + o * 2);
            else rb[u] = *reinterpret_cast<const float4*>(yb + o * 4);
            vb |= (ok ? 1u : 0u) << u;
        }
    };
    auto widen = [](const uint2 v) {
        const unsigned a = v.x, b = v.y;
        return make_float4(__uint_as_float(a << 16), __uint_as_float(a & 0xFFFF0000u), __uint_as_float(b << 16), __uint_as_float(b & 0xFFFF0000u));
    };
    const long ntiles = (rows + 63) / 64;
    const long t_begin = (long)blockIdx.x * tiles_per_wg, t_end = min(ntiles, t_begin + tiles_per_wg);
    if (t_begin < t_end) load_tile(t_begin);
    for (long t = t_begin; t < t_end; ++t) {
        __syncthreads();                                          // previous tile fully consumed
#pragma unroll
        for (int u = 0; u < XA; ++u) {
            const int i = tid + u * 256;
            float4 v;
            if constexpr (X16) v = widen(ra[u]); else v = ra[u];
            if (!((va >> u) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<uint2*>(As + (i >> 4) * RSA + (i & 15) * 8) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
        }
#pragma unroll
        for (int u = 0; u < XB; ++u) {
            const int i = tid + u * 256;
            float4 v;
            if constexpr (DY16) v = widen(rb[u]); else v = rb[u];
            if (!((vb >> u) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            bias4.x += v.x; bias4.y += v.y; bias4.z += v.z; bias4.w += v.w;
            uint2 o;
            if constexpr (DY16) o = ((vb >> u) & 1u) ? rb[u] : make_uint2(0u, 0u);
            else o = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
            *reinterpret_cast<uint2*>(Bs + (i / (16 * NCO)) * RSBB + (i % (16 * NCO)) * 8) = o;
        }
        __syncthreads();
        if (t + 1 < t_end) load_tile(t + 1);
#pragma unroll 1
        for (int k0 = 0; k0 < 64; k0 += 32) {
            const int p0 = k0 + 8 * q + qr, p1 = p0 + 4;
            const char* a0 = As + p0 * RSA + wi * 64 + pcz * 8;
            const char* a1 = As + p1 * RSA + wi * 64 + pcz * 8;
            const bf16x8 af0 = tr_frag(a0, a1), af1 = tr_frag(a0 + 32, a1 + 32);
#pragma unroll
            for (int s = 0; s < NCO; ++s) {
                const char* b0 = Bs + p0 * RSBB + s * 128 + wo * 64 + pcz * 8;
                const char* b1 = Bs + p1 * RSBB + s * 128 + wo * 64 + pcz * 8;
                const bf16x8 bf0 = tr_frag(b0, b1), bf1 = tr_frag(b0 + 32, b1 + 32);
                acc[s][0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0, bf0, acc[s][0][0], 0, 0, 0);
                acc[s][0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0, bf1, acc[s][0][1], 0, 0, 0);
                acc[s][1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1, bf0, acc[s][1][0], 0, 0, 0);
                acc[s][1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1, bf1, acc[s][1][1], 0, 0, 0);
            }
        }
    }
    auto target = [&](int co, float*& dWt, float*& dbt, int& cbase, int& cw) {
        const int sel = P.split ? co / P.split : 0;
        cbase = P.split ? sel * P.split : 0; cw = P.split ? P.split : P.Cout;
        dWt = sel == 0 ? P.dW : (sel == 1 ? P.dW1 : P.dW2);
        dbt = sel == 0 ? P.db : (sel == 1 ? P.db1 : P.db2);
    };
    float* const pslot = P.part ? P.part + (size_t)blockIdx.x * P.part_E : nullptr;       // deterministic mode: this K chunk's slot
    if (do_bias) {                                                // (uniform per workgroup)
        const int pc = tid % (16 * NCO);
        if (pslot) {
            __syncthreads();                                      // (the staging areas are free: fixed-order sum through them)
            float4* bred = reinterpret_cast<float4*>(Bs);
            bred[tid] = bias4;
            __syncthreads();
            for (int i = tid; i < 64 * NCO; i += 256) {
                const int co = co0 + i;
                float t = 0.f;
                for (int k = 0; k < 256 / (16 * NCO); ++k) { const float4 v = bred[k * 16 * NCO + (i >> 2)]; t += (i & 3) == 0 ? v.x : (i & 3) == 1 ? v.y : (i & 3) == 2 ? v.z : v.w; }
                if (co < P.Cout) P.part_b[(size_t)blockIdx.x * P.Cout + co] = t;
            }
        } else {
            atomicAdd(&bsum[pc * 4 + 0], bias4.x); atomicAdd(&bsum[pc * 4 + 1], bias4.y);
            atomicAdd(&bsum[pc * 4 + 2], bias4.z); atomicAdd(&bsum[pc * 4 + 3], bias4.w);
            __syncthreads();
            for (int i = tid; i < 64 * NCO; i += 256) {
                const int co = co0 + i;
                if (co < P.Cout) {
                    float* dWt; float* dbt; int cbase, cw;
                    target(co, dWt, dbt, cbase, cw);
                    if (dbt) atomicAdd(dbt + co - cbase, bsum[i]);
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NCO; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = co0 + s * 64 + wo * 32 + j * 16 + r;
            if (co >= P.Cout) continue;
            float* dWt; float* dbt; int cbase, cw;
            target(co, dWt, dbt, cbase, cw);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ci = ci0 + wi * 32 + i * 16 + 4 * q + e;
                    if (ci < Cin) {
                        if (pslot) pslot[(size_t)ci * P.Cout + co] = acc[s][i][j][e];
                        else atomicAdd(dWt + (size_t)ci * cw + co - cbase, acc[s][i][j][e]);
                    }
                }
        }
}

// deterministic mode: dst (+= the split targets) = sum over the slots, in slot order; one thread per element.  Slot k of element e is
// part[k * slot_stride + e]
__global__ __launch_bounds__(256) void slot_sum_kernel(const float* __restrict__ part, int nslots, size_t slot_stride, long E, int Cout, int split,
                                                       float* __restrict__ d0, float* __restrict__ d1, float* __restrict__ d2) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < E; e += (long)gridDim.x * 256) {
        float t = 0.f;
        // 8 slots in flight per thread (the slots were written by other CUs a moment ago: every load is a trip to the memory side)
        int k = 0;
        for (; k + 8 <= nslots; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + u) * slot_stride + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; k < nslots; ++k) t += part[(size_t)k * slot_stride + e];
        if (split) {
            const long row = e / Cout; const int co = (int)(e - row * Cout), sel = co / split;
            float* d = sel == 0 ? d0 : (sel == 1 ? d1 : d2);
            if (d) d[row * split + co - sel * split] += t;
        } else if (d0) d0[e] += t;
    }
}
// many slots (a 3x3 weight gradient of the wide tensors: 256 slots; bias rows): 16 lanes per element, lane j adds slots j, j + 16, ... in
// order, then one thread adds the 16 lane sums in lane order -- still a fixed order, 16 x the parallelism of the plain form and a sixteenth
// of its dependent load rounds (the plain form took 12 us for a 64-element bias row and 23 us for a 64 x 64 x 9 tile)
__global__ __launch_bounds__(256) void slot_sum16_kernel(const float* __restrict__ part, int nslots, size_t slot_stride, long E, int Cout, int split,
                                                         float* __restrict__ d0, float* __restrict__ d1, float* __restrict__ d2) {
    __shared__ float red[256];
    const int tid = threadIdx.x, el = tid & 15, j = tid >> 4;
    const long e = (long)blockIdx.x * 16 + el;
    float t = 0.f;
    if (e < E) {
        int k = j;
        for (; k + 7 * 16 < nslots; k += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + 16 * u) * slot_stride + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; k < nslots; k += 16) t += part[(size_t)k * slot_stride + e];
    }
    red[tid] = t;
    __syncthreads();
    if (j == 0 && e < E) {
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) sum += red[u * 16 + el];
        if (split) {
            const long row = e / Cout; const int co = (int)(e - row * Cout), sel = co / split;
            float* d = sel == 0 ? d0 : (sel == 1 ? d1 : d2);
            if (d) d[row * split + co - sel * split] += sum;
        } else if (d0) d0[e] += sum;
    }
}
hipError_t launch_slot_sum(const float* part, int nslots, size_t slot_stride, long E, int Cout, int split, float* d0, float* d1, float* d2, hipStream_t st) {
    if (nslots >= 32 && (E + 15) / 16 <= 65535 * 16) {
        hipLaunchKernelGGL(slot_sum16_kernel, dim3((unsigned)((E + 15) / 16)), dim3(256), 0, st, part, nslots, slot_stride, E, Cout, split, d0, d1, d2);
        return hipGetLastError();
    }
    const int blocks = (int)std::max<long>(1, std::min<long>((E + 255) / 256, 2048));
    hipLaunchKernelGGL(slot_sum_kernel, dim3(blocks), dim3(256), 0, st, part, nslots, slot_stride, E, Cout, split, d0, d1, d2);
    return hipGetLastError();
}
static hipError_t launch_wgrad_finalize(const float* part, int nslots, long E, int Cout, int split, float* d0, float* d1, float* d2, hipStream_t st) {
    return launch_slot_sum(part, nslots, (size_t)E, E, Cout, split, d0, d1, d2, st);
}
// slots a launch needs: `chunks` dW tiles of E floats + `chunks` bias rows
static bool wgrad_det_setup(WgradArgs& a, long chunks, long E) {
    a.part_E = E; a.part_b = nullptr;
    if (!a.part) return false;
    const size_t need = (size_t)chunks * ((size_t)E + (size_t)a.Cout);
    if (need > a.part_cap) { a.part = nullptr; return false; }
    a.part_b = a.part + (size_t)chunks * E;
    return true;
}
static hipError_t wgrad_det_finish(const WgradArgs& a, long chunks, hipStream_t st) {
    hipError_t e = launch_wgrad_finalize(a.part, (int)chunks, a.part_E, a.Cout, a.split, a.dW, a.dW1, a.dW2, st);
    if (e != hipSuccess || !a.db) return e;
    return launch_wgrad_finalize(a.part_b, (int)chunks, a.Cout, a.Cout, a.split, a.db, a.db1, a.db2, st);
}

template <bool X16, bool DY16, int NCO>
static hipError_t launch_wgrad1x1_t(const WgradArgs& a0, long target_wgs, hipStream_t st) {
    WgradArgs a = a0;
    const long rows = (long)a.NF * a.H * a.W;
    const int Cin = a.C0 + a.C1;
    const int ci_tiles = (Cin + 63) / 64, co_tiles = (a.Cout + 64 * NCO - 1) / (64 * NCO);
    const long ntiles = (rows + 63) / 64;
    const long chunks = std::max<long>(1, std::min<long>(ntiles, target_wgs / ((long)ci_tiles * co_tiles)));
    const int per = (int)((ntiles + chunks - 1) / chunks);
    const long nslots = (ntiles + per - 1) / per;
    const size_t lds = 64 * WG_RSB + 64 * (NCO * 128 + 16) + 64 * NCO * 4;
    const bool det = wgrad_det_setup(a, nslots, (long)Cin * a.Cout);
    hipLaunchKernelGGL((wgrad1x1_kernel<X16, DY16, NCO>), dim3((unsigned)nslots, ci_tiles, co_tiles), dim3(256), lds, st, a, rows, per);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !det) return e;
    return wgrad_det_finish(a, nslots, st);
}

// out[c] += sum over rows of x[row][c]   (bias / LayerNorm-beta gradients); x is [rows][C] fp32
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long rows, int C, float* __restrict__ part) {
    __shared__ float red[256];
    const int cpt = (C + 3) / 4;                              // float4 columns
    const int lanes_c = cpt < 256 ? cpt : 256;
    const int rl = 256 / lanes_c;                             // row lanes per workgroup
    const int cq = threadIdx.x % lanes_c, rq = threadIdx.x / lanes_c;
    for (int cb = cq; cb < cpt; cb += lanes_c) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rq < rl)
            for (long row = (long)blockIdx.x * rl + rq; row < rows; row += (long)gridDim.x * rl) {
                const float4 v = *reinterpret_cast<const float4*>(x + row * C + cb * 4);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        const float vals[4] = {s.x, s.y, s.z, s.w};
        for (int e = 0; e < 4; ++e) {
            __syncthreads();
            red[threadIdx.x] = (rq < rl) ? vals[e] : 0.f;
            __syncthreads();
            if (rq == 0) {
                float t = 0.f;
                for (int k = 0; k < rl; ++k) t += red[k * lanes_c + cq];
                if (cb * 4 + e < C) { if (part) part[(size_t)blockIdx.x * C + cb * 4 + e] = t; else atomicAdd(out + cb * 4 + e, t); }
            }
        }
    }
}

// y[i] += x[i]
__global__ void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ x, long n) {
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
        if (i + 3 < n) {
            float4 a = *reinterpret_cast<float4*>(y + i);
            const float4 b = *reinterpret_cast<const float4*>(x + i);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            *reinterpret_cast<float4*>(y + i) = a;
        } else for (long k = i; k < n; ++k) y[k] += x[k];
    }
}

hipError_t launch_conv_wgrad(WgradArgs a, hipStream_t st) {
    const int Cin = a.C0 + a.C1;
    a.taps = a.kind ? 16 : a.kh * a.kw;
    if (a.kind == 0) {
        a.sa = a.stride; a.sb = 1; a.ext = a.kh; a.halo = (a.stride == 1) ? (a.kh - 1) / 2 : (a.kh - 2) / 2;
        a.Hm = (a.H + a.stride - 1) / a.stride; a.Wm = (a.W + a.stride - 1) / a.stride; a.Hy = a.Hm; a.Wy = a.Wm;
        if (a.stride == 1) { a.PH = 8; a.PW = 8; } else { a.PH = 4; a.PW = 8; }
    } else {
        a.sa = 1; a.sb = 2; a.ext = 3; a.halo = 1; a.Hm = a.H; a.Wm = a.W; a.Hy = 2 * a.H; a.Wy = 2 * a.W; a.PH = 4; a.PW = 8;
    }
    const int wide_patch = 1;
    if (a.bf16_mma && a.kind == 0 && a.stride == 1 && a.kh == 3 && wide_patch && a.W >= 16) { a.PH = 8; a.PW = 16; }   // bf16 form: 128 positions per patch
    const int NT = a.taps <= 9 ? (a.taps == 1 ? 1 : 9) : 8;
    const int tap_groups = (a.taps + NT - 1) / NT;
    a.co_tiles = (a.Cout + 63) / 64;
    const int ci_tiles = (Cin + 63) / 64;
    const int IH = (a.PH - 1) * a.sa + a.ext, IW = (a.PW - 1) * a.sa + a.ext;
    const size_t lds = ((size_t)IH * IW + (size_t)a.PH * a.sb * a.PW * a.sb) * WG_LD * 4 + 3 * 64 * 4 + 32 * 4;
    const long patches = (long)a.NF * ((a.Hm + a.PH - 1) / a.PH) * ((a.Wm + a.PW - 1) / a.PW);
    const long tiles = (long)ci_tiles * a.co_tiles * tap_groups;
    const long target_wgs = 1024;
    // multi-tap tiles: the epilogue is 64 x 64 x NT atomic adds per workgroup and the chip sustains about one 256-byte atomic wave
    // instruction per 50 ns per CU, so 1024 workgroups spend ~115 us in the epilogue alone; 256 (one per CU) measured best
    const long target_wgs9 = 256;
    long chunks = std::max<long>(1, std::min<long>(patches, (a.taps > 1 ? target_wgs9 : target_wgs) / std::max<long>(1, tiles)));
    dim3 grid((unsigned)chunks, ci_tiles, a.co_tiles * tap_groups);
    float* const part_in = a.part;                        // (the 1x1 GEMM form below sizes its own slots)
    const bool det = wgrad_det_setup(a, chunks, (long)a.taps * Cin * a.Cout);
    if (a.bf16_mma && (a.PW == 8 || a.PW == 16)) {
        a.pwl = a.PW == 8 ? 3 : 4;
        const size_t lds16 = ((size_t)IH * IW + (size_t)a.PH * a.sb * a.PW * a.sb) * WG_RSB + 4 * 64 * 4 + 32 * 4;
#define VDX_WG16(NT_, NG_, PF_, X16_, DY16_) do { auto kfn = conv_wgrad16_kernel<NT_, NG_, PF_, X16_, DY16_>;                                                \
        if (lds16 > 64 * 1024) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16); if (e != hipSuccess) return e; } \
        hipLaunchKernelGGL(kfn, grid, dim3(256 * NG_), lds16, st, a); } while (0)
        a.m_iw = (unsigned)((1ull << 32) / (unsigned)IW) + 1u;
        a.m_bw = (unsigned)((1ull << 32) / (unsigned)(a.PW * a.sb)) + 1u;
        {   // register staging capacity of conv_wgrad16_kernel (XA = 6, XB = 4 float4 pieces per thread)
            const long nth = NT == 1 ? 256 : 512;
            if ((long)IH * IW * 16 > 6 * nth || (long)a.PH * a.sb * a.PW * a.sb * 16 > 4 * nth) return hipErrorInvalidValue;
        }
        // projections: the split-K GEMM form (wide output tiles when Cout allows: x is staged Cout / 256 times instead of Cout / 64)
        const int use_1x1 = 2;   // 0: patch kernel, 1: GEMM form for wide outputs only, 2: always
        if (use_1x1 && NT == 1 && a.kind == 0 && a.stride == 1 && !a.pro && (a.Cout % 256 == 0 || use_1x1 == 2)) {
            const long wgs4 = 512;
            a.part = part_in;
            if (a.Cout % 256 == 0) {
                if (a.x0_bf16) return a.dy_bf16 ? launch_wgrad1x1_t<true, true, 4>(a, wgs4, st) : launch_wgrad1x1_t<true, false, 4>(a, wgs4, st);
                return a.dy_bf16 ? launch_wgrad1x1_t<false, true, 4>(a, wgs4, st) : launch_wgrad1x1_t<false, false, 4>(a, wgs4, st);
            }
            if (a.x0_bf16) return a.dy_bf16 ? launch_wgrad1x1_t<true, true, 1>(a, target_wgs, st) : launch_wgrad1x1_t<true, false, 1>(a, target_wgs, st);
            return a.dy_bf16 ? launch_wgrad1x1_t<false, true, 1>(a, target_wgs, st) : launch_wgrad1x1_t<false, false, 1>(a, target_wgs, st);
        }
        const bool pf1 = false;
#define VDX_WG16_IO(NT_, NG_, PF_) do { if (a.x0_bf16) { if (a.dy_bf16) VDX_WG16(NT_, NG_, PF_, true, true); else VDX_WG16(NT_, NG_, PF_, true, false); } \
                                        else { if (a.dy_bf16) VDX_WG16(NT_, NG_, PF_, false, true); else VDX_WG16(NT_, NG_, PF_, false, false); } } while (0)
        if (NT == 1) { if (pf1) VDX_WG16_IO(1, 1, true); else VDX_WG16_IO(1, 1, false); }
        else if (NT == 9) VDX_WG16_IO(9, 2, true);
        else VDX_WG16_IO(8, 2, true);
#undef VDX_WG16_IO
#undef VDX_WG16
        { hipError_t e = hipGetLastError(); if (e != hipSuccess || !det) return e; }
        return wgrad_det_finish(a, chunks, st);
    }
#define VDX_WG(NT_) do { auto kfn = conv_wgrad_kernel<NT_>;                                                           \
        if (lds > 64 * 1024) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; } \
        hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, a); } while (0)
    if (NT == 1) VDX_WG(1); else if (NT == 9) VDX_WG(9); else VDX_WG(8);
#undef VDX_WG
    { hipError_t e = hipGetLastError(); if (e != hipSuccess || !det) return e; }
    return wgrad_det_finish(a, chunks, st);
}

hipError_t launch_colsum(const float* x, float* out, long rows, int C, hipStream_t st, float* part, size_t part_cap) {
    const int cpt = (C + 3) / 4, lanes_c = cpt < 256 ? cpt : 256, rl = 256 / lanes_c;
    const int blocks = (int)std::max<long>(1, std::min<long>((rows + rl - 1) / rl, 1024));
    if (part && (size_t)blocks * C > part_cap) part = nullptr;
    hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, st, x, out, rows, C, part);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !part) return e;
    return launch_wgrad_finalize(part, blocks, C, C, 0, out, nullptr, nullptr, st);      // every workgroup wrote its row of column sums: add them in order
}

hipError_t launch_add_inplace(float* y, const float* x, long n, hipStream_t st) {
    const int blocks = (int)std::max<long>(1, std::min<long>((n / 4 + 255) / 256, 2048));
    hipLaunchKernelGGL(add_inplace_kernel, dim3(blocks), dim3(256), 0, st, y, x, n);
    return hipGetLastError();
}

}  // namespace vdx
