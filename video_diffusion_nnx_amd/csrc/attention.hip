// Fused multi-head self-attention block on MFMA (gfx950):  y = out_proj(softmax(q k^T / sqrt(d)) v) + x
//
// Replaces what XLA did for the reference's MultiheadAttention as used inside Unet3D:
//   modules.py:247-326 (q/k/v LinearGeneral + bias, q / sqrt(d), softmax, PV, out LinearGeneral + bias)
//   wrapped by EinopsToAndFrom + PreNorm + Residual (modules.py:21-27,47-60,132-148; unet3d.py:86-96,196-208):
//   PreNorm is a no-op and drops pos_bias / focus_present_mask, Residual adds x.
// One kernel serves the temporal attention ('b (h w) f c': L = F tokens strided by H*W*C) and the
// bottleneck spatial attention ('b f (h w) c': L = H*W contiguous tokens) through (inner, stride) args.
//
// A workgroup owns 64 token rows = (64/LP) sequences padded to LP in {16,32,64} tokens.  Per head:
//   GEMM1  qkv_h[96, 64] = Wqkv_h[96, C] . x^T     (x and W K-tiles staged in LDS, 3x2 tiles per wave)
//   core   one wave = one 16-query tile: S = K Q^T (keys on the accumulator rows), softmax over the
//          keys with wavefront shuffles, P -> LDS, O^T = V^T P^T
//   GEMM2  y[C, 64] += Wo[:, h*32:(h+1)*32] . O_h^T  (accumulated in registers across heads)
// so q/k/v/scores never touch HBM: traffic = read x twice (GEMM1 + residual) + write y.
#include "vdx_common.h"
#include <type_traits>
#include "vdx_internal.h"
#include <stdlib.h>
#include <algorithm>

namespace vdx {

template <int MODE, int LP, int TMO>
__global__ __launch_bounds__(256) void attention_kernel(const AttnArgs P) {
    using M = Mma<MODE>;
    constexpr int KT = M::KT, KC = M::KC, RS = ROW_STRIDE;
    constexpr int APIECES = KT / 4;
    constexpr int NSEQ = 64 / LP, QT = LP / 16;
    constexpr int NCHD = 32 / KC;                       // chunks covering d = 32      (f32 2, bf16 1)
    constexpr int NCHL = (LP + KC - 1) / KC;            // chunks covering LP keys
    constexpr int RSV = NCHL * 64 + 16;                 // row stride of the key-contiguous matrices
    constexpr int D = 32;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    long* rowoff = reinterpret_cast<long*>(smem);               // [64]
    char* xs = smem + 512;
    char* ws = xs + 64 * RS;
    char* qs = ws + 96 * RS;
    char* ks = qs + 64 * RS;
    char* os = ks + 64 * RS;
    char* vT = os + 64 * RS;                                    // [NSEQ*32][RSV]
    char* ps = vT + NSEQ * 32 * RSV;                            // [64][RSV]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lp = lane & 15, q = lane >> 4;
    const int wc = w & 1, wr = w >> 1;
    const int HD = P.heads * D;

    if (tid < 64) {
        const int sl = tid / LP, tok = tid % LP;
        const long sg = (long)blockIdx.x * NSEQ + sl;
        long off = -1;
        if (sg < P.nseq && tok < P.L) off = (sg / P.inner) * P.outer_stride + (sg % P.inner) * P.inner_stride + (long)tok * P.tok_stride;
        rowoff[tid] = off;
    }
    for (int i = tid; i < (NSEQ * 32 + 64) * RSV / 4; i += 256) reinterpret_cast<float*>(vT)[i] = 0.f;   // zero K padding
    __syncthreads();

    f32x4 oacc[TMO][4];
#pragma unroll
    for (int i = 0; i < TMO; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const char* wq = reinterpret_cast<const char*>(P.wqkv);
    const char* wo = reinterpret_cast<const char*>(P.wo);
    const int nkt = P.CPad / KT;

    // x tile: resident across the 8 heads when C fits one K tile; else re-staged per head from L2
    const bool x_resident = (nkt == 1);
    auto stage_x = [&](int kt) {
        for (int i = tid; i < 64 * APIECES; i += 256) {
            const int row = i / APIECES, pc = i % APIECES;
            const int c = kt * KT + pc * 4;
            const long ro = rowoff[row];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ro >= 0 && c < P.C) v = load4_f32_or_bf16(P.x, (size_t)(ro + c), P.io_bf16);
            M::store4(xs + row * RS, pc * 4, v);
        }
    };
    // per-head weight tile (96 rows x 128 B) prefetched through registers one K tile ahead
    // three named registers (an indexed array here ends up in scratch memory)
    uint4 wp0 = make_uint4(0, 0, 0, 0), wp1 = wp0, wp2 = wp0;
    const int wrow_ = tid >> 3, wpc_ = tid & 7;                       // piece j covers weight-tile row wrow_ + 32 j  (part j = q,k,v)
    auto wfetch = [&](int h, int kt) {
        const size_t base = ((size_t)(h * D + wrow_) * P.CPad + (size_t)kt * KT) * M::ES + wpc_ * 16;
        const size_t part = (size_t)HD * P.CPad * M::ES;
        wp0 = *reinterpret_cast<const uint4*>(wq + base);
        wp1 = *reinterpret_cast<const uint4*>(wq + base + part);
        wp2 = *reinterpret_cast<const uint4*>(wq + base + 2 * part);
    };
    auto wput = [&]() {
        char* dst = ws + wrow_ * RS + wpc_ * 16;
        *reinterpret_cast<uint4*>(dst) = wp0;
        *reinterpret_cast<uint4*>(dst + 32 * RS) = wp1;
        *reinterpret_cast<uint4*>(dst + 64 * RS) = wp2;
    };
    if (x_resident) stage_x(0);
    wfetch(0, 0);

    for (int h = 0; h < P.heads; ++h) {
        // ---------------- GEMM1: q,k,v of head h for the 64 rows ----------------
        f32x4 acc[3][2];
#pragma unroll
        for (int i = 0; i < 3; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int kt = 0; kt < nkt; ++kt) {
            __syncthreads();                              // previous readers of xs / ws are done
            if (!x_resident) stage_x(kt);
            wput();
            __syncthreads();
            {
                int nh = h, nk = kt + 1;
                if (nk == nkt) { nk = 0; nh = h + 1; }
                if (nh < P.heads) wfetch(nh, nk);
            }
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                uint4 bf[2], af[3];
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) bf[tn] = *reinterpret_cast<const uint4*>(xs + ((wr * 2 + tn) * 16 + lp) * RS + ch * 64 + q * 16);
#pragma unroll
                for (int tm = 0; tm < 3; ++tm) af[tm] = *reinterpret_cast<const uint4*>(ws + ((wc * 3 + tm) * 16 + lp) * RS + ch * 64 + q * 16);
#pragma unroll
                for (int tm = 0; tm < 3; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) M::mma(acc[tm][tn], af[tm], bf[tn]);
            }
        }
        // epilogue of GEMM1: +bias, q * scale; q,k row-major [row][d], v transposed [seq][d][key]
#pragma unroll
        for (int tm = 0; tm < 3; ++tm) {
            const int ct = wc * 3 + tm, part = ct >> 1, d0 = (ct & 1) * 16 + 4 * q;
            const float4 bias = *reinterpret_cast<const float4*>(P.bqkv + part * HD + h * D + d0);
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int row = (wr * 2 + tn) * 16 + lp;
                float4 v = make_float4(acc[tm][tn][0] + bias.x, acc[tm][tn][1] + bias.y, acc[tm][tn][2] + bias.z, acc[tm][tn][3] + bias.w);
                if (part == 0) {
                    v.x *= P.scale; v.y *= P.scale; v.z *= P.scale; v.w *= P.scale;
                    M::store4(qs + row * RS, d0, v);
                } else if (part == 1) {
                    M::store4(ks + row * RS, d0, v);
                } else {
                    const int sl = row / LP, j = row % LP;
                    char* base = vT + (sl * 32 + d0) * RSV;
                    M::store1(base, j, v.x); M::store1(base + RSV, j, v.y);
                    M::store1(base + 2 * RSV, j, v.z); M::store1(base + 3 * RSV, j, v.w);
                }
            }
        }
        __syncthreads();
        // ---------------- core: wave w = 16-query tile qt of sequence sl ----------------
        {
            const int sl = w / QT, qt = w % QT;
            const int qrow = sl * LP + qt * 16 + lp;
            f32x4 s[QT];
#pragma unroll
            for (int jt = 0; jt < QT; ++jt) {
                s[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ch = 0; ch < NCHD; ++ch) {
                    const uint4 a = *reinterpret_cast<const uint4*>(ks + (sl * LP + jt * 16 + lp) * RS + ch * 64 + q * 16);
                    const uint4 bq = *reinterpret_cast<const uint4*>(qs + qrow * RS + ch * 64 + q * 16);
                    M::mma(s[jt], a, bq);
                }
            }
            float mx = -1e30f;
#pragma unroll
            for (int jt = 0; jt < QT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (jt * 16 + 4 * q + r >= P.L) s[jt][r] = -1e30f;
                    mx = fmaxf(mx, s[jt][r]);
                }
            mx = max_q(mx);
            float sum = 0.f;
#pragma unroll
            for (int jt = 0; jt < QT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[jt][r] = __expf(s[jt][r] - mx); sum += s[jt][r]; }
            sum = reduce_q(sum);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int jt = 0; jt < QT; ++jt)
                M::store4(ps + qrow * RSV, jt * 16 + 4 * q, make_float4(s[jt][0] * inv, s[jt][1] * inv, s[jt][2] * inv, s[jt][3] * inv));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // P rows are private to this wave: LDS ops of one wave stay in order
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ch = 0; ch < NCHL; ++ch) {
                    const uint4 a = *reinterpret_cast<const uint4*>(vT + (sl * 32 + dt * 16 + lp) * RSV + ch * 64 + q * 16);
                    const uint4 bp = *reinterpret_cast<const uint4*>(ps + qrow * RSV + ch * 64 + q * 16);
                    M::mma(o, a, bp);
                }
                M::store4(os + qrow * RS, dt * 16 + 4 * q, make_float4(o[0], o[1], o[2], o[3]));
            }
        }
        __syncthreads();
        // ---------------- GEMM2 partial: oacc += Wo[:, head h] . O_h^T ----------------
#pragma unroll
        for (int ch = 0; ch < NCHD; ++ch) {
            uint4 bf[4];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) bf[tn] = *reinterpret_cast<const uint4*>(os + (tn * 16 + lp) * RS + ch * 64 + q * 16);
#pragma unroll
            for (int tmo = 0; tmo < TMO; ++tmo) {
                const int co = (w * TMO + tmo) * 16 + lp;
                uint4 a = make_uint4(0, 0, 0, 0);
                if (co < P.C) a = *reinterpret_cast<const uint4*>(wo + ((size_t)co * P.HDPad + (size_t)h * D) * M::ES + ch * 64 + q * 16);
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) M::mma(oacc[tmo][tn], a, bf[tn]);
            }
        }
    }
    // ---------------- epilogue: + bias + residual ----------------
#pragma unroll
    for (int tmo = 0; tmo < TMO; ++tmo) {
        const int co = (w * TMO + tmo) * 16 + 4 * q;
        if (co >= P.C) continue;
        const float4 bo = *reinterpret_cast<const float4*>(P.bo + co);
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const long ro = rowoff[tn * 16 + lp];
            if (ro < 0) continue;
            const float4 xr = load4_f32_or_bf16(P.x, (size_t)(ro + co), P.io_bf16);
            float4 v;
            v.x = oacc[tmo][tn][0] + bo.x + xr.x; v.y = oacc[tmo][tn][1] + bo.y + xr.y;
            v.z = oacc[tmo][tn][2] + bo.z + xr.z; v.w = oacc[tmo][tn][3] + bo.w + xr.w;
            store4_f32_or_bf16(P.y, (size_t)(ro + co), v, P.io_bf16);
        }
    }
}


// Register-resident variant for sequences of <= 16 tokens (the temporal attention of every shipped config).
// Wave w owns sequence w of the workgroup (16 token rows) end to end.  The 16x16 accumulator layout (lane (c, q) holds
// rows 4q..4q+3) is exactly the K-slot layout of a K = 16 MFMA operand, so
//   S^T[j,i]  = sum_d K[j,d] Q[i,d]      : operands = the k / q accumulator tiles of GEMM1 (two d tiles)
//   O^T[d,i]  = sum_j V^T[d,j] P^T[j,i]  : A = the v accumulator (GEMM1 run with swapped operands: tokens on the rows),
//                                          B = the normalised scores, still in their accumulator registers
//   y[c,i]   += sum_d Wo[c, h*32+d] O^T[d,i] : B = the O^T accumulator tiles, A = 4-element weight fragments from L2
// never leave the register file: no q/k/v/P/O round trips through LDS, two workgroup barriers per head (weight tile only).
template <int MODE, int TMA, bool F8>   // TMA = C / 16 output-channel tiles (all owned by every wave, for its 16 rows); F8: fp8 QK^T / PV
__global__ __launch_bounds__(256) void attention_reg_kernel(const AttnArgs P) {
    using M = Mma<MODE>;
    constexpr int KT = M::KT, RS = ROW_STRIDE;
    constexpr int APIECES = KT / 4;
    constexpr int D = 32;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    long* rowoff = reinterpret_cast<long*>(smem);               // [64]
    char* xs = smem + 512;                                      // [64][RS]
    char* ws = xs + 64 * RS;                                    // [96][RS]
    constexpr int WOS = D * M::ES + 16;                         // Wo[:, head] slice rows: 32 k-elements + pad (conflict-free 8/16-byte reads)
    constexpr int PPR = D * M::ES / 16;                         // 16-byte pieces per slice row
    char* wos = ws + 96 * RS;                                   // [C][WOS]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lp = lane & 15, q = lane >> 4;
    const int HD = P.heads * D;
    if (tid < 64) {
        const int sl = tid >> 4, tok = tid & 15;
        const long sg = (long)blockIdx.x * 4 + sl;
        long off = -1;
        if (sg < P.nseq && tok < P.L) off = (sg / P.inner) * P.outer_stride + (sg % P.inner) * P.inner_stride + (long)tok * P.tok_stride;
        rowoff[tid] = off;
    }
    __syncthreads();

    f32x4 oacc[TMA];
#pragma unroll
    for (int i = 0; i < TMA; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* wq = reinterpret_cast<const char*>(P.wqkv);
    const char* wo = reinterpret_cast<const char*>(P.wo);
    const int nkt = P.CPad / KT;
    const bool x_resident = (nkt == 1);
    auto stage_x = [&](int kt) {
        for (int i = tid; i < 64 * APIECES; i += 256) {
            const int row = i / APIECES, pc = i % APIECES;
            const int c = kt * KT + pc * 4;
            const long ro = rowoff[row];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ro >= 0 && c < P.C) v = load4_f32_or_bf16(P.x, (size_t)(ro + c), P.io_bf16);
            M::store4(xs + row * RS, pc * 4, v);
        }
    };
    // three named registers (an indexed array here ends up in scratch memory)
    uint4 wp0 = make_uint4(0, 0, 0, 0), wp1 = wp0, wp2 = wp0;
    const int wrow_ = tid >> 3, wpc_ = tid & 7;                       // piece j covers weight-tile row wrow_ + 32 j  (part j = q,k,v)
    auto wfetch = [&](int h, int kt) {
        const size_t base = ((size_t)(h * D + wrow_) * P.CPad + (size_t)kt * KT) * M::ES + wpc_ * 16;
        const size_t part = (size_t)HD * P.CPad * M::ES;
        wp0 = *reinterpret_cast<const uint4*>(wq + base);
        wp1 = *reinterpret_cast<const uint4*>(wq + base + part);
        wp2 = *reinterpret_cast<const uint4*>(wq + base + 2 * part);
    };
    auto wput = [&]() {
        char* dst = ws + wrow_ * RS + wpc_ * 16;
        *reinterpret_cast<uint4*>(dst) = wp0;
        *reinterpret_cast<uint4*>(dst + 32 * RS) = wp1;
        *reinterpret_cast<uint4*>(dst + 64 * RS) = wp2;
    };
    if (x_resident) stage_x(0);
    wfetch(0, 0);

    for (int h = 0; h < P.heads; ++h) {
        // GEMM1 for this wave's 16 rows: q, k as [d rows][token cols]; v as [token rows][d cols] (operands swapped)
        f32x4 aq[2], ak[2], av[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { aq[i] = ak[i] = av[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int kt = 0; kt < nkt; ++kt) {
            __syncthreads();
            if (!x_resident) stage_x(kt);
            wput();
            if (kt == 0)                                  // Wo[:, head h]: read by every wave below (previous head's readers passed the barrier above)
                for (int i = tid; i < TMA * 16 * PPR; i += 256) {
                    const int row = i / PPR, pc = i % PPR;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (row < P.C) v = *reinterpret_cast<const uint4*>(wo + ((size_t)row * P.HDPad + (size_t)h * D) * M::ES + pc * 16);
                    *reinterpret_cast<uint4*>(wos + row * WOS + pc * 16) = v;
                }
            __syncthreads();
            {
                int nh = h, nk = kt + 1;
                if (nk == nkt) { nk = 0; nh = h + 1; }
                if (nh < P.heads) wfetch(nh, nk);
            }
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const uint4 xf = *reinterpret_cast<const uint4*>(xs + (w * 16 + lp) * RS + ch * 64 + q * 16);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const uint4 wqf = *reinterpret_cast<const uint4*>(ws + ((0 * 2 + t) * 16 + lp) * RS + ch * 64 + q * 16);
                    const uint4 wkf = *reinterpret_cast<const uint4*>(ws + ((1 * 2 + t) * 16 + lp) * RS + ch * 64 + q * 16);
                    const uint4 wvf = *reinterpret_cast<const uint4*>(ws + ((2 * 2 + t) * 16 + lp) * RS + ch * 64 + q * 16);
                    M::mma(aq[t], wqf, xf);
                    M::mma(ak[t], wkf, xf);
                    M::mma(av[t], xf, wvf);                   // swapped: rows = tokens, cols = d
                }
            }
        }
        // biases (q, k: per row d = 4q+reg ; v: per column d = lp), q scale
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float4 bq = *reinterpret_cast<const float4*>(P.bqkv + h * D + t * 16 + 4 * q);
            const float4 bk = *reinterpret_cast<const float4*>(P.bqkv + HD + h * D + t * 16 + 4 * q);
            const float bv = P.bqkv[2 * HD + h * D + t * 16 + lp];
            aq[t][0] = (aq[t][0] + bq.x) * P.scale; aq[t][1] = (aq[t][1] + bq.y) * P.scale;
            aq[t][2] = (aq[t][2] + bq.z) * P.scale; aq[t][3] = (aq[t][3] + bq.w) * P.scale;
            ak[t][0] += bk.x; ak[t][1] += bk.y; ak[t][2] += bk.z; ak[t][3] += bk.w;
            av[t][0] += bv; av[t][1] += bv; av[t][2] += bv; av[t][3] += bv;
        }
        // S^T[j, i]: lane (i, q) holds keys j = 4q..4q+3
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        core_mma16<M, F8>(s, ak[0], aq[0]);
        core_mma16<M, F8>(s, ak[1], aq[1]);
        float mx = -1e30f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { if (4 * q + r >= P.L) s[r] = -1e30f; mx = fmaxf(mx, s[r]); }
        mx = max_q(mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[r] = __expf(s[r] - mx); sum += s[r]; }
        sum = reduce_q(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] *= inv;
        // O^T[d, i] per d tile; then y += Wo[:, head h] . O
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
            core_mma16<M, F8>(o, av[t], s);
#pragma unroll
            for (int tm = 0; tm < TMA; ++tm) {
                const f32x4 a = M::load_w4(wos + (tm * 16 + lp) * WOS + (t * 16 + 4 * q) * M::ES);
                M::mma16(oacc[tm], a, o);
            }
        }
    }
    // epilogue: + bias + residual, rows of this wave's sequence
    const long ro = rowoff[w * 16 + lp];
#pragma unroll
    for (int tm = 0; tm < TMA; ++tm) {
        const int co = tm * 16 + 4 * q;
        if (co >= P.C || ro < 0) continue;
        const float4 bo = *reinterpret_cast<const float4*>(P.bo + co);
        float4 xr = make_float4(0.f, 0.f, 0.f, 0.f);
        xr = load4_f32_or_bf16(P.x, (size_t)(ro + co), P.io_bf16);
        float4 v;
        v.x = oacc[tm][0] + bo.x + xr.x; v.y = oacc[tm][1] + bo.y + xr.y;
        v.z = oacc[tm][2] + bo.z + xr.z; v.w = oacc[tm][3] + bo.w + xr.w;
        store4_f32_or_bf16(P.y, (size_t)(ro + co), v, P.io_bf16);
    }
}

// One wave per head (heads == 8, sequences of <= 16 tokens, C small enough for register-resident weights): the layout
// used for the temporal attention of the large levels.  A workgroup walks `nsub` sub-tiles of 4 sequences; wave h keeps
// head h's q/k/v projection rows in registers as MFMA fragments for the whole walk, the fp32 x sub-tile is fetched once
// for all heads (registers one sub-tile ahead -> LDS ring of 2), scores / softmax / PV stay in registers exactly as in
// attention_reg_kernel, the heads meet in LDS (os[64 rows][256]) and the out-projection is split over the 8 waves by
// (output-channel tile, sequence).  Global I/O is whole 16-byte pieces in (token, sequence, channel) order: one wave
// instruction covers the contiguous run of 4 adjacent sequences, the output tile goes through LDS (ys) to be stored the
// same way, and the fetched fp32 tile stays in registers as the residual.  Two barriers per sub-tile.
// The kernel is vector-issue bound, so the per-sequence VALU work is kept minimal: biases are the accumulators' initial
// values, 1/sqrt(d) is folded into the exponent, the key mask exists only for L < 16, cross-row reductions are permlane
// swaps, and every address is (uniform base of the sub-tile) + (per-thread constant).  Host guarantees inner % 4 == 0 and
// nseq % 4 == 0 (the 4 sequences of a sub-tile share their outer index) and 32-bit per-thread offsets.
// IO16: x and y are bf16 tensors (bf16 activation storage): pieces are 8 channels = 16 bytes, copied into the bf16 LDS tile
// as they are, and the raw piece is the residual.
// FULL: sequences of exactly 16 tokens and C == the tile's channels -- no key mask, every piece valid (compile-time: the kernel is
// vector-issue bound and the mask / validity selects are ~10 % of its VALU instructions)
template <int MODE, int NKT, int TMO, int TNO, bool IO16, bool F8, bool FULL = false>
__global__ __launch_bounds__(512) void attention_h8_kernel(const AttnArgs P, const int nsub) {
    using M = Mma<MODE>;
    static_assert(!IO16 || MODE == MODE_BF16, "bf16 activation storage implies bf16 MFMA operands");
    constexpr int KT = M::KT, KC = M::KC, RS = ROW_STRIDE, D = 32, HD = 256;
    constexpr int PCH = IO16 ? 8 : 4;                                 // channels per 16-byte piece
    constexpr int APIECES = KT / PCH;
    constexpr int XP = 64 * APIECES * NKT / 512;
    constexpr int PLANE = 64 * RS, BUF = NKT * PLANE;
    constexpr int RSO = HD * M::ES + 16;
    constexpr int NCHO = HD / KC;
    constexpr int CT = NKT * KT;                                      // channels held by the tile (== C)
    constexpr int RSY = CT * 4 + 16;
    // one-barrier pipeline with double-buffered os / ys where the LDS allows it: bf16 operands, C <= 64 (the level-0 kernels: 123 KB)
    constexpr bool PIPE = MODE == MODE_BF16 && NKT == 1;
    // 2-byte operands: the 32 columns of a head's block of os are stored in the order the PV accumulators hold them -- lane (token, q) owns
    // d = 4q..4q+3 (t = 0) and 16+4q..16+4q+3 (t = 1) = K slots 8q..8q+7 of the head's chunk -- so the lane writes ONE 16-byte piece
    // (ds_write_b128 over 8-lane groups at a 132-dword row stride: conflict-free) instead of two 8-byte pieces whose 16 rows collide two
    // by two (39 % of the kernel's LDS cycles were those conflicts: profiles/r02_pmc_step.md).  The out-projection's weight fragments
    // are loaded with the same permutation of K, so the product is unchanged.
#ifndef VDX_H8_DIAG
#define VDX_H8_DIAG 0        // knock-out switches for timing experiments (tools/mkvariant.sh); none in the product build
#endif
#ifndef VDX_H8_OSP
#define VDX_H8_OSP 2
#endif
    constexpr bool OSP = M::ES == 2 && VDX_H8_OSP != 0 && (VDX_H8_OSP == 1 || PIPE);
    static_assert(!OSP || MODE == MODE_BF16, "the packed os store writes bf16");
    // phase A on SB sequences at a time, stage by stage (projections of all SB, scores, softmax, PV): SB independent dependency chains
    // in flight instead of one (MFMA result -> cvt -> MFMA -> max -> permlane -> exp -> ... is ~30 dependent steps per sequence)
#ifndef VDX_H8_SB
#define VDX_H8_SB 4
#endif
    constexpr int SB = PIPE ? VDX_H8_SB : 1;

    extern __shared__ __attribute__((aligned(16))) char smem[];      // xs[2][NKT][64][RS] | os[64][RSO] | ys[64][RSY] (fp32); PIPE: os, ys twice
    char* os = smem + 2 * BUF;
    char* ys = os + 64 * RSO;
    const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
    const int lp = lane & 15, q = lane >> 4;

    // per-thread constants of its XP pieces: global element offset inside the sub-tile, LDS offsets (x tile, y tile)
    unsigned goff[XP];
    int xoff[XP], yoff[XP];
    bool pvalid[XP];
#pragma unroll
    for (int u = 0; u < XP; ++u) {
        const int i = tid + 512 * u;
        const int kt = i / (64 * APIECES), rem = i % (64 * APIECES);
        const int ridx = rem / APIECES, pc = rem % APIECES;          // ridx = token * 4 + sequence: 4 adjacent sequences =
        const int sl = ridx & 3, tok = ridx >> 2, c = kt * KT + pc * PCH; // one contiguous run per token
        goff[u] = (unsigned)(sl * P.inner_stride + tok * P.tok_stride + c);
        xoff[u] = kt * PLANE + (sl * 16 + tok) * RS + pc * PCH * M::ES; // LDS row = sequence * 16 + token
        yoff[u] = (sl * 16 + tok) * RSY + c * 4;
        pvalid[u] = FULL || (tok < P.L && c < P.C);
    }
    auto tile_base = [&](long sg0) __attribute__((always_inline)) -> long {                         // workgroup-uniform: scalar unit
        const unsigned inner = (unsigned)P.inner;
        return (long)((unsigned)sg0 / inner) * P.outer_stride + (long)((unsigned)sg0 % inner) * P.inner_stride;
    };
    float4 xpre[XP];                                   // IO16: the 16 raw bytes (8 bf16) travel in a float4
    auto fetch = [&](long sg0) __attribute__((always_inline)) {
        const size_t xb = (size_t)tile_base(sg0);
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            xpre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pvalid[u]) {
                if (IO16) xpre[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(P.x) + (xb + goff[u]) * 2);
                else xpre[u] = load4_f32_or_bf16(P.x, xb + goff[u], P.io_bf16);
            }
        }
    };
    auto put = [&](char* xs) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            if (IO16 || MODE == MODE_F32) *reinterpret_cast<float4*>(xs + xoff[u]) = xpre[u];
            else *reinterpret_cast<uint2*>(xs + xoff[u]) = make_uint2(pack_bf16x2(xpre[u].x, xpre[u].y), pack_bf16x2(xpre[u].z, xpre[u].w));
        }
    };

    // head h's projection rows: part 0 = q, 1 = k (A operands: rows d), 2 = v (B operand: cols d)
    uint4 wf[NKT][2][3][2];
    {
        const char* wq = reinterpret_cast<const char*>(P.wqkv);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int part = 0; part < 3; ++part)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        wf[kt][ch][part][t] = *reinterpret_cast<const uint4*>(
                            wq + ((size_t)(part * HD + h * D + t * 16 + lp) * P.CPad + kt * KT) * M::ES + ch * 64 + q * 16);
    }
    f32x4 bq[2], bk[2], bv[2];                         // biases in accumulator layout = the accumulators' initial values
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float4 a = *reinterpret_cast<const float4*>(P.bqkv + h * D + t * 16 + 4 * q);
        const float4 b = *reinterpret_cast<const float4*>(P.bqkv + HD + h * D + t * 16 + 4 * q);
        const float c = P.bqkv[2 * HD + h * D + t * 16 + lp];
        bq[t] = f32x4{a.x, a.y, a.z, a.w};
        bk[t] = f32x4{b.x, b.y, b.z, b.w};
        bv[t] = f32x4{c, c, c, c};
    }
    const float escale = P.scale * 1.44269504088896f;  // softmax(scale * s) = exp2((s - max s) * scale * log2 e) / sum
    // out-projection tiles of this wave
    // (TNO 4: a wave = TMO channel tiles x all 4 sequences; TNO 2: 4 channel tiles x 2 sequence pairs; TNO 1 -- C = 32, two channel tiles,
    //  the YAML-literal config_v2_2's level 0 --: 2 channel tiles x 4 sequences)
    const int cot0 = (TNO == 4) ? h * TMO : (TNO == 2) ? (h & 3) : (h & 1);
    const int tn0 = (TNO == 4) ? 0 : (TNO == 2) ? 2 * (h >> 2) : (h >> 1);
    const char* wo = reinterpret_cast<const char*>(P.wo);
    uint4 wof[TMO][NCHO];
#pragma unroll
    for (int tmo = 0; tmo < TMO; ++tmo)
#pragma unroll
        for (int ch = 0; ch < NCHO; ++ch)
        {
            const int orow = min((cot0 + tmo) * 16 + lp, P.C - 1);      // (TNO 1 with C < 32 never occurs: the launcher admits C = 32, 64, 128 only)
            const char* wrow = wo + (size_t)orow * P.HDPad * M::ES + ch * 64;
            if constexpr (OSP) {
                const uint2 lo = *reinterpret_cast<const uint2*>(wrow + q * 8), hi = *reinterpret_cast<const uint2*>(wrow + 32 + q * 8);
                wof[tmo][ch] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            } else wof[tmo][ch] = *reinterpret_cast<const uint4*>(wrow + q * 16);
        }
    f32x4 bo[TMO];
#pragma unroll
    for (int tmo = 0; tmo < TMO; ++tmo) {
        const float4 b = *reinterpret_cast<const float4*>(P.bo + (cot0 + tmo) * 16 + 4 * q);
        bo[tmo] = f32x4{b.x, b.y, b.z, b.w};
    }
    const bool masked = !FULL && P.L < 16;

    const long sg_first = (long)blockIdx.x * nsub * 4;
    // ---- the three phases of a sub-tile ------------------------------------------------------------------------------------------
    // (scalars of the argument block the phase lambdas use: captured as values, so that the block itself is not forced into scratch)
    float* const y_out = P.y;
    const int io16_rt = P.io_bf16, seq_len = P.L;
    // A: per-head attention of the 4 sequences of the x tile `xs` -> os_w[row][h*32 + d]
    auto phase_a = [&](const char* xs, char* os_w) __attribute__((always_inline)) {
#pragma unroll
        for (int s0 = 0; s0 < 4; s0 += SB) {           // SB sequences (16-row tiles) per batch
            f32x4 aq[SB][2], ak[SB][2], av[SB][2];
#pragma unroll
            for (int b = 0; b < SB; ++b)
#pragma unroll
                for (int t = 0; t < 2; ++t) { aq[b][t] = bq[t]; ak[b][t] = bk[t]; av[b][t] = bv[t]; }
#pragma unroll
            for (int b = 0; b < SB; ++b)
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int ch = 0; ch < (TNO == 1 ? 1 : 2); ++ch) {      // (C = 32: the second 32-channel chunk of the padded tile is zeros)
                        const uint4 xf = *reinterpret_cast<const uint4*>(xs + kt * PLANE + ((s0 + b) * 16 + lp) * RS + ch * 64 + q * 16);
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            M::mma(aq[b][t], wf[kt][ch][0][t], xf);
                            M::mma(ak[b][t], wf[kt][ch][1][t], xf);
                            M::mma(av[b][t], xf, wf[kt][ch][2][t]);      // swapped: rows = tokens, cols = d
                        }
                    }
            f32x4 sc[SB];                              // S^T[j, i] (unscaled): lane (i, q) holds keys j = 4q..4q+3
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                sc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
                core_mma16<M, F8>(sc[b], ak[b][0], aq[b][0]);
                core_mma16<M, F8>(sc[b], ak[b][1], aq[b][1]);
            }
            float mx[SB], sum[SB];
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                if (masked) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (4 * q + r >= seq_len) sc[b][r] = -1e30f;
                }
                mx[b] = fmaxf(fmaxf(sc[b][0], sc[b][1]), fmaxf(sc[b][2], sc[b][3]));
            }
#pragma unroll
            for (int b = 0; b < SB; ++b) mx[b] = max_q(mx[b]);
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                const float nmx = -mx[b] * escale;     // exp2((s - max) * k) = exp2(fma(s, k, -max * k)): one FMA per score
                sum[b] = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { sc[b][r] = __builtin_amdgcn_exp2f(fmaf(sc[b][r], escale, nmx)); sum[b] += sc[b][r]; }
            }
#pragma unroll
            for (int b = 0; b < SB; ++b) sum[b] = reduce_q(sum[b]);
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                const float inv = __builtin_amdgcn_rcpf(sum[b]);
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[b][r] *= inv;
            }
#pragma unroll
            for (int b = 0; b < SB; ++b) {             // O^T[d, i] -> os[row i][h*32 + d]
                f32x4 o[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    core_mma16<M, F8>(o[t], av[b][t], sc[b]);
                }
                char* orow = os_w + ((s0 + b) * 16 + lp) * RSO;
                if constexpr (OSP) {
                    *reinterpret_cast<uint4*>(orow + h * D * 2 + q * 16) =
                        make_uint4(pack_bf16x2(o[0][0], o[0][1]), pack_bf16x2(o[0][2], o[0][3]), pack_bf16x2(o[1][0], o[1][1]), pack_bf16x2(o[1][2], o[1][3]));
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t) M::store4(orow, h * D + t * 16 + 4 * q, make_float4(o[t][0], o[t][1], o[t][2], o[t][3]));
                }
            }
        }
    };
    // B: out-projection of this wave's (output-channel tile, sequences) from os_r -> ys_w (fp32)
    auto phase_b = [&](const char* os_r, char* ys_w) __attribute__((always_inline)) {
        f32x4 oacc[TMO][TNO];
#pragma unroll
        for (int i = 0; i < TMO; ++i)
#pragma unroll
            for (int j = 0; j < TNO; ++j) oacc[i][j] = bo[i];
#pragma unroll
        for (int ch = 0; ch < NCHO; ++ch) {
            uint4 bf[TNO];
#pragma unroll
            for (int tn = 0; tn < TNO; ++tn) bf[tn] = *reinterpret_cast<const uint4*>(os_r + ((tn0 + tn) * 16 + lp) * RSO + ch * 64 + q * 16);
#pragma unroll
            for (int tmo = 0; tmo < TMO; ++tmo)
#pragma unroll
                for (int tn = 0; tn < TNO; ++tn) M::mma(oacc[tmo][tn], wof[tmo][ch], bf[tn]);
        }
#pragma unroll
        for (int tmo = 0; tmo < TMO; ++tmo)
#pragma unroll
            for (int tn = 0; tn < TNO; ++tn)
                *reinterpret_cast<f32x4*>(ys_w + ((tn0 + tn) * 16 + lp) * RSY + ((cot0 + tmo) * 16 + 4 * q) * 4) = oacc[tmo][tn];
    };
    // C: every wave stores whole 16-byte pieces of the sub-tile at sg0 in fetch order: coalesced, + residual (the fetched rows)
    auto phase_c = [&](const char* ys_r, long sg0, const float4 (&xres)[XP]) __attribute__((always_inline)) {
        const size_t yb = (size_t)tile_base(sg0);
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            if (!pvalid[u]) continue;
            const float4 o4 = *reinterpret_cast<const float4*>(ys_r + yoff[u]);
            if (IO16) {
                const float4 o5 = *reinterpret_cast<const float4*>(ys_r + yoff[u] + 16);
                const unsigned r0 = __float_as_uint(xres[u].x), r1 = __float_as_uint(xres[u].y), r2 = __float_as_uint(xres[u].z), r3 = __float_as_uint(xres[u].w);
                uint4 w;
                w.x = pack_bf16x2(o4.x + __uint_as_float(r0 << 16), o4.y + __uint_as_float(r0 & 0xFFFF0000u));
                w.y = pack_bf16x2(o4.z + __uint_as_float(r1 << 16), o4.w + __uint_as_float(r1 & 0xFFFF0000u));
                w.z = pack_bf16x2(o5.x + __uint_as_float(r2 << 16), o5.y + __uint_as_float(r2 & 0xFFFF0000u));
                w.w = pack_bf16x2(o5.z + __uint_as_float(r3 << 16), o5.w + __uint_as_float(r3 & 0xFFFF0000u));
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(y_out) + (yb + goff[u]) * 2) = w;
            } else {
                store4_f32_or_bf16(y_out, yb + goff[u], make_float4(o4.x + xres[u].x, o4.y + xres[u].y, o4.z + xres[u].z, o4.w + xres[u].w), io16_rt);
            }
        }
    };

    // number of sub-tiles this workgroup walks (uniform)
    const long left = (P.nseq - sg_first + 3) / 4;
    const int nv = (int)(left < (long)nsub ? (left > 0 ? left : 0) : (long)nsub);
    if (nv <= 0) return;
    fetch(sg_first);
    put(smem);
    __syncthreads();
    if constexpr (PIPE) {
        // ---- software pipeline over sub-tiles, ONE barrier per sub-tile: in iteration `it` the workgroup stores tile it - 2 (C),
        //      runs the attention of tile it (A) and the out-projection of tile it - 1 (B); os and ys are double-buffered, so
        //      the three phases touch different buffers and a wave that is late in one phase does not park the others twice per tile
        //      (two-barrier form: 44 % of wave life parked at barriers / waits at level 0, rocprofv3 PMC) ----
        char* const os2[2] = {os, os + 64 * RSO};
        char* const ys2[2] = {os + 2 * 64 * RSO, os + 2 * 64 * RSO + 64 * RSY};
        float4 xres[2][XP];
        auto iter = [&](int it, auto parc) __attribute__((always_inline)) {
            constexpr int PAR = decltype(parc)::value;  // it & 1
            if (it >= 2 && it - 2 < nv) phase_c(ys2[PAR], sg_first + (long)(it - 2) * 4, xres[PAR]);
            const bool do_a = it < nv, more = it + 1 < nv;
            if (do_a) {
#pragma unroll
                for (int u = 0; u < XP; ++u) xres[PAR][u] = xpre[u];
#if !(VDX_H8_DIAG & 4)
                if (more) fetch(sg_first + (long)(it + 1) * 4);
#endif
#if !(VDX_H8_DIAG & 1)
                phase_a(smem + PAR * BUF, os2[PAR]);
#endif
            }
#if !(VDX_H8_DIAG & 2)
            if (it >= 1 && it - 1 < nv) phase_b(os2[PAR ^ 1], ys2[PAR ^ 1]);
#endif
            if (do_a && more) put(smem + (PAR ^ 1) * BUF);
            __syncthreads();
        };
        for (int it = 0; it < nv + 2; it += 2) {
            iter(it, std::integral_constant<int, 0>{});
            if (it + 1 < nv + 2) iter(it + 1, std::integral_constant<int, 1>{});
        }
    } else {
        for (int sub = 0; sub < nv; ++sub) {
            const long sg0 = sg_first + (long)sub * 4;
            const char* xs = smem + (sub & 1) * BUF;
            const bool more = sub + 1 < nv;
            float4 xcur[XP];                           // this sub-tile's fp32 rows = the residual of its output (no re-read)
#pragma unroll
            for (int u = 0; u < XP; ++u) xcur[u] = xpre[u];
            if (more) fetch(sg0 + 4);
            phase_a(xs, os);
            __syncthreads();
            phase_b(os, ys);
            if (more) put(smem + ((sub + 1) & 1) * BUF);
            __syncthreads();
            phase_c(ys, sg0, xcur);
        }
    }
}

// instrumentation (vdx.h: vdx_set_launch_hook): algorithmic work of an attention block over rows = nseq * L tokens: q|k|v
// projection 2 C 3HD + core 4 L HD (+ out-projection 2 HD C when `with_out`) FLOP per token; x in + result out in their storage type + weights
namespace {
struct AttnWork { double flops, bytes; };
AttnWork attn_work(const AttnArgs& a, int es_w, bool with_out) {
    const double rows = (double)a.nseq * a.L, HD = a.heads * 32.0, eio = a.io_bf16 ? 2.0 : 4.0;
    AttnWork w;
    w.flops = rows * (2.0 * a.C * 3 * HD + 4.0 * a.L * HD + (with_out ? 2.0 * HD * a.C : 0.0));
    w.bytes = rows * a.C * eio + (with_out ? rows * a.C * eio : rows * HD * 2.0) + es_w * (3.0 * HD * a.C + (with_out ? HD * a.C : 0.0));
    return w;
}
}  // namespace

template <int MODE, int NKT, int TMO, int TNO, bool IO16, bool F8 = false, bool FULL = false>
static hipError_t launch_attn_h8_t(const AttnArgs& a, hipStream_t st) {
    using M = Mma<MODE>;
    constexpr int NBUF = (MODE == MODE_BF16 && NKT == 1) ? 2 : 1;     // PIPE (see the kernel): os and ys double-buffered
    const size_t lds = 2 * (size_t)NKT * 64 * ROW_STRIDE + NBUF * ((size_t)64 * (256 * M::ES + 16) + (size_t)64 * (NKT * M::KT * 4 + 16));
    auto kfn = attention_h8_kernel<MODE, NKT, TMO, TNO, IO16, F8, FULL>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const long subtiles = (a.nseq + 3) / 4;
    // sub-tiles per workgroup: the per-head weight fragments (12 KB per wave) are loaded once per workgroup, so fatter workgroups
    // amortise them as long as ~1024 workgroups remain to fill the chip
    const int nsub_cap = 32;
    const int nsub = (int)std::min<long>(nsub_cap, std::max<long>(1, subtiles / 1024));
    const long blocks = (subtiles + nsub - 1) / nsub;
    const AttnWork aw = attn_work(a, M::ES, true);
    LaunchScope ls(st, "attention_h8_kernel", aw.flops, aw.bytes, "<%d, %d, %d, %d, %d, %d, %d> C%d L%d nseq%ld", MODE, NKT, TMO, TNO, (int)IO16, (int)F8, (int)FULL, a.C, a.L, a.nseq);
    hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(512), lds, st, a, nsub);
    return hipGetLastError();
}

// ---- level 0 (C = 64, 16 tokens, 8 heads, bf16 tensors): one WAVE per group of 4 sequences, all heads in the wave ("attention_w") ------
// attention_h8_kernel's knock-out timings add up: 310 us of load -> LDS -> store skeleton + 383 us of per-head attention +
// 111 us of out-projection = the 780 us it takes -- its waves walk the phases of a sub-tile in lockstep (heads meet in LDS twice per
// tile), so memory, VALU and matrix work of a CU never overlap.  Here nothing is shared between waves but the weights: a persistent
// workgroup per CU keeps the q|k|v image (768 x 64 bf16 = 96 KB) and the out-projection image (64 x 256 = 32 KB) in LDS, and each of its
// 8 waves walks its own groups of 4 adjacent sequences with NO workgroup barrier and NO activation byte in LDS:
//   * the x rows of a sequence are fetched straight into MFMA fragments (lane (token, q) = 16 bytes of a 64-byte K chunk; the 4 sequences
//     of a group are 4 adjacent pixels, so a token's 4 x 128 bytes are one contiguous run) one group ahead, and ARE the residual;
//   * per head: q, k, v^T = W_h x^T (+ bias as the accumulators' initial value) with every weight fragment read from LDS once for the 4
//     sequences, scores / softmax / PV in registers as in attention_h8_kernel, stage by stage over the 4 sequences (independent chains);
//   * the PV accumulators of lane (token, q) are d = 4q..4q+3 and 16+4q..16+4q+3 of the head = one B fragment of the out-projection
//     under a permutation of K that the LDS image of Wo carries too: the per-head output never leaves the registers and the 64 output
//     channels accumulate over the heads in 16 accumulators per lane;
//   * Wo's A-tile rows are permuted so that lane (token, q) ends up with 8 consecutive channels per 32-channel half: two 16-byte stores
//     per sequence.
// Waves drift apart freely, so one wave's loads / stores / softmax run under another's MFMAs.
// Measured at level 0 of the N shape (B = 64; attention_h8_kernel on the same boxes: 735-756 us): stages in the compiler's order 654-689 us;
// with every stage's weight fragments read one stage ahead and the 8 PV products issued before their packing (this form) 608-627; +
// a start skew between the two waves of a SIMD and s_setprio over the projection blocks 598-611.  Counters (r03): MFMA busy ~51 %, 45 % of
// wave life issue-stalled, 13 % parked.  Built on the same body and NOT faster: 2 sequences per wave with 12 waves (3 per SIMD, 144
// registers) 637, with 16 waves (128 registers, 68 B scratch) 668; the head loop rotated so that the q / k projections of head h + 1
// are issued in front of the softmax of head h (needs 2 sequences per wave to fit: 210 registers) 723 -- the SIMD's vector issue port
// (an MFMA holds it 8 of its 16 cycles, every VALU op 4, transcendentals 8: ~1800 cycles per head and 4 sequences against 1152 of
// matrix pipe) is what bounds it, not latency: more waves or more overlap inside a wave add issue work (LDS reads per MFMA double).
#ifndef VDX_AW_SKEW
#define VDX_AW_SKEW 40        // s_sleep units (64 cycles) the second wave of a SIMD starts late: the two leave the barrier in the same phase (0: 618-622 us, 16: 625, 40: 598)
#endif
#ifndef VDX_AW_PRIO
#define VDX_AW_PRIO 1          // s_setprio 1 over the projection MFMA blocks (611 vs 618-622 us)
#endif
#ifndef VDX_AW_SB
#define VDX_AW_SB 4           // sequences per wave and group (4: a token's 4 x 128 bytes are one run; 2: half the registers, twice the weight reads per MFMA)
#endif
#ifndef VDX_AW_NW
#define VDX_AW_NW 8           // waves per workgroup (8 = 2 per SIMD at <= 256 registers, 12 = 3 at <= 168)
#endif
// C = 64 or 32 (level 0 of dim-32 networks, the YAML-literal config_v2_2: one K chunk -- the packed rows are padded to 64 channels with
// zeros --, 32 output channels = one pair of tiles); FULL: sequences of exactly 16 tokens, else keys >= L are masked and the rows of
// tokens >= L neither loaded nor stored
template <bool F8, int C = 64, bool FULL = true>
__global__ __launch_bounds__(64 * VDX_AW_NW) void attention_w_kernel(const AttnArgs P, const int groups_per_wave, const long ngroups) {
    using M = Mma<MODE_BF16>;
    constexpr int HD = 256, D = 32, SB = VDX_AW_SB, NT = 64 * VDX_AW_NW;
    constexpr int NCH = C / 32, NTL = C / 16;           // K chunks of the projections, output-channel tiles
    static_assert(C == 64 || C == 32, "C");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Wq = smem;                                   // [768 rows][128 B]: 16-byte chunk c of row r at r * 128 + 16 * (c ^ (r & 7))
    char* Wo = Wq + 3 * HD * 128;                      // [64 rows][512 B]: row R = 16 * tile + i holds channel 32 (tile >> 1) + 8 (i >> 2) + 4 (tile & 1) + (i & 3);
                                                       // piece (head h, q) = d {4q..4q+3, 16+4q..16+4q+3} at 16 * ((4 h + q) ^ (R & 15))
    float* bl = reinterpret_cast<float*>(Wo + C * 512);    // bqkv [768] | bo [C]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lp = lane & 15, q = lane >> 4;

    for (int i = tid; i < 3 * HD * 8; i += NT) {
        const int r = i >> 3, c = i & 7;
        *reinterpret_cast<uint4*>(Wq + r * 128 + 16 * (c ^ (r & 7))) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(P.wqkv) + (size_t)r * 128 + c * 16);
    }
    for (int i = tid; i < C * 32; i += NT) {
        const int R = i >> 5, c = i & 31, hh = c >> 2, qq = c & 3;
        const int tile = R >> 4, ri = R & 15;
        const int co = 32 * (tile >> 1) + 8 * (ri >> 2) + 4 * (tile & 1) + (ri & 3);
        const char* src = reinterpret_cast<const char*>(P.wo) + (size_t)co * 512 + hh * 64 + qq * 8;
        const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 32);
        *reinterpret_cast<uint4*>(Wo + R * 512 + 16 * (c ^ (R & 15))) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    for (int i = tid; i < 3 * HD; i += NT) bl[i] = P.bqkv[i];
    if (tid < C) bl[3 * HD + tid] = P.bo[tid];
    __syncthreads();                                   // the only barrier: weights visible

    const long g_first = ((long)blockIdx.x * VDX_AW_NW + wave_u) * groups_per_wave;
    const long g_end = g_first + groups_per_wave < ngroups ? g_first + groups_per_wave : ngroups;
    if (g_first >= g_end) return;
#if VDX_AW_SKEW
    // the two waves of a SIMD (w, w + 4) leave the barrier in the same phase and would issue their MFMA-heavy and their VALU-heavy stages
    // at the same time: the second one starts half a head iteration late
    if (wave_u >= 4) __builtin_amdgcn_s_sleep(VDX_AW_SKEW);
    if (wave_u >= 8) __builtin_amdgcn_s_sleep(VDX_AW_SKEW);
#endif
    const char* const xg = reinterpret_cast<const char*>(P.x);
    char* const yg = reinterpret_cast<char*>(P.y);
    // per-lane byte offset inside a group: sequence b adds b * inner_stride elements
    const unsigned loff = (unsigned)(lp * P.tok_stride + q * 8) * 2u;
    const bool tok_ok = FULL || lp < P.L;             // this lane's token exists
    const unsigned sstr = (unsigned)P.inner_stride * 2u;
    auto group_base = [&](long g) __attribute__((always_inline)) -> size_t {                       // wave-uniform: scalar unit
        const unsigned inner = (unsigned)P.inner, sg0 = (unsigned)(g * SB);
        return ((size_t)(sg0 / inner) * P.outer_stride + (size_t)(sg0 % inner) * P.inner_stride) * 2;
    };
    uint4 xn[SB][NCH];
    auto fetch = [&](long g) __attribute__((always_inline)) {
        const char* p = xg + group_base(g) + loff;
#pragma unroll
        for (int b = 0; b < SB; ++b)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                xn[b][ch] = make_uint4(0u, 0u, 0u, 0u);
                if (tok_ok) xn[b][ch] = *reinterpret_cast<const uint4*>(p + b * sstr + ch * 64);
            }
    };
    const float escale = P.scale * 1.44269504088896f;
    // LDS read offsets of this lane (head 0): q|k|v rows t * 16 + lp of a part, chunk ch * 4 + q; Wo rows tile * 16 + lp, piece 4 h + q
    int wqo[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) wqo[ch] = lp * 128 + 16 * ((ch * 4 + q) ^ (lp & 7));            // (+ (part * 256 + h * 32 + t * 16) * 128: multiples of 8 rows)
    const int woo = lp * 512;                          // + tile * 8192 + 16 * ((4 h + q) ^ lp)

    uint4 wcur[2][2], wnext[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) wcur[ch][t] = *reinterpret_cast<const uint4*>(Wq + t * 16 * 128 + wqo[ch]);
    fetch(g_first);
    for (long g = g_first; g < g_end; ++g) {
        uint4 xc[SB][NCH];
#pragma unroll
        for (int b = 0; b < SB; ++b)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) xc[b][ch] = xn[b][ch];
        if (g + 1 < g_end) fetch(g + 1);
        f32x4 oacc[SB][NTL];                            // [sequence][tile = 2 wc + tm]
#pragma unroll
        for (int tile = 0; tile < NTL; ++tile) {
            const float4 b4 = *reinterpret_cast<const float4*>(bl + 3 * HD + 32 * (tile >> 1) + 8 * q + 4 * (tile & 1));
#pragma unroll
            for (int b = 0; b < SB; ++b) oacc[b][tile] = f32x4{b4.x, b4.y, b4.z, b4.w};
        }
        // weight fragments one stage ahead: the reads of the next stage's fragments are issued in front of the current stage's MFMAs
        // (every stage otherwise starts with an exposed LDS round trip), the PV products of the 4 sequences before their packing
#pragma unroll 1
        for (int h = 0; h < 8; ++h) {
            const char* wh = Wq + h * (D * 128);
            auto rd = [&](uint4 (&w)[2][2], const char* base) __attribute__((always_inline)) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) w[ch][t] = *reinterpret_cast<const uint4*>(base + t * 16 * 128 + wqo[ch]);
            };
            f32x4 aq[SB][2], ak[SB][2], av[SB][2];
            // q (fragments in wcur)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float4 b4 = *reinterpret_cast<const float4*>(bl + h * D + t * 16 + 4 * q);
#pragma unroll
                for (int b = 0; b < SB; ++b) aq[b][t] = f32x4{b4.x, b4.y, b4.z, b4.w};
            }
            rd(wnext, wh + HD * 128);
            __builtin_amdgcn_sched_barrier(0);
            if (VDX_AW_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int b = 0; b < SB; ++b) M::mma(aq[b][t], wcur[ch][t], xc[b][ch]);
            // k (fragments in wnext)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float4 b4 = *reinterpret_cast<const float4*>(bl + HD + h * D + t * 16 + 4 * q);
#pragma unroll
                for (int b = 0; b < SB; ++b) ak[b][t] = f32x4{b4.x, b4.y, b4.z, b4.w};
            }
            __builtin_amdgcn_sched_barrier(0);
            rd(wcur, wh + 2 * HD * 128);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int b = 0; b < SB; ++b) M::mma(ak[b][t], wnext[ch][t], xc[b][ch]);
            f32x4 sc[SB];
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                sc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
                core_mma16<M, F8>(sc[b], ak[b][0], aq[b][0]);
                core_mma16<M, F8>(sc[b], ak[b][1], aq[b][1]);
            }
            if constexpr (!FULL) {
#pragma unroll
                for (int b = 0; b < SB; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (4 * q + r >= P.L) sc[b][r] = -1e30f;
            }
            // v (fragments in wcur); the out-projection's fragments go to wnext
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float c = bl[2 * HD + h * D + t * 16 + lp];
#pragma unroll
                for (int b = 0; b < SB; ++b) av[b][t] = f32x4{c, c, c, c};
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const char* woh = Wo + woo + 16 * ((4 * h + q) ^ lp);
#pragma unroll
                for (int tile = 0; tile < NTL; ++tile) wnext[tile >> 1][tile & 1] = *reinterpret_cast<const uint4*>(woh + tile * 8192);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int b = 0; b < SB; ++b) M::mma(av[b][t], xc[b][ch], wcur[ch][t]);
            if (VDX_AW_PRIO) __builtin_amdgcn_s_setprio(0);
            float mx[SB], sum[SB];
#pragma unroll
            for (int b = 0; b < SB; ++b) mx[b] = fmaxf(fmaxf(sc[b][0], sc[b][1]), fmaxf(sc[b][2], sc[b][3]));
#pragma unroll
            for (int b = 0; b < SB; ++b) mx[b] = max_q(mx[b]);
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                const float nmx = -mx[b] * escale;
                sum[b] = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { sc[b][r] = __builtin_amdgcn_exp2f(fmaf(sc[b][r], escale, nmx)); sum[b] += sc[b][r]; }
            }
#pragma unroll
            for (int b = 0; b < SB; ++b) sum[b] = reduce_q(sum[b]);
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                const float inv = __builtin_amdgcn_rcpf(sum[b]);
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[b][r] *= inv;
            }
            __builtin_amdgcn_sched_barrier(0);
            rd(wcur, Wq + ((h + 1) & 7) * (D * 128));   // the next head's q fragments (head 0 of the next group after head 7)
            f32x4 o[SB][2];
#pragma unroll
            for (int b = 0; b < SB; ++b)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    o[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    core_mma16<M, F8>(o[b][t], av[b][t], sc[b]);
                }
            __builtin_amdgcn_sched_barrier(0);
            uint4 of[SB];
#pragma unroll
            for (int b = 0; b < SB; ++b)
                of[b] = make_uint4(pack_bf16x2(o[b][0][0], o[b][0][1]), pack_bf16x2(o[b][0][2], o[b][0][3]), pack_bf16x2(o[b][1][0], o[b][1][1]), pack_bf16x2(o[b][1][2], o[b][1][3]));
#pragma unroll
            for (int tile = 0; tile < NTL; ++tile)
#pragma unroll
                for (int b = 0; b < SB; ++b) M::mma(oacc[b][tile], wnext[tile >> 1][tile & 1], of[b]);
        }
        // + residual (the fetched rows), two 16-byte stores per sequence
        char* yp = yg + group_base(g) + loff;
#pragma unroll
        for (int b = 0; b < SB; ++b)
#pragma unroll
            for (int wc = 0; wc < NCH; ++wc) {
                const uint4 r = xc[b][wc];
                const f32x4 a0 = oacc[b][2 * wc], a1 = oacc[b][2 * wc + 1];
                uint4 w;
                w.x = pack_bf16x2(a0[0] + __uint_as_float(r.x << 16), a0[1] + __uint_as_float(r.x & 0xFFFF0000u));
                w.y = pack_bf16x2(a0[2] + __uint_as_float(r.y << 16), a0[3] + __uint_as_float(r.y & 0xFFFF0000u));
                w.z = pack_bf16x2(a1[0] + __uint_as_float(r.z << 16), a1[1] + __uint_as_float(r.z & 0xFFFF0000u));
                w.w = pack_bf16x2(a1[2] + __uint_as_float(r.w << 16), a1[3] + __uint_as_float(r.w & 0xFFFF0000u));
                if (tok_ok) *reinterpret_cast<uint4*>(yp + b * sstr + wc * 64) = w;
            }
    }
}

#ifndef VDX_ATTN_W
#define VDX_ATTN_W 1
#endif
static bool attn_w_eligible(const AttnArgs& a) {
    return VDX_ATTN_W && a.io_bf16 && (a.C == 64 || a.C == 32) && a.CPad == 64 && a.HDPad == 256 && a.heads == 8 && a.L >= 1 && a.L <= 16 && a.inner % 4 == 0 &&
           a.nseq % 4 == 0 && a.nseq >= 256 && a.nseq < (1L << 29) && 3 * a.inner_stride + 15 * a.tok_stride + a.C < (1L << 30);
}
template <bool F8>
static hipError_t launch_attn_w(const AttnArgs& a, hipStream_t st) {
    const size_t lds = 3 * 256 * 128 + (size_t)a.C * 512 + (3 * 256 + 64) * 4;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    const long ngroups = a.nseq / VDX_AW_SB;
    const long waves = std::min<long>((long)cus * VDX_AW_NW, ngroups);
    const int gpw = (int)((ngroups + waves - 1) / waves);
    const long blocks = (ngroups + (long)gpw * VDX_AW_NW - 1) / ((long)gpw * VDX_AW_NW);
    const AttnWork aw = attn_work(a, 2, true);
    auto go = [&](auto kfn) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        LaunchScope ls(st, "attention_w_kernel", aw.flops, aw.bytes, "<fp8 %d> C%d L%d nseq%ld", (int)F8, a.C, a.L, a.nseq);
        hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(64 * VDX_AW_NW), lds, st, a, gpw, ngroups);
        return hipGetLastError();
    };
    if (a.C == 64) return a.L == 16 ? go(attention_w_kernel<F8, 64, true>) : go(attention_w_kernel<F8, 64, false>);
    return a.L == 16 ? go(attention_w_kernel<F8, 32, true>) : go(attention_w_kernel<F8, 32, false>);
}

// ---- wide levels (C >= 256): one workgroup per (head, sequence chunk) ----------------------------------------------------------------
// The per-head q/k/v weights (96 rows x C bf16, up to 99 KB) are loaded into LDS ONCE per workgroup instead of once per 64 rows;
// every wave then walks its own sequences with no workgroup barrier: x fragments (16 tokens x 32 channels) come straight from
// global memory / L2 (each x row is read by the 8 head-workgroups), weight fragments from LDS, and the core runs in registers as in
// attention_reg_kernel.  Output: O[row][head*32 + d] bf16; the out-projection (+bias, +residual) is a plain 1x1 conv_igemm.
template <bool IO16, int TT, bool F8, int LT = 1>
__global__ __launch_bounds__(512) void attention_head_kernel(const AttnArgs P, const int seq_per_block, const int nchunks) {
    // TT 16-token tiles per wave at a time: each weight fragment read from LDS feeds TT MFMAs (one tile per read would make the kernel
    // LDS-bandwidth bound: 6 KB of fragments per 6 MFMAs per wave), and the x fragments run through a 4-deep register ring so that
    // global loads are issued four K steps ahead of their use.  LT = tiles per sequence: 1 = sequences of <= 16 tokens (TT sequences per
    // group), 4 = sequences of <= 64 tokens (the mid block's spatial attention over 8 x 8 pixels; round 3: that launch was the LDS-staged
    // attention_kernel, which re-stages the 786 KB of q|k|v weights for every 64 rows: 367 us at 0.08 of the MFMA peak): the scores of
    // a query tile run over the LT key tiles of its sequence, the softmax over 16 LT keys is in-lane + the two quad swaps.
    using M = Mma<MODE_BF16>;
    constexpr int D = 32, NS = TT / LT;                               // sequences per group
    static_assert(TT % LT == 0, "whole sequences per group");
    extern __shared__ __attribute__((aligned(16))) char smem[];      // W_h [96 rows][C * 2 + 32]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lp = lane & 15, q = lane >> 4;
    // XCD-aware decode (see sla_head_kernel): the 8 heads of one sequence chunk share an XCD and its L2
    const int h = (blockIdx.x >> 3) & 7, HD = P.heads * D;
    const int chunk_id = (blockIdx.x >> 6) * 8 + (blockIdx.x & 7);
    if (chunk_id >= nchunks) return;                                  // (uniform)
    const int RSW = P.C * 2 + 32;                                     // +32: conflict-free ds_read_b128 over 16 rows
    const int cpr = P.C / 8;                                          // 16-byte pieces per weight row
    for (int i = tid; i < 96 * cpr; i += 512) {
        const int row = i / cpr, pc = i - row * cpr;
        const int part = row >> 5, rr = row & 31;
        *reinterpret_cast<uint4*>(smem + row * RSW + pc * 16) = *reinterpret_cast<const uint4*>(
            reinterpret_cast<const char*>(P.wqkv) + ((size_t)(part * HD + h * D + rr) * P.CPad) * 2 + pc * 16);
    }
    f32x4 bq[2], bk[2], bv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float4 a = *reinterpret_cast<const float4*>(P.bqkv + h * D + t * 16 + 4 * q);
        const float4 b = *reinterpret_cast<const float4*>(P.bqkv + HD + h * D + t * 16 + 4 * q);
        const float c = P.bqkv[2 * HD + h * D + t * 16 + lp];
        bq[t] = f32x4{a.x, a.y, a.z, a.w}; bk[t] = f32x4{b.x, b.y, b.z, b.w}; bv[t] = f32x4{c, c, c, c};
    }
    const float escale = P.scale * 1.44269504088896f;
    const bool masked = P.L < 16 * LT;
    __syncthreads();
    const int s0 = chunk_id * seq_per_block;
    const int send = min((int)P.nseq, s0 + seq_per_block);
    const int nkt = P.C / 32;                                         // multiple of 4 (launcher)
    const char* wrow = smem + lp * RSW + q * 16;
    const int inner = (int)P.inner;

    // fetch side: group of NS sequences starting at fs, K step fk; tile tt = token tile tt % LT of sequence fs + tt / LT
    int fs = s0 + w * NS, fk = 0;
    long fro[TT];
    auto set_group = [&]() {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int sq = fs + tt / LT, tok = (tt % LT) * 16 + lp;
            fro[tt] = -1;
            if (tok < P.L && sq < send) fro[tt] = (long)(sq / inner) * P.outer_stride + (long)(sq % inner) * P.inner_stride + (long)tok * P.tok_stride + 8 * q;
        }
    };
    auto fetch = [&](uint4 (&dst)[TT]) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            dst[tt] = make_uint4(0, 0, 0, 0);
            if (fro[tt] >= 0) {
                const size_t e = (size_t)fro[tt] + fk * 32;
                if (IO16) dst[tt] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(P.x) + e * 2);
                else {
                    const float4 a = *reinterpret_cast<const float4*>(P.x + e), b = *reinterpret_cast<const float4*>(P.x + e + 4);
                    dst[tt] = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
                }
            }
        }
        if (++fk == nkt) { fk = 0; fs += 8 * NS; set_group(); }
    };
    uint4 ring[4][TT];
    set_group();
#pragma unroll
    for (int u = 0; u < 4; ++u) fetch(ring[u]);

    for (int cs = s0 + w * NS; cs < send; cs += 8 * NS) {
        long crow[TT];                                                // row index (x element offset / C) of this lane's token of tile tt, or -1
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int sq = cs + tt / LT, tok = (tt % LT) * 16 + lp;
            crow[tt] = -1;
            if (tok < P.L && sq < send) crow[tt] = ((long)(sq / inner) * P.outer_stride + (long)(sq % inner) * P.inner_stride + (long)tok * P.tok_stride) / P.C;
        }
        f32x4 aq[TT][2], ak[TT][2], av[TT][2];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int t = 0; t < 2; ++t) { aq[tt][t] = bq[t]; ak[tt][t] = bk[t]; av[tt][t] = bv[t]; }
        for (int kt = 0; kt < nkt; kt += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const uint4 wq = *reinterpret_cast<const uint4*>(wrow + (0 * 32 + t * 16) * RSW + (kt + u) * 64);
                    const uint4 wk = *reinterpret_cast<const uint4*>(wrow + (1 * 32 + t * 16) * RSW + (kt + u) * 64);
                    const uint4 wv = *reinterpret_cast<const uint4*>(wrow + (2 * 32 + t * 16) * RSW + (kt + u) * 64);
#pragma unroll
                    for (int tt = 0; tt < TT; ++tt) {
                        M::mma(aq[tt][t], wq, ring[u][tt]);
                        M::mma(ak[tt][t], wk, ring[u][tt]);
                        M::mma(av[tt][t], ring[u][tt], wv);           // swapped: rows = tokens, cols = d
                    }
                }
                fetch(ring[u]);
            }
        }
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {                             // query tile tt; its keys: the LT tiles of the same sequence
            const int t0 = (tt / LT) * LT;
            f32x4 sc[LT];                                             // S^T[j, i] (unscaled): lane (i, q) holds keys j = 16 tj + 4q..4q+3
            float mx = -1e30f;
#pragma unroll
            for (int tj = 0; tj < LT; ++tj) {
                sc[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
                core_mma16<M, F8>(sc[tj], ak[t0 + tj][0], aq[tt][0]);
                core_mma16<M, F8>(sc[tj], ak[t0 + tj][1], aq[tt][1]);
                if (masked) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (tj * 16 + 4 * q + r >= P.L) sc[tj][r] = -1e30f;
                }
                mx = fmaxf(mx, fmaxf(fmaxf(sc[tj][0], sc[tj][1]), fmaxf(sc[tj][2], sc[tj][3])));
            }
            mx = max_q(mx);
            float sum = 0.f;
#pragma unroll
            for (int tj = 0; tj < LT; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) { sc[tj][r] = __builtin_amdgcn_exp2f((sc[tj][r] - mx) * escale); sum += sc[tj][r]; }
            const float inv = __builtin_amdgcn_rcpf(reduce_q(sum));
#pragma unroll
            for (int tj = 0; tj < LT; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[tj][r] *= inv;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tj = 0; tj < LT; ++tj) core_mma16<M, F8>(o, av[t0 + tj][t], sc[tj]);
                if (crow[tt] >= 0) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(P.oscratch) + ((size_t)crow[tt] * HD + h * D + t * 16 + 4 * q) * 2) =
                                       make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
            }
        }
    }
}

hipError_t launch_attention_heads(AttnArgs a, hipStream_t st) {
    a.CPad = conv_cin_pad(MODE_BF16, a.C);
    if (a.heads != 8 || a.L > 64 || a.C % 128 || !a.oscratch || a.nseq >= (1L << 30)) return hipErrorInvalidValue;
    const size_t lds = (size_t)96 * (a.C * 2 + 32);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    constexpr int TT = 4;
    const int LT = a.L <= 16 ? 1 : 4, NS = TT / LT;
    // enough workgroups per head to fill the chip twice, at least one group of NS sequences per wave each
    long spb = std::max<long>(8 * NS, (a.nseq * a.heads + 511) / 512);
    spb = (spb + 8 * NS - 1) / (8 * NS) * (8 * NS);
    const long chunks = (a.nseq + spb - 1) / spb;
    auto go = [&](auto kfn) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        const AttnWork aw = attn_work(a, 2, false);
        LaunchScope ls(st, "attention_head_kernel", aw.flops, aw.bytes, "<io16 %d, %d, fp8 %d, lt %d> C%d L%d nseq%ld", a.io_bf16, TT, a.fp8_core, LT, a.C, a.L, a.nseq);
        hipLaunchKernelGGL(kfn, dim3((unsigned)((chunks + 7) / 8 * 64)), dim3(512), lds, st, a, (int)spb, (int)chunks);
        return hipGetLastError();
    };
    if (LT == 4) {                                     // (more than 16 tokens: the fp8 flag is ignored, as in the LDS-staged kernel)
        return a.io_bf16 ? go(attention_head_kernel<true, TT, false, 4>) : go(attention_head_kernel<false, TT, false, 4>);
    }
    if (a.fp8_core) return a.io_bf16 ? go(attention_head_kernel<true, TT, true>) : go(attention_head_kernel<false, TT, true>);
    return a.io_bf16 ? go(attention_head_kernel<true, TT, false>) : go(attention_head_kernel<false, TT, false>);
}

template <int MODE, int TMA>
static hipError_t launch_attn_reg_t(const AttnArgs& a, hipStream_t st) {
    const size_t lds = 512 + (size_t)(64 + 96) * ROW_STRIDE + (size_t)TMA * 16 * (32 * Mma<MODE>::ES + 16);
    const long blocks = (a.nseq + 3) / 4;
    const AttnWork aw = attn_work(a, Mma<MODE>::ES, true);
    LaunchScope ls(st, "attention_reg_kernel", aw.flops, aw.bytes, "<%d, %d, fp8 %d> C%d L%d nseq%ld io16 %d", MODE, TMA, a.fp8_core, a.C, a.L, a.nseq, a.io_bf16);
    if constexpr (MODE == MODE_BF16) {
        if (a.fp8_core) { hipLaunchKernelGGL((attention_reg_kernel<MODE, TMA, true>), dim3((unsigned)blocks), dim3(256), lds, st, a); return hipGetLastError(); }
    }
    hipLaunchKernelGGL((attention_reg_kernel<MODE, TMA, false>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_attn_reg(const AttnArgs& a, hipStream_t st) {
    if (a.C <= 64) return launch_attn_reg_t<MODE, 4>(a, st);
    if (a.C <= 128) return launch_attn_reg_t<MODE, 8>(a, st);
    if (a.C <= 256) return launch_attn_reg_t<MODE, 16>(a, st);
    if (a.C <= 512) return launch_attn_reg_t<MODE, 32>(a, st);
    if (a.C <= 1024) return launch_attn_reg_t<MODE, 64>(a, st);
    return hipErrorInvalidValue;
}

template <int MODE, int LP, int TMO>
static hipError_t launch_attn_t(const AttnArgs& a, hipStream_t st) {
    constexpr int KC = Mma<MODE>::KC;
    constexpr int NSEQ = 64 / LP;
    constexpr int NCHL = (LP + KC - 1) / KC;
    constexpr int RSV = NCHL * 64 + 16;
    const size_t lds = 512 + (size_t)(64 * 4 + 96) * ROW_STRIDE + (size_t)(NSEQ * 32 + 64) * RSV;
    auto kfn = attention_kernel<MODE, LP, TMO>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const long blocks = (a.nseq + NSEQ - 1) / NSEQ;
    const AttnWork aw = attn_work(a, Mma<MODE>::ES, true);
    LaunchScope ls(st, "attention_kernel", aw.flops, aw.bytes, "<%d, %d, %d> C%d L%d nseq%ld io16 %d", MODE, LP, TMO, a.C, a.L, a.nseq, a.io_bf16);
    hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

template <int MODE, int LP>
static hipError_t launch_attn_l(const AttnArgs& a, hipStream_t st) {
    if (a.C <= 64) return launch_attn_t<MODE, LP, 1>(a, st);
    if (a.C <= 128) return launch_attn_t<MODE, LP, 2>(a, st);
    if (a.C <= 256) return launch_attn_t<MODE, LP, 4>(a, st);
    if (a.C <= 512) return launch_attn_t<MODE, LP, 8>(a, st);
    if (a.C <= 1024) return launch_attn_t<MODE, LP, 16>(a, st);
    return hipErrorInvalidValue;
}

template <int MODE>
static hipError_t launch_attn_m(const AttnArgs& a, hipStream_t st) {
    const bool use_reg = true;        // debugging switch: force the LDS-staged kernel
    const bool use_h8 = true;        // debugging switch: skip the one-wave-per-head kernel
    const bool h8_ok = a.L <= 16 && a.heads == 8 && a.inner % 4 == 0 && a.nseq % 4 == 0 && a.nseq < (1L << 31) &&
                       3 * a.inner_stride + 15 * a.tok_stride + a.C < (1L << 31);
    if constexpr (MODE != MODE_F16) {                  // (the one-wave-per-head kernels hard-code the bf16 / f32 register formats)
    if (h8_ok && use_reg && use_h8) {
        const int nkt = a.CPad / Mma<MODE>::KT;
        if constexpr (MODE == MODE_BF16) {
            if (a.fp8_core) {                                 // fp8 QK^T / PV (vdx_set_attention_fp8)
                if (attn_w_eligible(a)) return launch_attn_w<true>(a, st);
                if (a.io_bf16 && a.C == 64) return launch_attn_h8_t<MODE, 1, 1, 2, true, true>(a, st);
                if (a.io_bf16 && a.C == 128) return launch_attn_h8_t<MODE, 2, 1, 4, true, true>(a, st);
                if (a.C == 64 && nkt == 1) return launch_attn_h8_t<MODE, 1, 1, 2, false, true>(a, st);
                if (a.C == 128 && nkt == 2) return launch_attn_h8_t<MODE, 2, 1, 4, false, true>(a, st);
            }
            if (attn_w_eligible(a)) return launch_attn_w<false>(a, st);                       // level 0: one wave per group of 4 sequences
            if (a.io_bf16 && a.C == 64 && a.L == 16) return launch_attn_h8_t<MODE, 1, 1, 2, true, false, true>(a, st);
            if (a.io_bf16 && a.C == 128 && a.L == 16) return launch_attn_h8_t<MODE, 2, 1, 4, true, false, true>(a, st);
            if (a.io_bf16 && a.C == 64) return launch_attn_h8_t<MODE, 1, 1, 2, true>(a, st);
            if (a.io_bf16 && a.C == 128) return launch_attn_h8_t<MODE, 2, 1, 4, true>(a, st);
            if (a.io_bf16 && a.C == 32) return launch_attn_h8_t<MODE, 1, 1, 1, true>(a, st);      // dim 32 (configs/config_v2_2.yaml as written): level 0
        }
        if (a.C == 64 && nkt == 1) return launch_attn_h8_t<MODE, 1, 1, 2, false>(a, st);
        if (a.C == 64 && nkt == 2) return launch_attn_h8_t<MODE, 2, 1, 2, false>(a, st);
        if (a.C == 128 && nkt == 2) return launch_attn_h8_t<MODE, 2, 1, 4, false>(a, st);
    }
    }
    if (a.L <= 16 && use_reg && !(MODE == MODE_F32 && a.C > 512)) return launch_attn_reg<MODE>(a, st);    // (f32 weight tiles of C = 1024 exceed the LDS: staged form)
    if (a.L <= 16) return launch_attn_l<MODE, 16>(a, st);
    if (a.L <= 32) return launch_attn_l<MODE, 32>(a, st);
    if (a.L <= 64) return launch_attn_l<MODE, 64>(a, st);
    return hipErrorInvalidValue;
}

// ---- long sequences (L > 64: the bottleneck spatial attention of frames larger than 64 x 64, e.g. 16 x 16 = 256 tokens at 128 x 128) ----
// The projections run as 1x1 convs (model.hip); this is the core: one workgroup per (sequence, head), one query row per thread, K and V
// of the head in LDS (every lane reads the same key: broadcast), online softmax in fp32 VALU -- exact in both arithmetic modes.  The
// block is ONE launch per forward with ~3 GFLOP at B = 1, so it is written for correctness, not for the matrix cores.
// qkv [rows][3 * heads * 32] fp32 (biased, q unscaled; modules.py:261-271,294), o [rows][heads * 32] fp32; rows = nseq * L, token-major.
__global__ __launch_bounds__(256) void attention_long_core_kernel(const float* __restrict__ qkv, float* __restrict__ o, int L, int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) float kv[];          // K [L][32] | V [L][32]
    const int seq = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    const int HD = heads * 32;
    const float* base = qkv + (size_t)seq * L * 3 * HD;
    float* Ks = kv; float* Vs = kv + (size_t)L * 32;
    for (int i = tid; i < L * 8; i += 256) {
        const int j = i >> 3, c = (i & 7) * 4;
        *reinterpret_cast<float4*>(Ks + j * 32 + c) = *reinterpret_cast<const float4*>(base + (size_t)j * 3 * HD + HD + h * 32 + c);
        *reinterpret_cast<float4*>(Vs + j * 32 + c) = *reinterpret_cast<const float4*>(base + (size_t)j * 3 * HD + 2 * HD + h * 32 + c);
    }
    __syncthreads();
    for (int row = tid; row < L; row += 256) {
        float q[32], acc[32];
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)row * 3 * HD + h * 32 + c);
            q[c] = v.x * scale; q[c + 1] = v.y * scale; q[c + 2] = v.z * scale; q[c + 3] = v.w * scale;      // q /= sqrt(d) (modules.py:294)
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) acc[c] = 0.f;
        float m = -3.0e38f, l = 0.f;
        for (int j = 0; j < L; ++j) {
            float sdot = 0.f;
#pragma unroll
            for (int c = 0; c < 32; ++c) sdot = fmaf(q[c], Ks[j * 32 + c], sdot);
            const float mn = fmaxf(m, sdot), corr = __expf(m - mn), pj = __expf(sdot - mn);
            l = l * corr + pj;
#pragma unroll
            for (int c = 0; c < 32; ++c) acc[c] = fmaf(pj, Vs[j * 32 + c], acc[c] * corr);
            m = mn;
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int c = 0; c < 32; c += 4)
            *reinterpret_cast<float4*>(o + ((size_t)seq * L + row) * HD + h * 32 + c) = make_float4(acc[c] * inv, acc[c + 1] * inv, acc[c + 2] * inv, acc[c + 3] * inv);
    }
}

hipError_t launch_attention_long_core(const float* qkv, float* o, long nseq, int L, int heads, float scale, hipStream_t st) {
    const size_t lds = (size_t)L * 64 * 4;
    if (lds > 160 * 1024 || nseq <= 0 || nseq >= (1L << 31)) return hipErrorInvalidValue;            // L <= 640 tokens
    auto kfn = attention_long_core_kernel;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    LaunchScope ls(st, "attention_long_core_kernel", 4.0 * nseq * L * L * heads * 32, 4.0 * nseq * L * heads * 32 * 4, "L%d nseq%ld heads%d", L, nseq, heads);
    hipLaunchKernelGGL(kfn, dim3((unsigned)nseq, heads), dim3(256), lds, st, qkv, o, L, heads, scale);
    return hipGetLastError();
}

hipError_t launch_attention(int mode, AttnArgs a, hipStream_t st) {
    a.CPad = conv_cin_pad(mode, a.C);
    a.HDPad = conv_cin_pad(mode, a.heads * 32);
    return mode == MODE_F32 ? launch_attn_m<MODE_F32>(a, st) : mode == MODE_F16 ? launch_attn_m<MODE_F16>(a, st) : launch_attn_m<MODE_BF16>(a, st);
}

}  // namespace vdx
