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
#include "vdx_internal.h"
#include <stdlib.h>

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
            if (ro >= 0 && c < P.C) v = *reinterpret_cast<const float4*>(P.x + ro + c);
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
            const float4 xr = *reinterpret_cast<const float4*>(P.x + ro + co);
            float4 v;
            v.x = oacc[tmo][tn][0] + bo.x + xr.x; v.y = oacc[tmo][tn][1] + bo.y + xr.y;
            v.z = oacc[tmo][tn][2] + bo.z + xr.z; v.w = oacc[tmo][tn][3] + bo.w + xr.w;
            *reinterpret_cast<float4*>(P.y + ro + co) = v;
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
template <int MODE, int TMA>            // TMA = C / 16 output-channel tiles (all owned by every wave, for its 16 rows)
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
            if (ro >= 0 && c < P.C) v = *reinterpret_cast<const float4*>(P.x + ro + c);
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
                if (nh < P.heads && !(P.dbg & 2)) wfetch(nh, nk);
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
        M::mma16(s, ak[0], aq[0]);
        M::mma16(s, ak[1], aq[1]);
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
            M::mma16(o, av[t], s);
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
        if (!(P.dbg & 4)) xr = *reinterpret_cast<const float4*>(P.x + ro + co);
        float4 v;
        v.x = oacc[tm][0] + bo.x + xr.x; v.y = oacc[tm][1] + bo.y + xr.y;
        v.z = oacc[tm][2] + bo.z + xr.z; v.w = oacc[tm][3] + bo.w + xr.w;
        if (!(P.dbg & 8)) *reinterpret_cast<float4*>(P.y + ro + co) = v;
    }
}

template <int MODE, int TMA>
static hipError_t launch_attn_reg_t(const AttnArgs& a, hipStream_t st) {
    const size_t lds = 512 + (size_t)(64 + 96) * ROW_STRIDE + (size_t)TMA * 16 * (32 * Mma<MODE>::ES + 16);
    const long blocks = (a.nseq + 3) / 4;
    hipLaunchKernelGGL((attention_reg_kernel<MODE, TMA>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_attn_reg(const AttnArgs& a, hipStream_t st) {
    if (a.C <= 64) return launch_attn_reg_t<MODE, 4>(a, st);
    if (a.C <= 128) return launch_attn_reg_t<MODE, 8>(a, st);
    if (a.C <= 256) return launch_attn_reg_t<MODE, 16>(a, st);
    if (a.C <= 512) return launch_attn_reg_t<MODE, 32>(a, st);
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
    hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

template <int MODE, int LP>
static hipError_t launch_attn_l(const AttnArgs& a, hipStream_t st) {
    if (a.C <= 64) return launch_attn_t<MODE, LP, 1>(a, st);
    if (a.C <= 128) return launch_attn_t<MODE, LP, 2>(a, st);
    if (a.C <= 256) return launch_attn_t<MODE, LP, 4>(a, st);
    if (a.C <= 512) return launch_attn_t<MODE, LP, 8>(a, st);
    return hipErrorInvalidValue;
}

template <int MODE>
static hipError_t launch_attn_m(const AttnArgs& a, hipStream_t st) {
    static const bool use_reg = getenv("VDX_ATTN_LDS") == nullptr;        // debugging switch: force the LDS-staged kernel
    if (a.L <= 16 && use_reg) return launch_attn_reg<MODE>(a, st);
    if (a.L <= 16) return launch_attn_l<MODE, 16>(a, st);
    if (a.L <= 32) return launch_attn_l<MODE, 32>(a, st);
    if (a.L <= 64) return launch_attn_l<MODE, 64>(a, st);
    return hipErrorInvalidValue;
}

hipError_t launch_attention(int mode, AttnArgs a, hipStream_t st) {
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("VDX_ATTN_DBG"); dbg = e ? atoi(e) : 0; } a.dbg = dbg; }
    a.CPad = conv_cin_pad(mode, a.C);
    a.HDPad = conv_cin_pad(mode, a.heads * 32);
    return mode == MODE_F32 ? launch_attn_m<MODE_F32>(a, st) : launch_attn_m<MODE_BF16>(a, st);
}

}  // namespace vdx
