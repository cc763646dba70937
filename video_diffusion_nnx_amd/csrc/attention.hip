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
    uint4 wpre[3];
    auto wfetch = [&](int h, int kt) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i = tid + 256 * j;
            const int row = i >> 3, pc = i & 7;
            const int grow = (row >> 5) * HD + h * D + (row & 31);
            wpre[j] = *reinterpret_cast<const uint4*>(wq + ((size_t)grow * P.CPad + (size_t)kt * KT) * M::ES + pc * 16);
        }
    };
    auto wput = [&]() {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i = tid + 256 * j;
            *reinterpret_cast<uint4*>(ws + (i >> 3) * RS + (i & 7) * 16) = wpre[j];
        }
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
    if (a.L <= 16) return launch_attn_l<MODE, 16>(a, st);
    if (a.L <= 32) return launch_attn_l<MODE, 32>(a, st);
    if (a.L <= 64) return launch_attn_l<MODE, 64>(a, st);
    return hipErrorInvalidValue;
}

hipError_t launch_attention(int mode, AttnArgs a, hipStream_t st) {
    a.CPad = conv_cin_pad(mode, a.C);
    a.HDPad = conv_cin_pad(mode, a.heads * 32);
    return mode == MODE_F32 ? launch_attn_m<MODE_F32>(a, st) : launch_attn_m<MODE_BF16>(a, st);
}

}  // namespace vdx
