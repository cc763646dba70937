// SpatialLinearAttention (+ Residual, no-op PreNorm) on MFMA (gfx950).
//
// Reference: /root/reference/modules.py:64-129 wrapped by Residual(PreNorm(...)) (unet3d.py:170-178).
// Per frame n and head h (N = H*W pixels, D = 32):
//   q = softmax_D(x Wq)   (over the 32 channels of the head; the `* scale` is dead code)
//   k = softmax_N(x Wk)   (over the N pixels)
//   ctx[d,e] = sum_n k[d,n] v[e,n] ;  out[e,n] = sum_d ctx[d,e] q[d,n] ;  y = to_out(out) + x
// Three launches, none of which writes q/k/v to HBM:
//   sla_ctx_kernel     (frame, pixel-chunk, head): K/V projection of 64-pixel sub-tiles, ONLINE softmax over
//                      pixels (running max / sum / rescaled 32x32 context in the accumulators) -> partial
//   sla_combine_kernel (frame, head): merges the chunk partials, normalises, emits ctx^T in the MMA type
//   sla_out_kernel     (frame, 64 pixels): Q projection of all heads, softmax over D with wavefront
//                      shuffles, out = ctx^T q, to_out GEMM accumulated in registers, + residual
#include "vdx_common.h"
#include "vdx_internal.h"

namespace vdx {

constexpr int SLA_PART = 32 * 32 + 64;     // floats per partial: ctx[32][32] | m[32] | s[32]

template <int MODE>
__global__ __launch_bounds__(256) void sla_ctx_kernel(const SlaArgs P) {
    using M = Mma<MODE>;
    constexpr int KT = M::KT, KC = M::KC, RS = ROW_STRIDE;
    constexpr int APIECES = KT / 4;
    constexpr int NCHN = 64 / KC;                      // chunks covering 64 pixels
    constexpr int RSE = NCHN * 64 + 16;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* m_run = reinterpret_cast<float*>(smem);     // [32]
    float* s_run = m_run + 32;                         // [32]
    float* alpha = s_run + 32;                         // [32]
    float* pm = alpha + 32;                            // [2][32] partial max / sum per row-wave
    char* xs = smem + 1024;
    char* ws = xs + 64 * RS;
    char* eT = ws + 64 * RS;                           // [32][RSE]
    char* vT = eT + 32 * RSE;                          // [32][RSE]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lp = lane & 15, q = lane >> 4;
    const int wc = w & 1, wr = w >> 1;
    const int h = blockIdx.y;
    const int n = blockIdx.x / P.nchunk, chunk = blockIdx.x % P.nchunk;
    const int R = P.nsub * 64;
    const float* xf = P.x + (size_t)n * P.N * P.C;
    const int nkt = P.CPad / KT;
    const char* wk = reinterpret_cast<const char*>(P.wk);
    const char* wv = reinterpret_cast<const char*>(P.wv);

    if (tid < 32) { m_run[tid] = -1e30f; s_run[tid] = 0.f; }
    f32x4 cacc = f32x4{0.f, 0.f, 0.f, 0.f};            // ctx tile (dt = w>>1 rows d, et = w&1 cols e)
    const int dt = w >> 1, et = w & 1;

    // weight tile (k | v rows of this head): resident across sub-tiles when C fits one K tile
    const bool w_resident = (nkt == 1);
    auto stage_w = [&](int kt) {
        for (int i = tid; i < 64 * 8; i += 256) {
            const int row = i >> 3, pc = i & 7;
            const char* src = (row < 32 ? wk : wv) + ((size_t)(h * 32 + (row & 31)) * P.CPad + (size_t)kt * KT) * M::ES + pc * 16;
            *reinterpret_cast<uint4*>(ws + row * RS + pc * 16) = *reinterpret_cast<const uint4*>(src);
        }
    };
    // x sub-tile prefetched through registers one sub-tile ahead (resident-weight case: one K tile per sub-tile)
    constexpr int XP = 64 * APIECES / 256;                  // float4 pieces per thread
    float4 xpre[XP];
    auto xfetch = [&](int r0, int kt) {
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int i = tid + 256 * u;
            const int row = i / APIECES, pc = i % APIECES;
            const int c = kt * KT + pc * 4;
            xpre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + row < P.N && c < P.C) xpre[u] = *reinterpret_cast<const float4*>(xf + (size_t)(r0 + row) * P.C + c);
        }
    };
    auto xput = [&]() {
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int i = tid + 256 * u;
            M::store4(xs + (i / APIECES) * RS, (i % APIECES) * 4, xpre[u]);
        }
    };
    if (w_resident) stage_w(0);
    xfetch(chunk * R, 0);

    for (int sub = 0; sub < P.nsub; ++sub) {
        const int r0 = chunk * R + sub * 64;
        if (r0 >= P.N) break;                          // uniform across the workgroup
        f32x4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int kt = 0; kt < nkt; ++kt) {
            __syncthreads();
            xput();
            if (!w_resident) stage_w(kt);
            __syncthreads();
            {   // next x tile: next K tile of this sub-tile, or the first of the next sub-tile
                int nk = kt + 1, nr0 = r0;
                if (nk == nkt) { nk = 0; nr0 = r0 + 64; }
                if (nr0 < P.N && nr0 < (chunk + 1) * R) xfetch(nr0, nk);
            }
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                uint4 bf[2], af[2];
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) bf[tn] = *reinterpret_cast<const uint4*>(xs + ((wr * 2 + tn) * 16 + lp) * RS + ch * 64 + q * 16);
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) af[tm] = *reinterpret_cast<const uint4*>(ws + ((wc * 2 + tm) * 16 + lp) * RS + ch * 64 + q * 16);
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) M::mma(acc[tm][tn], af[tm], bf[tn]);
            }
        }
        bool rv[2];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) rv[tn] = (r0 + (wr * 2 + tn) * 16 + lp) < P.N;
        // ---- online softmax statistics of k over the pixels of this sub-tile ----
        if (wc == 0) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float mx = fmaxf(rv[0] ? acc[tm][0][r] : -1e30f, rv[1] ? acc[tm][1][r] : -1e30f);
                    mx = max16(mx);
                    if (lp == 0) pm[wr * 32 + tm * 16 + 4 * q + r] = mx;
                }
        }
        __syncthreads();
        if (tid < 32) {
            const float mc = fmaxf(pm[tid], pm[32 + tid]);
            const float mo = m_run[tid];
            const float mn = fmaxf(mo, mc);
            alpha[tid] = __expf(mo - mn);
            m_run[tid] = mn;
        }
        __syncthreads();
        if (wc == 0) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = tm * 16 + 4 * q + r;
                    const float mn = m_run[d];
                    float ssum = 0.f;
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        const float e = rv[tn] ? __expf(acc[tm][tn][r] - mn) : 0.f;
                        M::store1(eT + d * RSE, (wr * 2 + tn) * 16 + lp, e);
                        ssum += e;
                    }
                    ssum = reduce16(ssum);
                    if (lp == 0) pm[wr * 32 + d] = ssum;
                }
        } else {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = tm * 16 + 4 * q + r;
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        M::store1(vT + e * RSE, (wr * 2 + tn) * 16 + lp, rv[tn] ? acc[tm][tn][r] : 0.f);
                }
        }
        __syncthreads();
        if (tid < 32) s_run[tid] = s_run[tid] * alpha[tid] + pm[tid] + pm[32 + tid];
        // ---- ctx = ctx * alpha[d] + e^T v over the 64 pixels ----
#pragma unroll
        for (int r = 0; r < 4; ++r) cacc[r] *= alpha[dt * 16 + 4 * q + r];
#pragma unroll
        for (int ch = 0; ch < NCHN; ++ch) {
            const uint4 a = *reinterpret_cast<const uint4*>(eT + (dt * 16 + lp) * RSE + ch * 64 + q * 16);
            const uint4 bv = *reinterpret_cast<const uint4*>(vT + (et * 16 + lp) * RSE + ch * 64 + q * 16);
            M::mma(cacc, a, bv);
        }
    }
    __syncthreads();
    float* part = P.part + ((size_t)(n * P.nchunk + chunk) * P.heads + h) * SLA_PART;
#pragma unroll
    for (int r = 0; r < 4; ++r) part[(dt * 16 + 4 * q + r) * 32 + et * 16 + lp] = cacc[r];
    if (tid < 32) { part[1024 + tid] = m_run[tid]; part[1056 + tid] = s_run[tid]; }
}

// merges chunk partials -> ctxT[n][h][e][d] (MMA element type), normalised by the softmax denominator
template <int MODE>
__global__ __launch_bounds__(256) void sla_combine_kernel(const float* __restrict__ part, void* __restrict__ ctxT, int nchunk, int heads) {
    using M = Mma<MODE>;
    const int nh = blockIdx.x;                                 // n * heads + h
    const int n = nh / heads, h = nh % heads;
    __shared__ float Mx[32], Sx[32];
    const int tid = threadIdx.x;
    if (tid < 32) {
        float mx = -1e30f;
        for (int c = 0; c < nchunk; ++c) mx = fmaxf(mx, part[((size_t)(n * nchunk + c) * heads + h) * SLA_PART + 1024 + tid]);
        float s = 0.f;
        for (int c = 0; c < nchunk; ++c) {
            const float* p = part + ((size_t)(n * nchunk + c) * heads + h) * SLA_PART;
            s += p[1056 + tid] * __expf(p[1024 + tid] - mx);
        }
        Mx[tid] = mx; Sx[tid] = s;
    }
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) {
        const int d = i >> 5, e = i & 31;
        float v = 0.f;
        for (int c = 0; c < nchunk; ++c) {
            const float* p = part + ((size_t)(n * nchunk + c) * heads + h) * SLA_PART;
            v += p[i] * __expf(p[1024 + d] - Mx[d]);
        }
        v /= Sx[d];
        char* row = reinterpret_cast<char*>(ctxT) + ((size_t)nh * 32 + e) * 32 * M::ES;
        M::store1(row, d, v);
    }
}

template <int MODE, int TMO>
__global__ __launch_bounds__(256) void sla_out_kernel(const SlaArgs P) {
    using M = Mma<MODE>;
    constexpr int KT = M::KT, KC = M::KC, RS = ROW_STRIDE;
    constexpr int APIECES = KT / 4;
    constexpr int NCHD = 32 / KC;
    constexpr int RSQ = (MODE == MODE_F32) ? 160 : 80;         // per-wave q buffer row stride ([64 rows][32 d])
    constexpr int HD = 256;
    constexpr int RSO = HD * M::ES + 16;                       // os row stride ([64 rows][256])
    constexpr int NCHO = HD / KC;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // phase A: xs [64][RS] | ws [256][RS]      phase B (aliases A): qb [4 waves][64][RSQ] | os [64][RSO]
    char* xs = smem;
    char* ws = xs + 64 * RS;
    char* qb = smem;
    char* os = qb + 4 * 64 * RSQ;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lp = lane & 15, q = lane >> 4;
    const int tiles = (P.N + 63) / 64;
    const int n = blockIdx.x / tiles, r0 = (blockIdx.x % tiles) * 64;
    const float* xf = P.x + (size_t)n * P.N * P.C;
    float* yf = P.y + (size_t)n * P.N * P.C;
    const int nkt = P.CPad / KT;
    const char* wq = reinterpret_cast<const char*>(P.wq);
    const char* wo = reinterpret_cast<const char*>(P.wo);

    // ---------------- GEMM1: q[256, 64] ; wave w owns heads 2w, 2w+1 ----------------
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        for (int i = tid; i < 64 * APIECES; i += 256) {
            const int row = i / APIECES, pc = i % APIECES;
            const int c = kt * KT + pc * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + row < P.N && c < P.C) v = *reinterpret_cast<const float4*>(xf + (size_t)(r0 + row) * P.C + c);
            M::store4(xs + row * RS, pc * 4, v);
        }
        for (int i = tid; i < 256 * 8; i += 256) {
            const int row = i >> 3, pc = i & 7;
            *reinterpret_cast<uint4*>(ws + row * RS + pc * 16) =
                *reinterpret_cast<const uint4*>(wq + ((size_t)row * P.CPad + (size_t)kt * KT) * M::ES + pc * 16);
        }
        __syncthreads();
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            uint4 bf[4], af[4];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) bf[tn] = *reinterpret_cast<const uint4*>(xs + (tn * 16 + lp) * RS + ch * 64 + q * 16);
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) af[tm] = *reinterpret_cast<const uint4*>(ws + ((w * 4 + tm) * 16 + lp) * RS + ch * 64 + q * 16);
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) M::mma(acc[tm][tn], af[tm], bf[tn]);
        }
    }
    __syncthreads();                                            // phase A buffers are dead from here
    // ---------------- softmax over D, out_h = ctx^T q, per head of this wave ----------------
    char* myq = qb + w * 64 * RSQ;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int h = 2 * w + hh;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            float mx = -1e30f;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[2 * hh + t2][tn][r]);
            mx = max_q(mx);
            float e[2][4], sum = 0.f;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 4; ++r) { e[t2][r] = __expf(acc[2 * hh + t2][tn][r] - mx); sum += e[t2][r]; }
            sum = reduce_q(sum);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
                M::store4(myq + (tn * 16 + lp) * RSQ, t2 * 16 + 4 * q, make_float4(e[t2][0] * inv, e[t2][1] * inv, e[t2][2] * inv, e[t2][3] * inv));
        }
        __syncthreads();
        const char* ct = reinterpret_cast<const char*>(P.ctxT) + (size_t)(n * P.heads + h) * 32 * 32 * M::ES;
#pragma unroll
        for (int et = 0; et < 2; ++et) {
            f32x4 o[4];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) o[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ch = 0; ch < NCHD; ++ch) {
                const uint4 a = *reinterpret_cast<const uint4*>(ct + (size_t)(et * 16 + lp) * 32 * M::ES + ch * 64 + q * 16);
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    const uint4 bq = *reinterpret_cast<const uint4*>(myq + (tn * 16 + lp) * RSQ + ch * 64 + q * 16);
                    M::mma(o[tn], a, bq);
                }
            }
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
                M::store4(os + (tn * 16 + lp) * RSO, h * 32 + et * 16 + 4 * q, make_float4(o[tn][0], o[tn][1], o[tn][2], o[tn][3]));
        }
        __syncthreads();
    }
    // ---------------- to_out: y[C, 64] = Wout[C, 256] . os^T  (+ residual) ----------------
    f32x4 oacc[TMO][4];
#pragma unroll
    for (int i = 0; i < TMO; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < NCHO; ++ch) {
        uint4 bf[4];
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) bf[tn] = *reinterpret_cast<const uint4*>(os + (tn * 16 + lp) * RSO + ch * 64 + q * 16);
#pragma unroll
        for (int tmo = 0; tmo < TMO; ++tmo) {
            const int co = (w * TMO + tmo) * 16 + lp;
            uint4 a = make_uint4(0, 0, 0, 0);
            if (co < P.C) a = *reinterpret_cast<const uint4*>(wo + (size_t)co * HD * M::ES + ch * 64 + q * 16);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) M::mma(oacc[tmo][tn], a, bf[tn]);
        }
    }
#pragma unroll
    for (int tmo = 0; tmo < TMO; ++tmo) {
        const int co = (w * TMO + tmo) * 16 + 4 * q;
        if (co >= P.C) continue;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const int row = r0 + tn * 16 + lp;
            if (row >= P.N) continue;
            const float4 xr = *reinterpret_cast<const float4*>(xf + (size_t)row * P.C + co);
            float4 v;
            v.x = oacc[tmo][tn][0] + xr.x; v.y = oacc[tmo][tn][1] + xr.y;
            v.z = oacc[tmo][tn][2] + xr.z; v.w = oacc[tmo][tn][3] + xr.w;
            *reinterpret_cast<float4*>(yf + (size_t)row * P.C + co) = v;
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------

void sla_plan(int N, int& nsub, int& nchunk) {
    const int tiles = (N + 63) / 64;
    nsub = std::min(tiles, 8);
    nchunk = (tiles + nsub - 1) / nsub;
}

size_t sla_workspace_bytes(int mode, int NF, int N, int heads) {
    int nsub, nchunk;
    sla_plan(N, nsub, nchunk);
    const size_t part = (size_t)NF * nchunk * heads * SLA_PART * 4;
    const size_t ctx = (size_t)NF * heads * 32 * 32 * (mode == MODE_F32 ? 4 : 2);
    return ((part + 255) / 256) * 256 + ((ctx + 255) / 256) * 256;
}

template <int MODE, int TMO>
static hipError_t launch_sla_out_t(const SlaArgs& a, hipStream_t st) {
    using M = Mma<MODE>;
    constexpr int RSQ = (MODE == MODE_F32) ? 160 : 80;
    constexpr int RSO = 256 * M::ES + 16;
    const size_t lds = std::max<size_t>((size_t)(64 + 256) * ROW_STRIDE, (size_t)4 * 64 * RSQ + (size_t)64 * RSO);
    auto kfn = sla_out_kernel<MODE, TMO>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int tiles = (a.N + 63) / 64;
    hipLaunchKernelGGL(kfn, dim3(a.NF * tiles), dim3(256), lds, st, a);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_sla_m(SlaArgs a, hipStream_t st) {
    using M = Mma<MODE>;
    constexpr int NCHN = 64 / M::KC;
    constexpr int RSE = NCHN * 64 + 16;
    sla_plan(a.N, a.nsub, a.nchunk);
    const size_t part_bytes = (((size_t)a.NF * a.nchunk * a.heads * SLA_PART * 4) + 255) / 256 * 256;
    a.part = reinterpret_cast<float*>(a.workspace);
    a.ctxT = reinterpret_cast<char*>(a.workspace) + part_bytes;
    const size_t lds1 = 1024 + (size_t)128 * ROW_STRIDE + (size_t)64 * RSE;
    hipLaunchKernelGGL(sla_ctx_kernel<MODE>, dim3(a.NF * a.nchunk, a.heads), dim3(256), lds1, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sla_combine_kernel<MODE>, dim3(a.NF * a.heads), dim3(256), 0, st, a.part, a.ctxT, a.nchunk, a.heads);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (a.C <= 64) return launch_sla_out_t<MODE, 1>(a, st);
    if (a.C <= 128) return launch_sla_out_t<MODE, 2>(a, st);
    if (a.C <= 256) return launch_sla_out_t<MODE, 4>(a, st);
    if (a.C <= 512) return launch_sla_out_t<MODE, 8>(a, st);
    return hipErrorInvalidValue;
}

hipError_t launch_sla(int mode, SlaArgs a, hipStream_t st) {
    a.CPad = conv_cin_pad(mode, a.C);
    return mode == MODE_F32 ? launch_sla_m<MODE_F32>(a, st) : launch_sla_m<MODE_BF16>(a, st);
}

}  // namespace vdx
