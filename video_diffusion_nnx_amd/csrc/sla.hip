// SpatialLinearAttention (+ Residual, no-op PreNorm) on MFMA (gfx950).
//
// Reference: /root/reference/modules.py:64-129 wrapped by Residual(PreNorm(...)) (unet3d.py:170-178).
// Per frame n and head h (N = H*W pixels, D = 32):
//   q = softmax_D(x Wq)   (over the 32 channels of the head; the `* scale` is dead code)
//   k = softmax_N(x Wk)   (over the N pixels)
//   ctx[d,e] = sum_n k[d,n] v[e,n] ;  out[e,n] = sum_d ctx[d,e] q[d,n] ;  y = to_out(out) + x
// Three launches, none of which writes q/k/v to HBM:
//   sla_ctx_kernel     (frame, pixel-chunk, head; heads of a chunk co-scheduled on one XCD): K/V projection of 64-pixel sub-tiles, ONLINE softmax over
//                      pixels (running max / sum / rescaled 32x32 context in the accumulators) -> partial
//   sla_combine_kernel (frame, head): merges the chunk partials, normalises, emits ctx^T in the MMA type
//   sla_out_kernel     (frame, 64 pixels): Q projection of all heads, softmax over D with wavefront
//                      shuffles, out = ctx^T q, to_out GEMM accumulated in registers, + residual
#include "vdx_common.h"
#include "vdx_internal.h"
#include <stdlib.h>

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
    // XCD-aware block order: workgroup ids are dealt round-robin to the 8 XCDs, so inside each group of 8*heads ids the
    // blocks {j, j+8, j+16, ...} share an XCD (and its L2) and run at the same time: give them the heads of ONE pixel
    // chunk, so x is fetched from HBM once instead of once per head.
    const int gsz = 8 * P.heads;
    const int grp = blockIdx.x / gsz, j = blockIdx.x % gsz;
    const int h = j >> 3;
    const int cidx = grp * 8 + (j & 7);
    if (cidx >= P.NF * P.nchunk) return;               // uniform across the workgroup
    const int n = cidx / P.nchunk, chunk = cidx % P.nchunk;
    const int R = P.nsub * 64;
    const size_t xbase = (size_t)n * P.N * P.C;         // element offset of frame n in x
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
            if (r0 + row < P.N && c < P.C) xpre[u] = load4_f32_or_bf16(P.x, xbase + (size_t)(r0 + row) * P.C + c, P.io_bf16);
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
    const size_t xbase = (size_t)n * P.N * P.C;         // element offset of frame n in x and y
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
            if (r0 + row < P.N && c < P.C) v = load4_f32_or_bf16(P.x, xbase + (size_t)(r0 + row) * P.C + c, P.io_bf16);
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
            const float4 xr = load4_f32_or_bf16(P.x, xbase + (size_t)row * P.C + co, P.io_bf16);
            float4 v;
            v.x = oacc[tmo][tn][0] + xr.x; v.y = oacc[tmo][tn][1] + xr.y;
            v.z = oacc[tmo][tn][2] + xr.z; v.w = oacc[tmo][tn][3] + xr.w;
            store4_f32_or_bf16(P.y, xbase + (size_t)row * P.C + co, v, P.io_bf16);
        }
    }
}

// ---- one wave per head: the kernels used when heads == 8 and the x tile fits the LDS double buffer -----------------
// Workgroup = (frame, chunk of nsub 64-pixel sub-tiles), 8 waves, wave h = head h.  The fp32 x sub-tile is fetched once for
// all heads (registers one sub-tile ahead -> LDS ring of 2), every head's projection weights stay in registers as MFMA
// fragments for the whole chunk, and the projections are oriented so that their accumulators are directly the operands
// of the following K=16 MFMA (Mma::mma16): no q/k/v transposes through LDS, one barrier per sub-tile.

template <int MODE, int NKT, bool IO16>
struct SlaTile {
    using M = Mma<MODE>;
    static_assert(!IO16 || MODE == MODE_BF16, "bf16 activation storage implies bf16 MFMA operands");
    static constexpr int PCH = IO16 ? 8 : 4;                  // channels per 16-byte piece (IO16: x is a bf16 tensor)
    static constexpr int APIECES = M::KT / PCH;               // pieces per pixel per K tile
    static constexpr int XP = 64 * APIECES * NKT / 512;       // pieces per thread
    static constexpr int PLANE = 64 * ROW_STRIDE;             // one K tile of a sub-tile: [64 pixels][ROW_STRIDE]
    static constexpr int BUF = NKT * PLANE;
    float4 xpre[XP];                                          // IO16: 16 raw bytes (8 bf16), copied to the bf16 LDS tile as they are
    __device__ __forceinline__ void fetch(const float* x, size_t xbase, int io_bf16, int r0, int N, int C, int tid) {
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int i = tid + 512 * u;
            const int kt = i / (64 * APIECES), rem = i % (64 * APIECES);
            const int row = rem / APIECES, c = kt * M::KT + (rem % APIECES) * PCH;
            xpre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + row < N && c < C) {
                if (IO16) xpre[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(x) + (xbase + (size_t)(r0 + row) * C + c) * 2);
                else xpre[u] = load4_f32_or_bf16(x, xbase + (size_t)(r0 + row) * C + c, io_bf16);
            }
        }
    }
    __device__ __forceinline__ void put(char* xs, int tid) const {
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int i = tid + 512 * u;
            const int kt = i / (64 * APIECES), rem = i % (64 * APIECES);
            char* dst = xs + kt * PLANE + (rem / APIECES) * ROW_STRIDE;
            if (IO16) *reinterpret_cast<float4*>(dst + (rem % APIECES) * 16) = xpre[u];
            else M::store4(dst, (rem % APIECES) * 4, xpre[u]);
        }
    }
};

// online-softmax step of sla_ctx8 over one 32-pixel half sub-tile: channel d = dt*16 + lp lives in lane lp, its pixels in
// (tn, q, r).  e = exp(a - m) is evaluated as exp2(a * log2e - m * log2e): one FMA + one v_exp per element.
// FULL = every pixel of the sub-tile exists (no masks); else `nleft` = pixels left counting from this lane's (tn = 0, r = 0).
template <bool FULL>
__device__ __forceinline__ void ctx8_softmax_step(f32x4 (&acc)[2][4], float (&m_run)[2], float (&s_run)[2], f32x4 (&cacc)[2][2], int nleft) {
    constexpr float L2E = 1.44269504088896f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        if (!FULL) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) if (tn * 16 + r >= nleft) acc[tn][dt][r] = -1e30f;
        }
        float mx = fmaxf(fmaxf(fmaxf(acc[0][dt][0], acc[0][dt][1]), fmaxf(acc[0][dt][2], acc[0][dt][3])),
                         fmaxf(fmaxf(acc[1][dt][0], acc[1][dt][1]), fmaxf(acc[1][dt][2], acc[1][dt][3])));
        mx = max_q(mx);
        const float mn = fmaxf(m_run[dt], mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run[dt] - mn) * L2E);
        m_run[dt] = mn;
        const float nb = -mn * L2E;
        float ssum = 0.f;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float e = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[tn][dt][r], L2E, nb));
                if (!FULL) e = (tn * 16 + r >= nleft) ? 0.f : e;
                acc[tn][dt][r] = e;
                ssum += e;
            }
        s_run[dt] = s_run[dt] * alpha + reduce_q(ssum);
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int r = 0; r < 4; ++r) cacc[et][dt][r] *= alpha;
    }
}

template <int MODE, int NKT, bool IO16>
__global__ __launch_bounds__(512) void sla_ctx8_kernel(const SlaArgs P) {
    using M = Mma<MODE>;
    using T = SlaTile<MODE, NKT, IO16>;
    constexpr int RS = ROW_STRIDE;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // xs[2][NKT][64][RS]
    const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
    const int lp = lane & 15, q = lane >> 4;
    const int n = blockIdx.x / P.nchunk, chunk = blockIdx.x % P.nchunk;
    const int R = P.nsub * 64;
    const size_t xbase = (size_t)n * P.N * P.C;         // element offset of frame n in x

    // this head's K (tm 0,1) and V (tm 2,3) projection rows as B-operand fragments, resident for the whole chunk
    uint4 wf[NKT][2][4];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                const char* base = reinterpret_cast<const char*>(tm < 2 ? P.wk : P.wv);
                wf[kt][ch][tm] = *reinterpret_cast<const uint4*>(base + ((size_t)(h * 32 + (tm & 1) * 16 + lp) * P.CPad + kt * M::KT) * M::ES + ch * 64 + q * 16);
            }
    float m_run[2] = {-1e30f, -1e30f}, s_run[2] = {0.f, 0.f};
    f32x4 cacc[2][2];                                  // ctx^T tiles [et][dt]: lane (lp, q) = (d = dt*16+lp, e = et*16+4q+r)
#pragma unroll
    for (int i = 0; i < 2; ++i) { cacc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; cacc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    T t;
    t.fetch(P.x, xbase, P.io_bf16, chunk * R, P.N, P.C, tid);
    t.put(smem, tid);
    __syncthreads();
    for (int sub = 0; sub < P.nsub; ++sub) {
        const int r0 = chunk * R + sub * 64;
        if (r0 >= P.N) break;                          // uniform across the workgroup
        const char* xs = smem + (sub & 1) * T::BUF;
        const bool more = (sub + 1 < P.nsub) && (r0 + 64 < P.N);
        if (more) t.fetch(P.x, xbase, P.io_bf16, r0 + 64, P.N, P.C, tid);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {               // two 32-pixel online-softmax steps per sub-tile
            f32x4 acc[2][4];                           // [pixel tile][k0 k1 v0 v1]: lane (lp, q) = (channel lp, pixels 4q+r)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    uint4 xa[2];
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        xa[tn] = *reinterpret_cast<const uint4*>(xs + kt * T::PLANE + ((hf * 2 + tn) * 16 + lp) * RS + ch * 64 + q * 16);
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                        for (int tm = 0; tm < 4; ++tm) M::mma(acc[tn][tm], xa[tn], wf[kt][ch][tm]);
                }
            if (r0 + 64 <= P.N) ctx8_softmax_step<true>(acc, m_run, s_run, cacc, 0);                  // uniform branch
            else ctx8_softmax_step<false>(acc, m_run, s_run, cacc, P.N - (r0 + hf * 32 + 4 * q));
            // ctx^T[e, d] += sum_n v[e, n] e_k[d, n]: both operands are the accumulators above (k = pixel 4q+r);
            // invalid pixels contribute v = 0 (x rows zero-filled, no bias) and e_k = 0
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int et = 0; et < 2; ++et)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) M::mma16(cacc[et][dt], acc[tn][2 + et], acc[tn][dt]);
        }
        if (more) t.put(smem + ((sub + 1) & 1) * T::BUF, tid);
        __syncthreads();
    }
    if (P.nchunk == 1) {
        // the frame is one chunk (many frames: B = 64): this wave holds the whole context of (frame, head) -- normalise and write
        // ctxT[n][h][e][d] here, no partials and no combine launch
        char* ct = reinterpret_cast<char*>(P.ctxT) + (size_t)(n * P.heads + h) * 32 * 32 * M::ES;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const float inv = 1.0f / s_run[dt];
#pragma unroll
            for (int et = 0; et < 2; ++et)
#pragma unroll
                for (int r = 0; r < 4; ++r) M::store1(ct + (size_t)(et * 16 + 4 * q + r) * 32 * M::ES, dt * 16 + lp, cacc[et][dt][r] * inv);
        }
        return;
    }
    float* part = P.part + ((size_t)(n * P.nchunk + chunk) * P.heads + h) * SLA_PART;
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            *reinterpret_cast<float4*>(part + (dt * 16 + lp) * 32 + et * 16 + 4 * q) =
                make_float4(cacc[et][dt][0], cacc[et][dt][1], cacc[et][dt][2], cacc[et][dt][3]);
    if (q == 0) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) { part[1024 + dt * 16 + lp] = m_run[dt]; part[1056 + dt * 16 + lp] = s_run[dt]; }
    }
}

// q projection + softmax over D + out = ctx^T q per head in registers; the heads meet in LDS (os) for the to_out GEMM,
// which the 8 waves split by (output-channel tile, pixel tile).  TMO x TNO = tiles per wave: C/16 * 4 / 8.
template <int MODE, int NKT, int TMO, int TNO, bool IO16>
__global__ __launch_bounds__(512, NKT == 1 ? 4 : 2) void sla_out8_kernel(const SlaArgs P) {
    using M = Mma<MODE>;
    using T = SlaTile<MODE, NKT, IO16>;
    constexpr int RS = ROW_STRIDE, KC = M::KC, HD = 256;
    constexpr int RSO = HD * M::ES + 16;
    constexpr int NCHO = HD / KC;
    constexpr bool WO_RES = (TMO * NCHO <= 16);        // to_out fragments register-resident

    extern __shared__ __attribute__((aligned(16))) char smem[];      // xs[2][NKT][64][RS] | os[64][RSO] | IO16: ys[64][RSY] (fp32)
    char* os = smem + 2 * T::BUF;
    constexpr int RSY = NKT * M::KT * 4 + 16;
    char* ys = os + 64 * RSO;
    const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
    const int lp = lane & 15, q = lane >> 4;
    const int n = blockIdx.x / P.nchunk, chunk = blockIdx.x % P.nchunk;
    const int R = P.nsub * 64;
    const size_t xbase = (size_t)n * P.N * P.C;         // element offset of frame n in x and y

    uint4 wqf[NKT][2][2];                              // Wq rows d = tm*16+lp of this head (A operand)
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
                wqf[kt][ch][tm] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(P.wq) +
                    ((size_t)(h * 32 + tm * 16 + lp) * P.CPad + kt * M::KT) * M::ES + ch * 64 + q * 16);
    f32x4 cf[2][2];                                    // ctx^T[e = et*16+lp][d = dt*16+4q..+3] (A operand of mma16)
    {
        const char* ct = reinterpret_cast<const char*>(P.ctxT) + (size_t)(n * P.heads + h) * 32 * 32 * M::ES;
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) cf[et][dt] = M::load_w4(ct + ((size_t)(et * 16 + lp) * 32 + dt * 16 + 4 * q) * M::ES);
    }
    // to_out tiles of this wave
    // (TNO 1 = the C = 32 form -- level 0 of the YAML-literal config_v2_2 --: two output-channel tiles x four pixel tiles over the 8 waves)
    const int cot0 = (TNO == 4) ? h * TMO : (TNO == 2) ? (h & 3) : (h & 1);   // first output-channel tile
    const int tn0 = (TNO == 4) ? 0 : (TNO == 2) ? 2 * (h >> 2) : (h >> 1);     // first pixel tile
    const char* wo = reinterpret_cast<const char*>(P.wo);
    uint4 wof[WO_RES ? TMO : 1][WO_RES ? NCHO : 1];
    if (WO_RES) {
#pragma unroll
        for (int tmo = 0; tmo < TMO; ++tmo)
#pragma unroll
            for (int ch = 0; ch < NCHO; ++ch)
                wof[tmo][ch] = *reinterpret_cast<const uint4*>(wo + (size_t)((cot0 + tmo) * 16 + lp) * HD * M::ES + ch * 64 + q * 16);
    }

    T t;
    t.fetch(P.x, xbase, IO16 ? 1 : P.io_bf16, chunk * R, P.N, P.C, tid);
    t.put(smem, tid);
    __syncthreads();
    for (int sub = 0; sub < P.nsub; ++sub) {
        const int r0 = chunk * R + sub * 64;
        if (r0 >= P.N) break;
        const char* xs = smem + (sub & 1) * T::BUF;
        const bool more = (sub + 1 < P.nsub) && (r0 + 64 < P.N);
        float4 xcur[IO16 ? T::XP : 1];                  // IO16: this sub-tile's raw bf16 pieces = the residual of its output (no re-read)
        if (IO16) {
#pragma unroll
            for (int u = 0; u < T::XP; ++u) xcur[u] = t.xpre[u];
        }
        if (more) t.fetch(P.x, xbase, IO16 ? 1 : P.io_bf16, r0 + 64, P.N, P.C, tid);
        // q[d, n] of this head: lane (lp, q) = (pixel lp, channels 4q+r)
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                uint4 xb[4];
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) xb[tn] = *reinterpret_cast<const uint4*>(xs + kt * T::PLANE + (tn * 16 + lp) * RS + ch * 64 + q * 16);
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) M::mma(acc[tm][tn], wqf[kt][ch][tm], xb[tn]);
            }
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {               // softmax over the 32 channels of the head, per pixel
            float mx = -1e30f;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[tm][tn][r]);
            mx = max_q(mx);
            const float nmx = -mx * 1.44269504088896f;  // exp(a - max) = exp2(fma(a, log2 e, -max log2 e)): one FMA per element (vector-issue bound)
            float sum = 0.f;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[tm][tn][r] = __builtin_amdgcn_exp2f(fmaf(acc[tm][tn][r], 1.44269504088896f, nmx)); sum += acc[tm][tn][r]; }
            const float inv = __builtin_amdgcn_rcpf(reduce_q(sum));     // v_rcp_f32 (1 ulp) instead of the IEEE division sequence
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[tm][tn][r] *= inv;
        }
        // out[e, n] = sum_d ctx^T[e, d] q[d, n]  ->  os[n][h*32 + e]
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) M::mma16(o, cf[et][dt], acc[dt][tn]);
                M::store4(os + (tn * 16 + lp) * RSO, h * 32 + et * 16 + 4 * q, make_float4(o[0], o[1], o[2], o[3]));
            }
        __syncthreads();
        // to_out + residual
        f32x4 oacc[TMO][TNO];
#pragma unroll
        for (int i = 0; i < TMO; ++i)
#pragma unroll
            for (int j = 0; j < TNO; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < NCHO; ++ch) {
            uint4 bf[TNO];
#pragma unroll
            for (int tn = 0; tn < TNO; ++tn) bf[tn] = *reinterpret_cast<const uint4*>(os + ((tn0 + tn) * 16 + lp) * RSO + ch * 64 + q * 16);
#pragma unroll
            for (int tmo = 0; tmo < TMO; ++tmo) {
                uint4 a;
                if (WO_RES) a = wof[tmo][ch];
                else a = *reinterpret_cast<const uint4*>(wo + (size_t)((cot0 + tmo) * 16 + lp) * HD * M::ES + ch * 64 + q * 16);
#pragma unroll
                for (int tn = 0; tn < TNO; ++tn) M::mma(oacc[tmo][tn], a, bf[tn]);
            }
        }
        if constexpr (IO16) {
            // bf16 tensors: the out tile goes through LDS (ys, fp32) so that every thread stores ONE whole 16-byte piece in fetch order
            // (1 KB contiguous per wave instruction) with the residual taken from the fetched piece in registers.  (First form: each
            // lane loaded the 8 residual bytes of its 4 channels right before storing 8 bytes -- a dependent L2 / HBM round trip per
            // tile in the epilogue and 32-byte row fragments per instruction: 52 % of wave life parked.)
#pragma unroll
            for (int tmo = 0; tmo < TMO; ++tmo)
#pragma unroll
                for (int tn = 0; tn < TNO; ++tn)
                    *reinterpret_cast<f32x4*>(ys + ((tn0 + tn) * 16 + lp) * RSY + ((cot0 + tmo) * 16 + 4 * q) * 4) = oacc[tmo][tn];
            if (more) t.put(smem + ((sub + 1) & 1) * T::BUF, tid);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < T::XP; ++u) {
                const int i = tid + 512 * u;
                const int kt = i / (64 * T::APIECES), rem = i % (64 * T::APIECES);
                const int row = rem / T::APIECES, c = kt * M::KT + (rem % T::APIECES) * 8;
                if (r0 + row >= P.N || c >= P.C) continue;
                const float4 o4 = *reinterpret_cast<const float4*>(ys + row * RSY + c * 4), o5 = *reinterpret_cast<const float4*>(ys + row * RSY + c * 4 + 16);
                const unsigned x0 = __float_as_uint(xcur[u].x), x1 = __float_as_uint(xcur[u].y), x2 = __float_as_uint(xcur[u].z), x3 = __float_as_uint(xcur[u].w);
                uint4 w;
                w.x = pack_bf16x2(o4.x + __uint_as_float(x0 << 16), o4.y + __uint_as_float(x0 & 0xFFFF0000u));
                w.y = pack_bf16x2(o4.z + __uint_as_float(x1 << 16), o4.w + __uint_as_float(x1 & 0xFFFF0000u));
                w.z = pack_bf16x2(o5.x + __uint_as_float(x2 << 16), o5.y + __uint_as_float(x2 & 0xFFFF0000u));
                w.w = pack_bf16x2(o5.z + __uint_as_float(x3 << 16), o5.w + __uint_as_float(x3 & 0xFFFF0000u));
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(P.y) + (xbase + (size_t)(r0 + row) * P.C + c) * 2) = w;
            }
            // (the next iteration writes ys only after its own first barrier, which every wave reaches after these reads)
        } else {
#pragma unroll
            for (int tmo = 0; tmo < TMO; ++tmo) {
                const int co = (cot0 + tmo) * 16 + 4 * q;
#pragma unroll
                for (int tn = 0; tn < TNO; ++tn) {
                    const int row = r0 + (tn0 + tn) * 16 + lp;
                    if (row >= P.N) continue;
                    const float4 xr = load4_f32_or_bf16(P.x, xbase + (size_t)row * P.C + co, P.io_bf16);
                    store4_f32_or_bf16(P.y, xbase + (size_t)row * P.C + co,
                                       make_float4(oacc[tmo][tn][0] + xr.x, oacc[tmo][tn][1] + xr.y, oacc[tmo][tn][2] + xr.z, oacc[tmo][tn][3] + xr.w), P.io_bf16);
                }
            }
            if (more) t.put(smem + ((sub + 1) & 1) * T::BUF, tid);
            __syncthreads();
        }
    }
}

// ---- C = 64, bf16 tensors, >= 128 frames: the second half of the block with one WAVE per 64 pixels, all heads in the wave ("sla_out_w") ----
// sla_out8_kernel (one wave per head) moves everything through LDS: the x tile, the per-head outputs (os), the out tile (ys), two barriers per
// 64 pixels -- LDS array 47 % active with 39 % bank conflicts, 40 % of wave life parked (profiles/r03_pmc_step.md).  The scheme of
// attention_w_kernel applies: a persistent workgroup per CU keeps Wq (256 x 64 bf16 = 32 KB) and the to_out image (64 x 256 = 32 KB, rows and K
// permuted as there) in LDS and walks whole frames; the context of the current frame (8 heads x 32 x 32 bf16 = 16 KB, written by the first
// half) sits in a double-buffered LDS image, ONE barrier per frame.  Each wave owns groups of 64 pixels: x rows straight into B fragments (one
// group ahead; they are the residual), per head q^T = Wq_h x^T (16 MFMAs), softmax over the head's 32 channels in the accumulators, out^T =
// ctx^T q as K = 16 MFMAs on the accumulators, its result packed as the B fragment of the to_out MFMAs -- no activation byte in LDS.
template <int DUMMY = 0>
__global__ __launch_bounds__(512) void sla_out_w_kernel(const SlaArgs P, const int frames_per_block) {
    using M = Mma<MODE_BF16>;
    constexpr int HD = 256, CRS = 72;                  // ctx image row stride (bytes): 32 d x 2 + 8 (b64 reads of 16 rows: at most 2-way conflicts)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Wq = smem;                                   // [256 rows][128 B]: 16-byte chunk c of row r at r * 128 + 16 * (c ^ (r & 7))
    char* Wo = Wq + HD * 128;                          // [64 rows][512 B]: as attention_w_kernel
    char* Cx = Wo + 64 * 512;                          // [2][256 rows (h * 32 + e)][CRS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lp = lane & 15, q = lane >> 4;
    const int f0 = blockIdx.x * frames_per_block, f1 = min(f0 + frames_per_block, P.NF);
    if (f0 >= f1) return;

    for (int i = tid; i < HD * 8; i += 512) {
        const int r = i >> 3, c = i & 7;
        *reinterpret_cast<uint4*>(Wq + r * 128 + 16 * (c ^ (r & 7))) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(P.wq) + (size_t)r * 128 + c * 16);
    }
    for (int i = tid; i < 64 * 32; i += 512) {
        const int R = i >> 5, c = i & 31, hh = c >> 2, qq = c & 3;
        const int tile = R >> 4, ri = R & 15;
        const int co = 32 * (tile >> 1) + 8 * (ri >> 2) + 4 * (tile & 1) + (ri & 3);
        const char* src = reinterpret_cast<const char*>(P.wo) + (size_t)co * 512 + hh * 64 + qq * 8;
        const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 32);
        *reinterpret_cast<uint4*>(Wo + R * 512 + 16 * (c ^ (R & 15))) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    // context image of frame f into buffer par: 256 rows x 64 bytes = 1024 pieces of 16 bytes, two per thread
    auto load_ctx = [&](int f, int par) __attribute__((always_inline)) {
        const char* src = reinterpret_cast<const char*>(P.ctxT) + (size_t)f * (8 * 32 * 32 * 2);
        uint4 v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) v[u] = *reinterpret_cast<const uint4*>(src + (size_t)(tid + 512 * u) * 16);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + 512 * u, row = i >> 2, pc = i & 3;
            char* d = Cx + par * (256 * CRS) + row * CRS + pc * 16;
            *reinterpret_cast<uint2*>(d) = make_uint2(v[u].x, v[u].y);
            *reinterpret_cast<uint2*>(d + 8) = make_uint2(v[u].z, v[u].w);
        }
    };
    load_ctx(f0, 0);
    __syncthreads();                                   // weights + the first context visible

    const int gpf = P.N >> 6;                          // groups of 64 pixels per frame
    const char* const xg = reinterpret_cast<const char*>(P.x);
    char* const yg = reinterpret_cast<char*>(P.y);
    const unsigned loff = (unsigned)(lp * 64 + q * 8) * 2u;          // this lane's bytes inside a 16-pixel tile: pixel lp, channels 8 q ..
    int wqo[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) wqo[ch] = lp * 128 + 16 * ((ch * 4 + q) ^ (lp & 7));
    const int woo = lp * 512;

    // the groups of this wave, frame by frame: (f, g) with g = wave_u, wave_u + 8, ...; `next` walks one group ahead for the prefetch
    uint4 xn[4][2];
    auto fetch = [&](int f, int g) __attribute__((always_inline)) {
        const char* p = xg + ((size_t)f * P.N + (size_t)g * 64) * 128 + loff;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) xn[tn][ch] = *reinterpret_cast<const uint4*>(p + tn * (16 * 128) + ch * 64);
    };
    uint4 wcur[2][2], wnext[2][2];
    auto rd = [&](uint4 (&w)[2][2], const char* base) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) w[ch][t] = *reinterpret_cast<const uint4*>(base + t * 16 * 128 + wqo[ch]);
    };
    rd(wcur, Wq);
    if (wave_u < gpf) fetch(f0, wave_u);
    for (int f = f0; f < f1; ++f) {
        const int par = (f - f0) & 1;
        if (f + 1 < f1) load_ctx(f + 1, par ^ 1);     // (the other buffer was last read before the previous frame's barrier)
        const char* cimg = Cx + par * (256 * CRS);
        for (int g = wave_u; g < gpf; g += 8) {
            uint4 xc[4][2];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) { xc[tn][0] = xn[tn][0]; xc[tn][1] = xn[tn][1]; }
            if (g + 8 < gpf) fetch(f, g + 8);
            else if (f + 1 < f1 && wave_u < gpf) fetch(f + 1, wave_u);
            f32x4 yacc[4][4];                          // [pixel tile][channel tile = 2 wc + tm]
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int tile = 0; tile < 4; ++tile) yacc[tn][tile] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int h = 0; h < 8; ++h) {
                // q^T[d, px] of this head (fragments in wcur); the to_out fragments go to wnext meanwhile
                f32x4 acc[2][4];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                {
                    const char* woh = Wo + woo + 16 * ((4 * h + q) ^ lp);
#pragma unroll
                    for (int tile = 0; tile < 4; ++tile) wnext[tile >> 1][tile & 1] = *reinterpret_cast<const uint4*>(woh + tile * 8192);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 4; ++tn) M::mma(acc[tm][tn], wcur[ch][tm], xc[tn][ch]);
                // ctx^T fragments of this head: A operand of the K = 16 product, lane (e = lp, q): d = dt * 16 + 4q .. + 3 (4 bf16)
                uint2 cf[2][2];
#pragma unroll
                for (int et = 0; et < 2; ++et)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) cf[et][dt] = *reinterpret_cast<const uint2*>(cimg + (h * 32 + et * 16 + lp) * CRS + dt * 32 + q * 8);
                __builtin_amdgcn_sched_barrier(0);
                rd(wcur, Wq + ((h + 1) & 7) * (32 * 128));     // the next head's (head 0 of the next group after head 7) q fragments
                // softmax over the 32 channels of the head, per pixel (lane (px, q) holds d = 16 tm + 4q + r)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    float mx = -1e30f;
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[tm][tn][r]);
                    mx = max_q(mx);
                    const float nmx = -mx * 1.44269504088896f;
                    float sum = 0.f;
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { acc[tm][tn][r] = __builtin_amdgcn_exp2f(fmaf(acc[tm][tn][r], 1.44269504088896f, nmx)); sum += acc[tm][tn][r]; }
                    const float inv = __builtin_amdgcn_rcpf(reduce_q(sum));
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[tm][tn][r] *= inv;
                }
                // out^T[e, px] = sum_d ctx^T[e, d] q[d, px]; lane (px, q): e = 16 et + 4q + r -> the to_out B fragment (K slots in accumulator order)
                uint4 of[4];
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    f32x4 o[2];
#pragma unroll
                    for (int et = 0; et < 2; ++et) {
                        o[et] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            const uint2 qb = make_uint2(pack_bf16x2(acc[dt][tn][0], acc[dt][tn][1]), pack_bf16x2(acc[dt][tn][2], acc[dt][tn][3]));
                            o[et] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, cf[et][dt]), __builtin_bit_cast(s16x4, qb), o[et], 0, 0, 0);
                        }
                    }
                    of[tn] = make_uint4(pack_bf16x2(o[0][0], o[0][1]), pack_bf16x2(o[0][2], o[0][3]), pack_bf16x2(o[1][0], o[1][1]), pack_bf16x2(o[1][2], o[1][3]));
                }
#pragma unroll
                for (int tile = 0; tile < 4; ++tile)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) M::mma(yacc[tn][tile], wnext[tile >> 1][tile & 1], of[tn]);
            }
            // + residual (the fetched rows), two 16-byte stores per pixel
            char* yp = yg + ((size_t)f * P.N + (size_t)g * 64) * 128 + loff;
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int wc = 0; wc < 2; ++wc) {
                    const uint4 r = xc[tn][wc];
                    const f32x4 a0 = yacc[tn][2 * wc], a1 = yacc[tn][2 * wc + 1];
                    uint4 w;
                    w.x = pack_bf16x2(a0[0] + __uint_as_float(r.x << 16), a0[1] + __uint_as_float(r.x & 0xFFFF0000u));
                    w.y = pack_bf16x2(a0[2] + __uint_as_float(r.y << 16), a0[3] + __uint_as_float(r.y & 0xFFFF0000u));
                    w.z = pack_bf16x2(a1[0] + __uint_as_float(r.z << 16), a1[1] + __uint_as_float(r.z & 0xFFFF0000u));
                    w.w = pack_bf16x2(a1[2] + __uint_as_float(r.w << 16), a1[3] + __uint_as_float(r.w & 0xFFFF0000u));
                    *reinterpret_cast<uint4*>(yp + tn * (16 * 128) + wc * 64) = w;
                }
        }
        __syncthreads();                               // the next frame's context is visible; everybody is done with this one
    }
}

// ---- wide levels (C >= 256): one workgroup per (head, chunk of frames), one wave per frame ------------------------------------------
// The head's q/k/v weight rows (96 x C bf16) are loaded into LDS once per workgroup; each wave then owns whole frames and needs no
// workgroup barrier, no partials and no combine pass: phase 1 walks the frame's pixels in steps of 16*TT with the k/v projections
// computed transposed (rows = pixels), so the softmax-over-pixels statistics are in-lane + quad reductions and e^T v is an MFMA over
// the pixel index with the accumulators as operands; the 32x32 context stays in registers.  Phase 2 walks the pixels again (x is hot
// in L2): q projection, softmax over D, out = ctx^T q -> O[row][head*32 + e] bf16.  to_out (+ residual) is a plain 1x1 conv_igemm.
template <bool IO16, int TT>
__global__ __launch_bounds__(512) void sla_head_kernel(const SlaArgs P, void* __restrict__ O, const int frames_per_block, const int nchunks) {
    using M = Mma<MODE_BF16>;
    constexpr int D = 32;
    constexpr float L2E = 1.44269504088896f;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // W_h [q 32 | k 32 | v 32 rows][C * 2 + 32]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lp = lane & 15, q = lane >> 4;
    // XCD-aware decode: workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2), so ids {c, c + 8, ..., c + 56}
    // of one group of 64 -- the 8 heads of ONE frame chunk -- land on the same XCD at about the same time and x comes from HBM once
    // instead of once per head
    const int h = (blockIdx.x >> 3) & 7, HD = P.heads * D;
    const int chunk_id = (blockIdx.x >> 6) * 8 + (blockIdx.x & 7);
    if (chunk_id >= nchunks) return;                                  // (uniform)
    const int RSW = P.C * 2 + 32;
    const int cpr = P.C / 8;
    for (int i = tid; i < 96 * cpr; i += 512) {
        const int row = i / cpr, pc = i - row * cpr;
        const int part = row >> 5, rr = row & 31;
        const char* src = reinterpret_cast<const char*>(part == 0 ? P.wq : part == 1 ? P.wk : P.wv);
        *reinterpret_cast<uint4*>(smem + row * RSW + pc * 16) =
            *reinterpret_cast<const uint4*>(src + ((size_t)(h * D + rr) * P.CPad) * 2 + pc * 16);
    }
    __syncthreads();
    const int nkt = P.C / 32;
    const char* wrow = smem + lp * RSW + q * 16;
    const int n_end = min(P.NF, (chunk_id + 1) * frames_per_block);
    auto xload = [&](size_t e) -> uint4 {                              // 8 consecutive channels at element offset e as bf16
        if (IO16) return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(P.x) + e * 2);
        const float4 a = *reinterpret_cast<const float4*>(P.x + e), b = *reinterpret_cast<const float4*>(P.x + e + 4);
        return make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
    };
    for (int n = chunk_id * frames_per_block + w; n < n_end; n += 8) {
        const size_t xbase = (size_t)n * P.N * P.C + (size_t)lp * P.C + 8 * q;     // token lp of the frame, channel group q
        // x fragments run through a 4-deep register ring: global loads are issued four K steps ahead of their use.  The walk is
        // pixels 0..N (phase 1) and then 0..N again (phase 2), so the ring simply wraps once.
        int ftok = 0, fk = 0, fpass = 0;
        uint4 ring[4][TT];
        auto fetch = [&](uint4 (&dst)[TT]) {
            if (fpass < 2) {
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) dst[tt] = xload(xbase + (size_t)(ftok + tt * 16) * P.C + fk * 32);
            }
            if (++fk == nkt) { fk = 0; ftok += 16 * TT; if (ftok >= P.N) { ftok = 0; ++fpass; } }
        };
        // ---------------- phase 1: ctx[d, e] = sum_n softmax_n(k)[d, n] v[e, n] ----------------
        float mrun[2] = {-1e30f, -1e30f}, srun[2] = {0.f, 0.f};                        // per d = t*16 + lp (replicated over q)
        f32x4 ctx[2][2];                                                               // [dt][et]: rows d = 4q+r, cols e = lp
#pragma unroll
        for (int i = 0; i < 2; ++i) { ctx[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ctx[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        ftok = 0; fk = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) fetch(ring[u]);
        for (int tok0 = 0; tok0 < P.N; tok0 += 16 * TT) {
            f32x4 ak[TT][2], av[TT][2];                                                // rows = pixels 4q+r, cols = d / e = lp
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                for (int t = 0; t < 2; ++t) { ak[tt][t] = f32x4{0.f, 0.f, 0.f, 0.f}; av[tt][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            for (int kt = 0; kt < nkt; kt += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const uint4 wk = *reinterpret_cast<const uint4*>(wrow + (32 + t * 16) * RSW + (kt + u) * 64);
                        const uint4 wv = *reinterpret_cast<const uint4*>(wrow + (64 + t * 16) * RSW + (kt + u) * 64);
#pragma unroll
                        for (int tt = 0; tt < TT; ++tt) { M::mma(ak[tt][t], ring[u][tt], wk); M::mma(av[tt][t], ring[u][tt], wv); }
                    }
                    fetch(ring[u]);
                }
            }
            float al[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float mx = -1e30f;
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) mx = fmaxf(mx, fmaxf(fmaxf(ak[tt][t][0], ak[tt][t][1]), fmaxf(ak[tt][t][2], ak[tt][t][3])));
                mx = max_q(mx);
                const float mn = fmaxf(mrun[t], mx);
                al[t] = __builtin_amdgcn_exp2f((mrun[t] - mn) * L2E);
                mrun[t] = mn;
                float sum = 0.f;
#pragma unroll
                for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ak[tt][t][r] = __builtin_amdgcn_exp2f((ak[tt][t][r] - mn) * L2E); sum += ak[tt][t][r]; }
                srun[t] = srun[t] * al[t] + reduce_q(sum);
            }
            if (tok0 > 0) {                                                            // rescale: alpha of row d = 4q+r lives in lane 4q+r
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((4 * q + r) * 4, __builtin_bit_cast(int, al[dt])));
                        ctx[dt][0][r] *= a; ctx[dt][1][r] *= a;
                    }
            }
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int et = 0; et < 2; ++et) M::mma16(ctx[dt][et], ak[tt][dt], av[tt][et]);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const float inv = __builtin_amdgcn_rcpf(srun[dt]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((4 * q + r) * 4, __builtin_bit_cast(int, inv)));
                ctx[dt][0][r] *= a; ctx[dt][1][r] *= a;
            }
        }
        // ---------------- phase 2: out[e, n] = sum_d ctx[d, e] softmax_d(q)[d, n] ----------------
        const size_t obase = ((size_t)n * P.N + lp) * HD + h * D + 4 * q;
        for (int tok0 = 0; tok0 < P.N; tok0 += 16 * TT) {
            f32x4 aq[TT][2];                                                           // rows d = 4q+r, cols = pixels lp
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) { aq[tt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; aq[tt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            for (int kt = 0; kt < nkt; kt += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const uint4 wq = *reinterpret_cast<const uint4*>(wrow + (t * 16) * RSW + (kt + u) * 64);
#pragma unroll
                        for (int tt = 0; tt < TT; ++tt) M::mma(aq[tt][t], wq, ring[u][tt]);
                    }
                    fetch(ring[u]);
                }
            }
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                float mx = fmaxf(fmaxf(fmaxf(aq[tt][0][0], aq[tt][0][1]), fmaxf(aq[tt][0][2], aq[tt][0][3])),
                                 fmaxf(fmaxf(aq[tt][1][0], aq[tt][1][1]), fmaxf(aq[tt][1][2], aq[tt][1][3])));
                mx = max_q(mx);
                float sum = 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { aq[tt][t][r] = __builtin_amdgcn_exp2f((aq[tt][t][r] - mx) * L2E); sum += aq[tt][t][r]; }
                const float inv = __builtin_amdgcn_rcpf(reduce_q(sum));
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) aq[tt][t][r] *= inv;
#pragma unroll
                for (int et = 0; et < 2; ++et) {
                    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
                    M::mma16(o, ctx[0][et], aq[tt][0]);
                    M::mma16(o, ctx[1][et], aq[tt][1]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(O) + (obase + (size_t)(tok0 + tt * 16) * HD + et * 16) * 2) =
                        make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
                }
            }
        }
    }
}

// instrumentation (vdx.h: vdx_set_launch_hook): algorithmic work of the two halves of a SpatialLinearAttention block over px = NF * N
// pixels.  ctx half: k, v projections 2 C 512 + context 2 * 256 * 32 FLOP per pixel, reads x; out half: q projection
// 2 C 256 + ctx^T q 2 * 256 * 32 + to_out 2 * 256 * C (when `with_out`), reads x, writes y (or the per-head output [px][256] bf16).
namespace {
struct SlaWork { double flops, bytes; };
SlaWork sla_work(const SlaArgs& a, int es_w, bool ctx_half, bool out_half, bool with_out) {
    const double px = (double)a.NF * a.N, eio = a.io_bf16 ? 2.0 : 4.0, ctx_bytes = (double)a.NF * a.heads * 32 * 32 * 4;
    SlaWork w{0.0, 0.0};
    if (ctx_half) { w.flops += px * (2.0 * a.C * 512 + 2.0 * 256 * 32); w.bytes += px * a.C * eio + ctx_bytes + es_w * 512.0 * a.C; }
    if (out_half) {
        w.flops += px * (2.0 * a.C * 256 + 2.0 * 256 * 32 + (with_out ? 2.0 * 256 * a.C : 0.0));
        w.bytes += px * a.C * eio + (with_out ? px * a.C * eio : px * 256 * 2.0) + (ctx_half ? 0.0 : ctx_bytes) + es_w * (256.0 * a.C + (with_out ? 256.0 * a.C : 0.0));
    }
    return w;
}
}  // namespace

hipError_t launch_sla_heads(SlaArgs a, void* O, hipStream_t st) {
    a.CPad = conv_cin_pad(MODE_BF16, a.C);
    if (a.heads != 8 || a.C % 128 || a.N % 16 || !O) return hipErrorInvalidValue;
    const size_t lds = (size_t)96 * (a.C * 2 + 32);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    long fpb = (a.NF * (long)a.heads + 255) / 256;       // one round of workgroups over the chip (one 8-wave workgroup per CU: ~200 VGPRs)
    fpb = std::max<long>(8, (fpb + 7) / 8 * 8);
    const long chunks = (a.NF + fpb - 1) / fpb;
    auto go = [&](auto kfn) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        const SlaWork sw = sla_work(a, 2, true, true, false);
        LaunchScope ls(st, "sla_head_kernel", sw.flops, sw.bytes, "<io16 %d, %d> C%d N%d NF%d", a.io_bf16, a.N % 64 == 0 ? 4 : 1, a.C, a.N, a.NF);
        hipLaunchKernelGGL(kfn, dim3((unsigned)((chunks + 7) / 8 * 64)), dim3(512), lds, st, a, O, (int)fpb, (int)chunks);
        return hipGetLastError();
    };
    if (a.N % 64 == 0) return a.io_bf16 ? go(sla_head_kernel<true, 4>) : go(sla_head_kernel<false, 4>);
    return a.io_bf16 ? go(sla_head_kernel<true, 1>) : go(sla_head_kernel<false, 1>);
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
    const SlaWork sw = sla_work(a, M::ES, false, true, true);
    LaunchScope ls(st, "sla_out_kernel", sw.flops, sw.bytes, "<%d, %d> C%d N%d NF%d io16 %d", MODE, TMO, a.C, a.N, a.NF, a.io_bf16);
    hipLaunchKernelGGL(kfn, dim3(a.NF * tiles), dim3(256), lds, st, a);
    return hipGetLastError();
}

#ifndef VDX_SLA_W
#define VDX_SLA_W 1
#endif
#ifndef VDX_SLA_W_MIN_N
#define VDX_SLA_W_MIN_N 2048
#endif
static bool sla_out_w_eligible(const SlaArgs& a) {
    // (frames of 1024 pixels = 2 groups per wave and frame: the per-frame context load + barrier cost more than the staging they replace: 103 vs 90 us)
    return VDX_SLA_W && a.io_bf16 && a.C == 64 && a.CPad == 64 && a.heads == 8 && a.N % 64 == 0 && a.N >= VDX_SLA_W_MIN_N && a.NF >= 128 && (size_t)a.NF * a.N * 128 < (1ull << 40);
}
static hipError_t launch_sla_out_w(const SlaArgs& a, hipStream_t st) {
    const size_t lds = 256 * 128 + 64 * 512 + 2 * 256 * 72;
    auto kfn = sla_out_w_kernel<0>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    const int blocks0 = std::min(a.NF, cus);
    const int fpb = (a.NF + blocks0 - 1) / blocks0;
    const int blocks = (a.NF + fpb - 1) / fpb;
    const SlaWork sw = sla_work(a, 2, false, true, true);
    LaunchScope ls(st, "sla_out_w_kernel", sw.flops, sw.bytes, "C%d N%d NF%d", a.C, a.N, a.NF);
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(512), lds, st, a, fpb);
    return hipGetLastError();
}

template <int MODE, int NKT, int TMO, int TNO, bool IO16>
static hipError_t launch_sla8_t(const SlaArgs& a, hipStream_t st) {
    using M = Mma<MODE>;
    using T = SlaTile<MODE, NKT, IO16>;
    const size_t lds_ctx = 2 * (size_t)T::BUF;
    const size_t lds_out = lds_ctx + (size_t)64 * (256 * M::ES + 16) + (IO16 ? (size_t)64 * (NKT * M::KT * 4 + 16) : 0);
    auto kc = sla_ctx8_kernel<MODE, NKT, IO16>;
    auto ko = sla_out8_kernel<MODE, NKT, TMO, TNO, IO16>;
    hipError_t e;
    if (lds_ctx > 64 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void*>(kc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ctx)) != hipSuccess) return e;
    if (lds_out > 64 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void*>(ko), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_out)) != hipSuccess) return e;
    {
        const SlaWork sw = sla_work(a, M::ES, true, false, false);
        LaunchScope ls(st, "sla_ctx8_kernel", sw.flops, sw.bytes, "<%d, %d, %d> C%d N%d NF%d nchunk%d", MODE, NKT, (int)IO16, a.C, a.N, a.NF, a.nchunk);
        hipLaunchKernelGGL(kc, dim3(a.NF * a.nchunk), dim3(512), lds_ctx, st, a);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (a.nchunk > 1) {                                       // (one chunk per frame: sla_ctx8_kernel wrote ctxT itself)
        LaunchScope ls(st, "sla_combine_kernel", 0.0, (double)a.NF * a.nchunk * a.heads * SLA_PART * 4, "<%d> NF%d nchunk%d", MODE, a.NF, a.nchunk);
        hipLaunchKernelGGL(sla_combine_kernel<MODE>, dim3(a.NF * a.heads), dim3(256), 0, st, a.part, a.ctxT, a.nchunk, a.heads);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if constexpr (MODE == MODE_BF16 && IO16 && NKT == 1 && TNO == 2) {
        if (sla_out_w_eligible(a)) return launch_sla_out_w(a, st);      // C = 64, many frames: one wave per 64 pixels
    }
    const SlaWork sw = sla_work(a, M::ES, false, true, true);
    LaunchScope ls(st, "sla_out8_kernel", sw.flops, sw.bytes, "<%d, %d, %d, %d, %d> C%d N%d NF%d", MODE, NKT, TMO, TNO, (int)IO16, a.C, a.N, a.NF);
    hipLaunchKernelGGL(ko, dim3(a.NF * a.nchunk), dim3(512), lds_out, st, a);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_sla_m(SlaArgs a, hipStream_t st) {
    using M = Mma<MODE>;
    constexpr int NCHN = 64 / M::KC;
    constexpr int RSE = NCHN * 64 + 16;
    sla_plan(a.N, a.nsub, a.nchunk);
    // the workspace is sized for that plan; with many frames longer chunks still fill the chip and halve the partial-context
    // traffic (8 x 4.4 KB per chunk, written here and re-read by the combine) per doubling
    {
        const int tiles = (a.N + 63) / 64;
#ifndef VDX_SLA_MIN_WGS
#define VDX_SLA_MIN_WGS 1024
#endif
        const long min_wgs = VDX_SLA_MIN_WGS;
        while (a.nsub * 2 <= tiles && (long)a.NF * ((tiles + 2 * a.nsub - 1) / (2 * a.nsub)) >= min_wgs) a.nsub *= 2;
        a.nchunk = (tiles + a.nsub - 1) / a.nsub;
    }
    const size_t part_bytes = (((size_t)a.NF * a.nchunk * a.heads * SLA_PART * 4) + 255) / 256 * 256;
    a.part = reinterpret_cast<float*>(a.workspace);
    a.ctxT = reinterpret_cast<char*>(a.workspace) + part_bytes;
    const bool generic_only = false;
    const int nkt = a.CPad / M::KT;
    if constexpr (MODE != MODE_F16)                           // (the one-wave-per-head kernels hard-code the bf16 / f32 register formats)
    if (a.heads == 8 && (a.C % 64 == 0 || a.C == 32) && !generic_only) {     // one wave per head; x tile double-buffered in LDS
        if constexpr (MODE == MODE_BF16) {
            if (a.io_bf16 && a.C == 64) return launch_sla8_t<MODE, 1, 1, 2, true>(a, st);
            if (a.io_bf16 && a.C == 128) return launch_sla8_t<MODE, 2, 1, 4, true>(a, st);
            if (a.io_bf16 && a.C == 32) return launch_sla8_t<MODE, 1, 1, 1, true>(a, st);       // dim 32 (configs/config_v2_2.yaml as written): level 0
        }
        if (nkt == 1 && a.C == 64) return launch_sla8_t<MODE, 1, 1, 2, false>(a, st);
        if (nkt == 2 && a.C == 64) return launch_sla8_t<MODE, 2, 1, 2, false>(a, st);
        if (nkt == 2 && a.C == 128) return launch_sla8_t<MODE, 2, 1, 4, false>(a, st);
    }
    const size_t lds1 = 1024 + (size_t)128 * ROW_STRIDE + (size_t)64 * RSE;
    hipError_t e;
    {
        const SlaWork sw = sla_work(a, M::ES, true, false, false);
        LaunchScope ls(st, "sla_ctx_kernel", sw.flops, sw.bytes, "<%d> C%d N%d NF%d nchunk%d io16 %d", MODE, a.C, a.N, a.NF, a.nchunk, a.io_bf16);
        hipLaunchKernelGGL(sla_ctx_kernel<MODE>, dim3(((a.NF * a.nchunk + 7) / 8) * 8 * a.heads), dim3(256), lds1, st, a);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    {
        LaunchScope ls(st, "sla_combine_kernel", 0.0, (double)a.NF * a.nchunk * a.heads * SLA_PART * 4, "<%d> NF%d nchunk%d", MODE, a.NF, a.nchunk);
        hipLaunchKernelGGL(sla_combine_kernel<MODE>, dim3(a.NF * a.heads), dim3(256), 0, st, a.part, a.ctxT, a.nchunk, a.heads);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (a.C <= 64) return launch_sla_out_t<MODE, 1>(a, st);
    if (a.C <= 128) return launch_sla_out_t<MODE, 2>(a, st);
    if (a.C <= 256) return launch_sla_out_t<MODE, 4>(a, st);
    if (a.C <= 512) return launch_sla_out_t<MODE, 8>(a, st);
    if (a.C <= 1024) return launch_sla_out_t<MODE, 16>(a, st);     // dim 128: the 1024-channel bottleneck level
    return hipErrorInvalidValue;
}

hipError_t launch_sla(int mode, SlaArgs a, hipStream_t st) {
    a.CPad = conv_cin_pad(mode, a.C);
    return mode == MODE_F32 ? launch_sla_m<MODE_F32>(a, st) : mode == MODE_F16 ? launch_sla_m<MODE_F16>(a, st) : launch_sla_m<MODE_BF16>(a, st);
}

}  // namespace vdx
