// Backward cores of the two attention flavours (gfx950).  The projection GEMMs around them (q/k/v recompute, dO = dy Wo^T,
// dx = dq Wq^T + ..., and every weight gradient) run on conv_igemm / conv_wgrad; only the small per-sequence /
// per-frame algebra lives here (VALU + LDS; 1.6 % + 2.3 % of the network FLOPs).
//
// attn_core_bwd: autodiff of softmax(q k^T) v per (sequence, head)        reference forward modules.py:294-323
// sla_bwd_a/b  : autodiff of SpatialLinearAttention's core per (frame, head) reference forward modules.py:105-118
#include "vdx_common.h"
#include "vdx_internal.h"

namespace vdx {

// one workgroup = one sequence of L <= 64 tokens, loop over heads.  qkv [npix][3*HD] (+bias, q unscaled), dO [npix][HD]
// -> O, dq, dk, dv [npix][HD] each.  Token t of sequence s is pixel row (s / inner) * outer_p + (s % inner) + t * tok_p.
__global__ __launch_bounds__(256) void attn_core_bwd_kernel(AttnBwdArgs P) {
    extern __shared__ float sm[];
    const int L = P.L, LD = 33, LP1 = L + 1;
    float* q = sm; float* k = q + L * LD; float* v = k + L * LD; float* dO = v + L * LD;
    float* Pm = dO + L * LD;                 // [L][L+1]
    float* dS = Pm + L * LP1;                // [L][L+1]
    const int tid = threadIdx.x;
    const long s = blockIdx.x;
    const long row0 = (s / P.inner) * P.outer_p + (s % P.inner);
    const int HD = P.heads * 32;
    for (int h = 0; h < P.heads; ++h) {
        __syncthreads();
        for (int i = tid; i < L * 32; i += 256) {
            const int t = i >> 5, d = i & 31;
            const size_t row = (size_t)(row0 + (long)t * P.tok_p);
            const float* src = P.qkv + row * 3 * HD + h * 32 + d;
            q[t * LD + d] = src[0] * P.scale; k[t * LD + d] = src[HD]; v[t * LD + d] = src[2 * HD];
            dO[t * LD + d] = P.dO[row * HD + h * 32 + d];
        }
        __syncthreads();
        for (int i = tid; i < L * L; i += 256) {
            const int a = i / L, b = i - a * L;
            float acc = 0.f, acc2 = 0.f;
#pragma unroll 8
            for (int d = 0; d < 32; ++d) { acc = fmaf(q[a * LD + d], k[b * LD + d], acc); acc2 = fmaf(dO[a * LD + d], v[b * LD + d], acc2); }
            Pm[a * LP1 + b] = acc; dS[a * LP1 + b] = acc2;           // scores, dP
        }
        __syncthreads();
        if (tid < L) {
            float m = -1e30f;
            for (int j = 0; j < L; ++j) m = fmaxf(m, Pm[tid * LP1 + j]);
            float sum = 0.f;
            for (int j = 0; j < L; ++j) { const float e = __expf(Pm[tid * LP1 + j] - m); Pm[tid * LP1 + j] = e; sum += e; }
            const float inv = 1.0f / sum;
            float dr = 0.f;
            for (int j = 0; j < L; ++j) { const float p = Pm[tid * LP1 + j] * inv; Pm[tid * LP1 + j] = p; dr = fmaf(dS[tid * LP1 + j], p, dr); }
            for (int j = 0; j < L; ++j) dS[tid * LP1 + j] = Pm[tid * LP1 + j] * (dS[tid * LP1 + j] - dr);
        }
        __syncthreads();
        for (int i = tid; i < L * 32; i += 256) {
            const int t = i >> 5, d = i & 31;
            float o = 0.f, dq = 0.f, dk = 0.f, dv = 0.f;
            for (int j = 0; j < L; ++j) {
                o = fmaf(Pm[t * LP1 + j], v[j * LD + d], o);
                dq = fmaf(dS[t * LP1 + j], k[j * LD + d], dq);
                dk = fmaf(dS[j * LP1 + t], q[j * LD + d], dk);
                dv = fmaf(Pm[j * LP1 + t], dO[j * LD + d], dv);
            }
            const size_t row = (size_t)(row0 + (long)t * P.tok_p);
            const size_t o_ = row * HD + h * 32 + d;
            P.O[o_] = o; P.dq[o_] = dq * P.scale; P.dk[o_] = dk; P.dv[o_] = dv;
        }
    }
}

constexpr int SLA_A = 2 * 1024 + 96;     // floats per (frame, head): ctx | dctx | kmax | ksum | T

// pass A: one workgroup per (frame, head): softmax-over-pixels statistics of k, ctx = ksm^T v, dctx = qsm^T dOut, T = sum_e dctx*ctx
__global__ __launch_bounds__(256) void sla_bwd_a_kernel(SlaBwdArgs P) {
    __shared__ float red[8][32];
    __shared__ float kmax[32], ksum[32];
    __shared__ float ks[32][33], vs[32][33], qs[32][33], ds[32][33];
    const int tid = threadIdx.x, h = blockIdx.y, n = blockIdx.x;
    const int d = tid & 31, g = tid >> 5;
    const size_t base = (size_t)n * P.N * 256 + h * 32;
    float m = -1e30f;
    for (int p = g; p < P.N; p += 8) m = fmaxf(m, P.k[base + (size_t)p * 256 + d]);
    red[g][d] = m;
    __syncthreads();
    if (tid < 32) { float t = red[0][tid]; for (int i = 1; i < 8; ++i) t = fmaxf(t, red[i][tid]); kmax[tid] = t; }
    __syncthreads();
    float sacc = 0.f;
    for (int p = g; p < P.N; p += 8) sacc += __expf(P.k[base + (size_t)p * 256 + d] - kmax[d]);
    red[g][d] = sacc;
    __syncthreads();
    if (tid < 32) { float t = 0.f; for (int i = 0; i < 8; ++i) t += red[i][tid]; ksum[tid] = t; }
    __syncthreads();
    const int pd = tid >> 3, e0 = (tid & 7) * 4;           // this thread owns ctx[pd][e0..e0+3]
    float c[4] = {0.f, 0.f, 0.f, 0.f}, dc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p0 = 0; p0 < P.N; p0 += 32) {
        __syncthreads();
        for (int i = tid; i < 32 * 32; i += 256) {
            const int pp = i >> 5, dd = i & 31;
            const bool ok = p0 + pp < P.N;
            const size_t o = base + (size_t)(p0 + pp) * 256 + dd;
            ks[pp][dd] = ok ? __expf(P.k[o] - kmax[dd]) / ksum[dd] : 0.f;
            vs[pp][dd] = ok ? P.v[o] : 0.f;
            qs[pp][dd] = ok ? P.q[o] : -1e30f;
            ds[pp][dd] = ok ? P.dOut[o] : 0.f;
        }
        __syncthreads();
        if (tid < 32) {                                      // softmax over D of each staged q row
            float mx = -1e30f;
            for (int dd = 0; dd < 32; ++dd) mx = fmaxf(mx, qs[tid][dd]);
            float su = 0.f;
            for (int dd = 0; dd < 32; ++dd) { const float e = (p0 + tid < P.N) ? __expf(qs[tid][dd] - mx) : 0.f; qs[tid][dd] = e; su += e; }
            const float inv = su > 0.f ? 1.0f / su : 0.f;
            for (int dd = 0; dd < 32; ++dd) qs[tid][dd] *= inv;
        }
        __syncthreads();
        for (int pp = 0; pp < 32; ++pp) {
            const float a = ks[pp][pd], b = qs[pp][pd];
#pragma unroll
            for (int e = 0; e < 4; ++e) { c[e] = fmaf(a, vs[pp][e0 + e], c[e]); dc[e] = fmaf(b, ds[pp][e0 + e], dc[e]); }
        }
    }
    float* out = P.A + ((size_t)n * P.heads + h) * SLA_A;
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { out[pd * 32 + e0 + e] = c[e]; out[1024 + pd * 32 + e0 + e] = dc[e]; t += c[e] * dc[e]; }
    t += __shfl_xor(t, 1); t += __shfl_xor(t, 2); t += __shfl_xor(t, 4);      // the 8 lanes sharing pd are consecutive
    if ((tid & 7) == 0) out[2048 + 64 + pd] = t;
    if (tid < 32) { out[2048 + tid] = kmax[tid]; out[2048 + 32 + tid] = ksum[tid]; }
}

// pass B: one thread = one pixel (all 32 channels of a head), loop over heads: out (forward, for dWout), dq, dk, dv
__global__ __launch_bounds__(256) void sla_bwd_b_kernel(SlaBwdArgs P) {
    __shared__ float ctx[32][33], dctx[32][33];
    __shared__ float km[32], ksu[32], T[32];
    const int tid = threadIdx.x;
    const int tiles = (P.N + 255) / 256;
    const int n = blockIdx.x / tiles, p = (blockIdx.x % tiles) * 256 + tid;
    const bool ok = p < P.N;
    for (int h = 0; h < P.heads; ++h) {
        const float* A = P.A + ((size_t)n * P.heads + h) * SLA_A;
        __syncthreads();
        for (int i = tid; i < 1024; i += 256) { ctx[i >> 5][i & 31] = A[i]; dctx[i >> 5][i & 31] = A[1024 + i]; }
        if (tid < 32) { km[tid] = A[2048 + tid]; ksu[tid] = A[2048 + 32 + tid]; T[tid] = A[2048 + 64 + tid]; }
        __syncthreads();
        if (!ok) continue;
        const size_t o = ((size_t)n * P.N + p) * 256 + h * 32;
        float q[32], x[32];
        float mx = -1e30f;
#pragma unroll
        for (int d = 0; d < 32; d += 4) { const float4 t = *reinterpret_cast<const float4*>(P.q + o + d); q[d] = t.x; q[d + 1] = t.y; q[d + 2] = t.z; q[d + 3] = t.w; }
#pragma unroll
        for (int d = 0; d < 32; ++d) mx = fmaxf(mx, q[d]);
        float su = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) { q[d] = __expf(q[d] - mx); su += q[d]; }
        const float inv = 1.0f / su;
#pragma unroll
        for (int d = 0; d < 32; ++d) q[d] *= inv;                               // qsm
#pragma unroll
        for (int e = 0; e < 32; e += 4) { const float4 t = *reinterpret_cast<const float4*>(P.dOut + o + e); x[e] = t.x; x[e + 1] = t.y; x[e + 2] = t.z; x[e + 3] = t.w; }
        // out[e] = sum_d ctx[d][e] qsm[d] ;  dqsm[d] = sum_e ctx[d][e] dOut[e]
        float outv[32], dqs[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) outv[e] = 0.f;
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) {
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < 32; ++e) { acc = fmaf(ctx[d][e], x[e], acc); outv[e] = fmaf(ctx[d][e], q[d], outv[e]); }
            dqs[d] = acc; dot = fmaf(q[d], acc, dot);
        }
#pragma unroll
        for (int d = 0; d < 32; d += 4) {
            *reinterpret_cast<float4*>(P.dq + o + d) = make_float4(q[d] * (dqs[d] - dot), q[d + 1] * (dqs[d + 1] - dot), q[d + 2] * (dqs[d + 2] - dot), q[d + 3] * (dqs[d + 3] - dot));
            *reinterpret_cast<float4*>(P.O + o + d) = make_float4(outv[d], outv[d + 1], outv[d + 2], outv[d + 3]);
        }
        // k side: ksm[d], dksm[d] = sum_e dctx[d][e] v[e], dk = ksm (dksm - T) ; dv[e] = sum_d ksm[d] dctx[d][e]
#pragma unroll
        for (int d = 0; d < 32; d += 4) { const float4 t = *reinterpret_cast<const float4*>(P.k + o + d); q[d] = t.x; q[d + 1] = t.y; q[d + 2] = t.z; q[d + 3] = t.w; }
#pragma unroll
        for (int d = 0; d < 32; ++d) q[d] = __expf(q[d] - km[d]) / ksu[d];      // ksm
#pragma unroll
        for (int e = 0; e < 32; e += 4) { const float4 t = *reinterpret_cast<const float4*>(P.v + o + e); x[e] = t.x; x[e + 1] = t.y; x[e + 2] = t.z; x[e + 3] = t.w; }
#pragma unroll
        for (int e = 0; e < 32; ++e) outv[e] = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) {
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < 32; ++e) { acc = fmaf(dctx[d][e], x[e], acc); outv[e] = fmaf(dctx[d][e], q[d], outv[e]); }
            dqs[d] = q[d] * (acc - T[d]);
        }
#pragma unroll
        for (int d = 0; d < 32; d += 4) {
            *reinterpret_cast<float4*>(P.dk + o + d) = make_float4(dqs[d], dqs[d + 1], dqs[d + 2], dqs[d + 3]);
            *reinterpret_cast<float4*>(P.dv + o + d) = make_float4(outv[d], outv[d + 1], outv[d + 2], outv[d + 3]);
        }
    }
}

hipError_t launch_attn_core_bwd(const AttnBwdArgs& a, hipStream_t st) {
    const size_t lds = ((size_t)4 * a.L * 33 + 2 * a.L * (a.L + 1)) * 4;
    auto kfn = attn_core_bwd_kernel;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kfn, dim3((unsigned)a.nseq), dim3(256), lds, st, a);
    return hipGetLastError();
}

size_t sla_bwd_scratch_floats(int NF, int heads) { return (size_t)NF * heads * SLA_A; }

hipError_t launch_sla_bwd(const SlaBwdArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(sla_bwd_a_kernel, dim3(a.NF, a.heads), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int tiles = (a.N + 255) / 256;
    hipLaunchKernelGGL(sla_bwd_b_kernel, dim3(a.NF * tiles), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace vdx
