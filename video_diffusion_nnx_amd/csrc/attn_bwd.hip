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

// ---- bf16-mode form for sequences of <= 16 tokens: one wave per sequence, loop over heads, everything on MFMA ----------
// Q (scaled), K, V, dO of one (sequence, head) are rounded to bf16 into four [16 tokens][32] LDS images (wave-private).
//   S^T = K Q^T, S = Q K^T, dP^T = V dO^T, dP = dO V^T     four v_mfma_f32_16x16x32_bf16 on row reads of the images
//   softmax / dS in both orientations (the accumulator of X^T is the B operand "K = row index of X^T" of a K=16 MFMA)
//   O^T = V^T P^T, dV^T = dO^T P, dQ^T = K^T dS^T, dK^T = Q^T dS     A operands = ds_read_b64_tr_b16 of the images
// so no score / probability ever goes through LDS, and each lane ends with 4 consecutive channels of one token: float4 stores.
typedef short s16x4b __attribute__((ext_vector_type(4)));
constexpr int AB_RS = 32 * 2 + 16;          // bytes per token row of an image

__device__ __forceinline__ s16x4b tr_read4(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4b*)(uintptr_t)(unsigned)(uintptr_t)p);
}
__device__ __forceinline__ s16x4b pack4_bf16(const f32x4& v) {
    const uint2 u = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    return __builtin_bit_cast(s16x4b, u);
}

__global__ __launch_bounds__(256) void attn_core_bwd16_kernel(AttnBwdArgs P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [4 waves][4 images][16][AB_RS]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lp = lane & 15, q = lane >> 4;
    char* img = smem + w * (4 * 16 * AB_RS);
    char* Qi = img; char* Ki = Qi + 16 * AB_RS; char* Vi = Ki + 16 * AB_RS; char* Di = Vi + 16 * AB_RS;
    const int HD = P.heads * 32;
    const long s = (long)blockIdx.x * 4 + w;
    const bool live = s < P.nseq;                            // a dead wave still runs (EXEC must stay full for the tr reads): it
    const long sc = live ? s : 0;                            // recomputes sequence 0 and stores nothing
    const long row0 = (sc / P.inner) * P.outer_p + (sc % P.inner);
    // staging roles: lane -> (token row lr, 16-byte quarter pieces lq and lq + 4 of the 32 channels)
    const int lr = lane >> 2, lq = lane & 3;
    const bool rvalid = lr < P.L;
    const size_t grow = (size_t)(row0 + (long)lr * P.tok_p);
    // transposing-read roles: group q supplies token rows 4q..4q+3; in-group lane 4*qr + pc -> row 4q + qr, 8-byte chunk pc
    const int troff = (4 * q + (lp >> 2)) * AB_RS + (lp & 3) * 8;
    const size_t orow = (size_t)(row0 + (long)lp * P.tok_p);          // output: lane (token lp, q) writes channels 4q..4q+3 (+16)
    const bool ovalid = live && lp < P.L;
    const float L2E = 1.44269504088896f;
    for (int h = 0; h < P.heads; ++h) {
        // ---- stage the four images (wave-private: LDS ops of one wave stay in order) ----
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = (lq + 4 * u) * 4;
            float4 vq = make_float4(0.f, 0.f, 0.f, 0.f), vk = vq, vv = vq, vd = vq;
            if (rvalid) {
                const float* src = P.qkv + grow * 3 * HD + h * 32 + c;
                vq = *reinterpret_cast<const float4*>(src);
                vk = *reinterpret_cast<const float4*>(src + HD);
                vv = *reinterpret_cast<const float4*>(src + 2 * HD);
                vd = *reinterpret_cast<const float4*>(P.dO + grow * HD + h * 32 + c);
            }
            *reinterpret_cast<uint2*>(Qi + lr * AB_RS + c * 2) = make_uint2(pack_bf16x2(vq.x * P.scale, vq.y * P.scale), pack_bf16x2(vq.z * P.scale, vq.w * P.scale));
            *reinterpret_cast<uint2*>(Ki + lr * AB_RS + c * 2) = make_uint2(pack_bf16x2(vk.x, vk.y), pack_bf16x2(vk.z, vk.w));
            *reinterpret_cast<uint2*>(Vi + lr * AB_RS + c * 2) = make_uint2(pack_bf16x2(vv.x, vv.y), pack_bf16x2(vv.z, vv.w));
            *reinterpret_cast<uint2*>(Di + lr * AB_RS + c * 2) = make_uint2(pack_bf16x2(vd.x, vd.y), pack_bf16x2(vd.z, vd.w));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- scores and dP in both orientations (row reads: lane (r, q) = token r, channels 8q..8q+7) ----
        const bf16x8 qf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Qi + lp * AB_RS + q * 16));
        const bf16x8 kf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Ki + lp * AB_RS + q * 16));
        const bf16x8 vf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Vi + lp * AB_RS + q * 16));
        const bf16x8 df = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Di + lp * AB_RS + q * 16));
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 ST = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, z, 0, 0, 0);    // lane (a = lp, q): keys b = 4q+e
        f32x4 S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf, z, 0, 0, 0);     // lane (b = lp, q): queries a = 4q+e
        f32x4 dPT = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, df, z, 0, 0, 0);   // dP^T[b][a]
        f32x4 dP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, vf, z, 0, 0, 0);    // dP[a][b]
        // ---- orientation "T": query a = lp, its keys in (q, e) ----
        f32x4 PT, dST;
        {
            float mx = -1e30f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { if (4 * q + e >= P.L) ST[e] = -1e30f; mx = fmaxf(mx, ST[e]); }
            mx = max_q(mx);
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { PT[e] = __builtin_amdgcn_exp2f((ST[e] - mx) * L2E); sum += PT[e]; }
            const float inv = __builtin_amdgcn_rcpf(reduce_q(sum));
            float dr = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { PT[e] *= inv; dr = fmaf(dPT[e], PT[e], dr); }
            dr = reduce_q(dr);
#pragma unroll
            for (int e = 0; e < 4; ++e) dST[e] = PT[e] * (dPT[e] - dr);
        }
        // ---- orientation "N": key b = lp, queries a = 4q+e: the row reductions run over the 16 lanes of a DPP row ----
        f32x4 Pn, dSn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sv = (lp >= P.L) ? -1e30f : S[e];
            const float mx = max16(sv);
            const float pe = __builtin_amdgcn_exp2f((sv - mx) * L2E);
            const float pn = pe * __builtin_amdgcn_rcpf(reduce16(pe));
            const float dr = reduce16(dP[e] * pn);
            Pn[e] = pn; dSn[e] = pn * (dP[e] - dr);
        }
        const s16x4b bPT = pack4_bf16(PT), bdST = pack4_bf16(dST), bPn = pack4_bf16(Pn), bdSn = pack4_bf16(dSn);
        // ---- O^T = V^T P^T, dV^T = dO^T P, dQ^T = K^T dS^T, dK^T = Q^T dS  (A = transposing read of an image, 16 channels per tile) ----
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const s16x4b av = tr_read4(Vi + troff + t * 32), ad = tr_read4(Di + troff + t * 32);
            const s16x4b ak = tr_read4(Ki + troff + t * 32), aq = tr_read4(Qi + troff + t * 32);
            const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bPT, z, 0, 0, 0);
            const f32x4 dv = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ad, bPn, z, 0, 0, 0);
            const f32x4 dq = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ak, bdST, z, 0, 0, 0);
            const f32x4 dk = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(aq, bdSn, z, 0, 0, 0);
            if (ovalid) {
                const size_t o_ = orow * HD + h * 32 + t * 16 + 4 * q;
                *reinterpret_cast<float4*>(P.O + o_) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(P.dv + o_) = make_float4(dv[0], dv[1], dv[2], dv[3]);
                *reinterpret_cast<float4*>(P.dq + o_) = make_float4(dq[0] * P.scale, dq[1] * P.scale, dq[2] * P.scale, dq[3] * P.scale);
                *reinterpret_cast<float4*>(P.dk + o_) = make_float4(dk[0], dk[1], dk[2], dk[3]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // images are rewritten for the next head
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
    if (a.bf16_mma && a.L <= 16) {
        const long blocks = (a.nseq + 3) / 4;
        hipLaunchKernelGGL(attn_core_bwd16_kernel, dim3((unsigned)blocks), dim3(256), 4 * 4 * 16 * AB_RS, st, a);
        return hipGetLastError();
    }
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
