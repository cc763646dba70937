// Backward cores of the two attention flavours (gfx950).  The projection GEMMs around them (q/k/v recompute, dO = dy Wo^T,
// dx = dq Wq^T + ..., and every weight gradient) run on conv_igemm / conv_wgrad; only the small per-sequence /
// per-frame algebra lives here (VALU + LDS; 1.6 % + 2.3 % of the network FLOPs).
//
// attn_core_bwd: autodiff of softmax(q k^T) v per (sequence, head)        reference forward modules.py:294-323
// sla_bwd_a/b  : autodiff of SpatialLinearAttention's core per (frame, head) reference forward modules.py:105-118
#include "vdx_common.h"
#include "vdx_internal.h"

namespace vdx {

// one workgroup = one (sequence, head) of L <= 64 tokens (blockIdx.y = head).  qkv [npix][3*HD] (+bias, q unscaled), dO [npix][HD]
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
    {
        const int h = blockIdx.y;
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
            const size_t o_ = row * HD + h * 32 + d, g_ = row * P.dstride + h * 32 + d;
            P.O[o_] = o; P.dq[g_] = dq * P.scale; P.dk[g_] = dk; P.dv[g_] = dv;
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

template <bool IO16>
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
    const float SL2E = P.scale * 1.44269504088896f;           // softmax(scale * s) = exp2((s - max s) * scale * log2 e) / sum
    for (int h = 0; h < P.heads; ++h) {
        // ---- stage the four images (wave-private: LDS ops of one wave stay in order).  q stays UNSCALED here: 1/sqrt(d) is folded
        //      into the exponent of both softmaxes and into the dq / dk outputs ----
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = (lq + 4 * u) * 4;
            uint2 pq = make_uint2(0u, 0u), pk = pq, pv = pq, pd = pq;
            if (rvalid) {
                if (IO16) {
                    const char* src = reinterpret_cast<const char*>(P.qkv) + (grow * 3 * HD + h * 32 + c) * 2;
                    pq = *reinterpret_cast<const uint2*>(src);
                    pk = *reinterpret_cast<const uint2*>(src + HD * 2);
                    pv = *reinterpret_cast<const uint2*>(src + 2 * HD * 2);
                    pd = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(P.dO) + (grow * HD + h * 32 + c) * 2);
                } else {
                    const float* src = P.qkv + grow * 3 * HD + h * 32 + c;
                    const float4 vq = *reinterpret_cast<const float4*>(src), vk = *reinterpret_cast<const float4*>(src + HD);
                    const float4 vv = *reinterpret_cast<const float4*>(src + 2 * HD), vd = *reinterpret_cast<const float4*>(P.dO + grow * HD + h * 32 + c);
                    pq = make_uint2(pack_bf16x2(vq.x, vq.y), pack_bf16x2(vq.z, vq.w)); pk = make_uint2(pack_bf16x2(vk.x, vk.y), pack_bf16x2(vk.z, vk.w));
                    pv = make_uint2(pack_bf16x2(vv.x, vv.y), pack_bf16x2(vv.z, vv.w)); pd = make_uint2(pack_bf16x2(vd.x, vd.y), pack_bf16x2(vd.z, vd.w));
                }
            }
            *reinterpret_cast<uint2*>(Qi + lr * AB_RS + c * 2) = pq;
            *reinterpret_cast<uint2*>(Ki + lr * AB_RS + c * 2) = pk;
            *reinterpret_cast<uint2*>(Vi + lr * AB_RS + c * 2) = pv;
            *reinterpret_cast<uint2*>(Di + lr * AB_RS + c * 2) = pd;
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
            for (int e = 0; e < 4; ++e) { PT[e] = __builtin_amdgcn_exp2f((ST[e] - mx) * SL2E); sum += PT[e]; }
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
            const float pe = __builtin_amdgcn_exp2f((sv - mx) * SL2E);
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
                const size_t o_ = orow * HD + h * 32 + t * 16 + 4 * q, g_ = orow * P.dstride + h * 32 + t * 16 + 4 * q;
                store4_f32_or_bf16(P.O, o_, make_float4(o[0], o[1], o[2], o[3]), IO16);
                store4_f32_or_bf16(P.dv, g_, make_float4(dv[0], dv[1], dv[2], dv[3]), IO16);
                store4_f32_or_bf16(P.dq, g_, make_float4(dq[0] * P.scale, dq[1] * P.scale, dq[2] * P.scale, dq[3] * P.scale), IO16);
                store4_f32_or_bf16(P.dk, g_, make_float4(dk[0] * P.scale, dk[1] * P.scale, dk[2] * P.scale, dk[3] * P.scale), IO16);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // images are rewritten for the next head
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---- fused form for the temporal attention of the widest level (bf16 mode, C == 64, 8 heads, <= 16 tokens) ------------------
// At level 0 the q|k|v tensor is 12x the block input, so the unfused chain (q/k/v recompute -> dO projection -> core -> dx projection)
// moves ~2.7 GB per block at batch 4; here one wave owns a sequence, keeps its x and g rows as MFMA B fragments, and per head
//   q^T, k^T, v^T = W_h x^T + b (A = rows of the packed forward weights, resident in LDS for the whole workgroup), dO^T = Wo_h^T g^T
//   -> the four wave-private [token][32] images of attn_core_bwd16_kernel -> the same core
//   dx^T += W_q,h^T dq^T + W_k,h^T dk^T + W_v,h^T dv^T    (A = transposing reads of the SAME weight image, B = the packed outputs)
// and writes O and dq|dk|dv (bf16) only for the weight-gradient kernels: x, g in (134 MB), O, dq|dk|dv, dx out.
constexpr int AX_RSW = 64 * 2 + 16;          // bytes per row of the weight image (conflict-free for the row and the transposing reads)

__device__ __forceinline__ uint4 pack8u_bf16(const float4& a, const float4& b) {
    return make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
}

__global__ __launch_bounds__(512, 2) void attn_bwd16x_kernel(AttnBwdXArgs P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Wi = smem;                                                  // [768 = (part, head, d)][AX_RSW]: K = input channel
    float* bias = reinterpret_cast<float*>(Wi + 768 * AX_RSW);         // [768]
    char* imgs = reinterpret_cast<char*>(bias + 768);                 // [8 waves][4 images][16][AB_RS]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lp = lane & 15, q = lane >> 4;
    for (int i = tid; i < 768 * 8; i += 512) {
        const int row = i >> 3, c = i & 7;
        *reinterpret_cast<uint4*>(Wi + row * AX_RSW + c * 16) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(P.wqkv) + (size_t)row * 128 + c * 16);
    }
    for (int i = tid; i < 768; i += 512) bias[i] = P.bqkv[i];
    __syncthreads();
    char* img = imgs + w * (4 * 16 * AB_RS);
    char* Qi = img; char* Ki = Qi + 16 * AB_RS; char* Vi = Ki + 16 * AB_RS; char* Di = Vi + 16 * AB_RS;
    const int troff = (4 * q + (lp >> 2)) * AB_RS + (lp & 3) * 8;      // transposing reads of the images (attn_core_bwd16_kernel)
    const int trW = (4 * q + (lp >> 2)) * AX_RSW + (lp & 3) * 8;       // ... of a 16-row group of the weight image
    const bool tvalid = lp < P.L;
    const float SL2E = P.scale * 1.44269504088896f;
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* woT = reinterpret_cast<const char*>(P.woT);
    for (long s = (long)blockIdx.x * 8 + w; s < P.nseq; s += (long)gridDim.x * 8) {     // (wave-uniform trip count: EXEC stays full)
        const long row0 = (s / P.inner) * P.outer_p + (s % P.inner);
        const size_t prow = (size_t)(row0 + (long)lp * P.tok_p);       // this lane's token row
        // x and g rows of token lp as B fragments: K step ks covers channels 32 ks + 8 q .. + 7
        uint4 xf[2], gf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a, d = a;
            uint4 x16 = make_uint4(0u, 0u, 0u, 0u);
            if (tvalid) {
                const size_t xe = prow * 64 + ks * 32 + 8 * q;
                const float* gp = P.g + xe;
                if (P.x_bf16) x16 = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(P.x) + xe * 2);      // bf16 activation storage: the fragment as it is
                else { a = *reinterpret_cast<const float4*>(P.x + xe); b = *reinterpret_cast<const float4*>(P.x + xe + 4); }
                c = *reinterpret_cast<const float4*>(gp); d = *reinterpret_cast<const float4*>(gp + 4);
            }
            xf[ks] = P.x_bf16 ? x16 : pack8u_bf16(a, b); gf[ks] = pack8u_bf16(c, d);
        }
        f32x4 dxT[4] = {z, z, z, z};                                   // dx^T: rows = channels 16 ct + 4q + e, column = token lp
#pragma unroll 1
        for (int h = 0; h < 8; ++h) {
            // ---- projections of this head: accumulator rows = d (16 t + 4q + e), column = token lp ----
            f32x4 qT[2], kT[2], vT[2], dT[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + h * 32 + t * 16 + 4 * q);
                const float4 bk = *reinterpret_cast<const float4*>(bias + 256 + h * 32 + t * 16 + 4 * q);
                const float4 bv = *reinterpret_cast<const float4*>(bias + 512 + h * 32 + t * 16 + 4 * q);
                qT[t] = f32x4{bq.x, bq.y, bq.z, bq.w}; kT[t] = f32x4{bk.x, bk.y, bk.z, bk.w}; vT[t] = f32x4{bv.x, bv.y, bv.z, bv.w};
                dT[t] = z;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const char* wr = Wi + (h * 32 + t * 16 + lp) * AX_RSW + ks * 64 + q * 16;
                    const bf16x8 aq = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wr));
                    const bf16x8 ak = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wr + 256 * AX_RSW));
                    const bf16x8 av = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wr + 512 * AX_RSW));
                    const bf16x8 ao = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(woT + ((size_t)(h * 32 + t * 16 + lp) * 64 + ks * 32 + q * 8) * 2));
                    const bf16x8 xb = __builtin_bit_cast(bf16x8, xf[ks]), gb = __builtin_bit_cast(bf16x8, gf[ks]);
                    qT[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, xb, qT[t], 0, 0, 0);
                    kT[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak, xb, kT[t], 0, 0, 0);
                    vT[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, xb, vT[t], 0, 0, 0);
                    dT[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ao, gb, dT[t], 0, 0, 0);
                }
            // ---- the four images: row = token lp, 4 channels 16 t + 4q .. (tokens past L: zero rows, as the unfused staging) ----
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int co = (t * 16 + 4 * q) * 2;
                const uint2 zz = make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(Qi + lp * AB_RS + co) = tvalid ? __builtin_bit_cast(uint2, pack4_bf16(qT[t])) : zz;
                *reinterpret_cast<uint2*>(Ki + lp * AB_RS + co) = tvalid ? __builtin_bit_cast(uint2, pack4_bf16(kT[t])) : zz;
                *reinterpret_cast<uint2*>(Vi + lp * AB_RS + co) = tvalid ? __builtin_bit_cast(uint2, pack4_bf16(vT[t])) : zz;
                *reinterpret_cast<uint2*>(Di + lp * AB_RS + co) = __builtin_bit_cast(uint2, pack4_bf16(dT[t]));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- core (attn_core_bwd16_kernel) ----
            const bf16x8 qf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Qi + lp * AB_RS + q * 16));
            const bf16x8 kf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Ki + lp * AB_RS + q * 16));
            const bf16x8 vf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Vi + lp * AB_RS + q * 16));
            const bf16x8 df = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Di + lp * AB_RS + q * 16));
            f32x4 ST = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, z, 0, 0, 0);
            f32x4 S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf, z, 0, 0, 0);
            f32x4 dPT = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, df, z, 0, 0, 0);
            f32x4 dP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, vf, z, 0, 0, 0);
            f32x4 PT, dST;
            {
                float mx = -1e30f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { if (4 * q + e >= P.L) ST[e] = -1e30f; mx = fmaxf(mx, ST[e]); }
                mx = max_q(mx);
                float sum = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { PT[e] = __builtin_amdgcn_exp2f((ST[e] - mx) * SL2E); sum += PT[e]; }
                const float inv = __builtin_amdgcn_rcpf(reduce_q(sum));
                float dr = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { PT[e] *= inv; dr = fmaf(dPT[e], PT[e], dr); }
                dr = reduce_q(dr);
#pragma unroll
                for (int e = 0; e < 4; ++e) dST[e] = PT[e] * (dPT[e] - dr);
            }
            f32x4 Pn, dSn;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sv = (lp >= P.L) ? -1e30f : S[e];
                const float mx = max16(sv);
                const float pe = __builtin_amdgcn_exp2f((sv - mx) * SL2E);
                const float pn = pe * __builtin_amdgcn_rcpf(reduce16(pe));
                const float dr = reduce16(dP[e] * pn);
                Pn[e] = pn; dSn[e] = pn * (dP[e] - dr);
            }
            const s16x4b bPT = pack4_bf16(PT), bdST = pack4_bf16(dST), bPn = pack4_bf16(Pn), bdSn = pack4_bf16(dSn);
            uint2 pq[2], pk[2], pv[2];                                 // dq^T, dk^T, dv^T of this head, packed: rows d = 16 t + 4q + e, column = token lp
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const s16x4b av = tr_read4(Vi + troff + t * 32), ad = tr_read4(Di + troff + t * 32);
                const s16x4b ak = tr_read4(Ki + troff + t * 32), aq = tr_read4(Qi + troff + t * 32);
                const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bPT, z, 0, 0, 0);
                const f32x4 dv = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ad, bPn, z, 0, 0, 0);
                const f32x4 dq = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ak, bdST, z, 0, 0, 0);
                const f32x4 dk = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(aq, bdSn, z, 0, 0, 0);
                const uint2 po = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
                pv[t] = make_uint2(pack_bf16x2(dv[0], dv[1]), pack_bf16x2(dv[2], dv[3]));
                pq[t] = make_uint2(pack_bf16x2(dq[0] * P.scale, dq[1] * P.scale), pack_bf16x2(dq[2] * P.scale, dq[3] * P.scale));
                pk[t] = make_uint2(pack_bf16x2(dk[0] * P.scale, dk[1] * P.scale), pack_bf16x2(dk[2] * P.scale, dk[3] * P.scale));
                if (tvalid) {
                    const int cc = h * 32 + t * 16 + 4 * q;
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(P.O) + (prow * 256 + cc) * 2) = po;
                    char* dst = reinterpret_cast<char*>(P.dqkv) + (prow * 768 + cc) * 2;
                    *reinterpret_cast<uint2*>(dst) = pq[t];
                    *reinterpret_cast<uint2*>(dst + 512) = pk[t];
                    *reinterpret_cast<uint2*>(dst + 1024) = pv[t];
                }
            }
            // ---- dx^T += W_part,h^T d(part)^T: K slots (q, e) = d 4q + e (tile 0) | 16 + 4q + e (tile 1) on both operands ----
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                const uint2* pp = part == 0 ? pq : (part == 1 ? pk : pv);
                const bf16x8 bfrag = __builtin_bit_cast(bf16x8, make_uint4(pp[0].x, pp[0].y, pp[1].x, pp[1].y));
                const char* wb = Wi + (part * 256 + h * 32) * AX_RSW + trW;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const s16x4b lo = tr_read4(wb + ct * 32), hi = tr_read4(wb + 16 * AX_RSW + ct * 32);
                    typedef short s16x8b __attribute__((ext_vector_type(8)));
                    const s16x8b a8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    dxT[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), bfrag, dxT[ct], 0, 0, 0);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // images are rewritten for the next head
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // ---- dx = g + ...: lane (token lp, q) owns channels 16 ct + 4q .. + 3 ----
        if (tvalid) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const size_t o = prow * 64 + ct * 16 + 4 * q;
                const float4 gg = *reinterpret_cast<const float4*>(P.g + o);
                *reinterpret_cast<float4*>(P.dx + o) = make_float4(gg.x + dxT[ct][0], gg.y + dxT[ct][1], gg.z + dxT[ct][2], gg.w + dxT[ct][3]);
            }
        }
    }
}

hipError_t launch_attn_bwd_fused(const AttnBwdXArgs& a, hipStream_t st) {
    if (a.L < 1 || a.L > 16 || a.nseq < 1) return hipErrorInvalidValue;
    const size_t lds = (size_t)768 * AX_RSW + 768 * 4 + (size_t)8 * 4 * 16 * AB_RS;
    auto kfn = attn_bwd16x_kernel;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    const long want = (a.nseq + 7) / 8;
    hipLaunchKernelGGL(kfn, dim3((unsigned)std::min<long>(want, cus)), dim3(512), lds, st, a);
    return hipGetLastError();
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

// ---- bf16-mode form of pass A: the two 32x32 reductions over the pixels of a (frame, head) on MFMA ----------------------------
// Wave w of the 4-wave workgroup takes the 32-pixel tiles w, w+4, ... .  Phase 1: column maxima of k.  Phase 2 per tile: exp(k - max),
// v, softmax_D(q) and dOut are rounded to bf16 into four [32 pixels][32] LDS images (wave-private) and
//   ctx_un[d,e] += sum_n exp(k)[n,d] v[n,e],   dctx[d,e] += sum_n qsm[n,d] dOut[n,e]
// are eight v_mfma_f32_16x16x32_bf16 whose K-strided fragments are transposing LDS reads (ds_read_b64_tr_b16), as in
// conv_wgrad16_kernel; the column sums of exp(k) are a by-product.  The waves' partial tiles meet in LDS; ctx = ctx_un / ksum.
constexpr int SA_RS = 32 * 2 + 16;           // bytes per pixel row of an image

__device__ __forceinline__ bf16x8 sa_frag(const char* r0, const char* r1) {
    typedef short s16x8t __attribute__((ext_vector_type(8)));
    const s16x4b lo = tr_read4(r0), hi = tr_read4(r1);
    const s16x8t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// max / sum over the 8 lanes {l ^ 8, l ^ 16, l ^ 32 combinations} that share (lane & 7)
__device__ __forceinline__ float colred_max(float v) {
    v = fmaxf(v, dpp_row<0x128>(v));                          // row_ror:8 = lane ^ 8 inside a 16-lane row
    return max_q(v);
}
__device__ __forceinline__ float colred_sum(float v) {
    v += dpp_row<0x128>(v);
    return reduce_q(v);
}
// sum / max over the 8 lanes that share (lane >> 3): lane ^ 1, ^ 2 (quad_perm), ^ 4 (row_half_mirror on quad-uniform values)
__device__ __forceinline__ float rowred_max(float v) {
    v = fmaxf(v, dpp_row<0xB1>(v)); v = fmaxf(v, dpp_row<0x4E>(v)); return fmaxf(v, dpp_row<0x141>(v));
}
__device__ __forceinline__ float rowred_sum(float v) {
    v += dpp_row<0xB1>(v); v += dpp_row<0x4E>(v); return v + dpp_row<0x141>(v);
}

template <bool IO16>
__global__ __launch_bounds__(256) void sla_bwd_a16_kernel(SlaBwdArgs P) {
    __shared__ __attribute__((aligned(16))) char imgs[4][4][32 * SA_RS];      // [wave][ks, v, qs, dOut]
    __shared__ float part[4][2][1024];                                        // per-wave ctx_un / dctx tiles
    __shared__ float kmx[4][32], ksm[4][32];
    __shared__ float kmax[32], ksum[32];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int rg = lane >> 3, pc = lane & 7;                  // staging: pixel row rg + 8u, channels 4pc..4pc+3
    const int h = blockIdx.y, n = blockIdx.x;
    const size_t base = (size_t)n * P.N * 256 + h * 32 + pc * 4;
    const int ntiles = (P.N + 31) / 32;
    const float L2E = 1.44269504088896f;
    // ---- phase 1: column maxima of k ----
    float4 mx = make_float4(-1e30f, -1e30f, -1e30f, -1e30f);
    for (int t = w; t < ntiles; t += 4)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = t * 32 + rg + 8 * u;
            if (p < P.N) {
                const float4 kv = load4_f32_or_bf16(P.k, base + (size_t)p * 256, IO16);
                mx.x = fmaxf(mx.x, kv.x); mx.y = fmaxf(mx.y, kv.y); mx.z = fmaxf(mx.z, kv.z); mx.w = fmaxf(mx.w, kv.w);
            }
        }
    mx.x = colred_max(mx.x); mx.y = colred_max(mx.y); mx.z = colred_max(mx.z); mx.w = colred_max(mx.w);
    if (rg == 0) *reinterpret_cast<float4*>(&kmx[w][pc * 4]) = mx;
    __syncthreads();
    if (tid < 32) kmax[tid] = fmaxf(fmaxf(kmx[0][tid], kmx[1][tid]), fmaxf(kmx[2][tid], kmx[3][tid]));
    __syncthreads();
    const float4 km = *reinterpret_cast<const float4*>(&kmax[pc * 4]);
    // ---- phase 2 ----
    char* Ki = imgs[w][0]; char* Vi = imgs[w][1]; char* Qi = imgs[w][2]; char* Di = imgs[w][3];
    f32x4 cacc[2][2], dacc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { cacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float4 ks4 = make_float4(0.f, 0.f, 0.f, 0.f);            // column sums of exp(k - max), this lane's rows
    // transposing-read roles: group q supplies pixels 8q..8q+7 of the tile; in-group lane 4*qr + cc -> pixel 8q + qr (+4), 8-byte chunk cc
    const int troff0 = (8 * q + (r >> 2)) * SA_RS + (r & 3) * 8, troff1 = troff0 + 4 * SA_RS;
    for (int t = w; t < ntiles; t += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = rg + 8 * u, p = t * 32 + row;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv, dv = kv, qv = make_float4(-1e30f, -1e30f, -1e30f, -1e30f);
            const bool ok = p < P.N;
            if (ok) {
                const size_t o = base + (size_t)p * 256;
                kv = load4_f32_or_bf16(P.k, o, IO16); vv = load4_f32_or_bf16(P.v, o, IO16);
                qv = load4_f32_or_bf16(P.q, o, IO16); dv = load4_f32_or_bf16(P.dOut, o, IO16);
                kv.x = __builtin_amdgcn_exp2f((kv.x - km.x) * L2E); kv.y = __builtin_amdgcn_exp2f((kv.y - km.y) * L2E);
                kv.z = __builtin_amdgcn_exp2f((kv.z - km.z) * L2E); kv.w = __builtin_amdgcn_exp2f((kv.w - km.w) * L2E);
                ks4.x += kv.x; ks4.y += kv.y; ks4.z += kv.z; ks4.w += kv.w;
            }
            // softmax over the 32 channels of this pixel: 8 lanes x 4 values
            const float m = rowred_max(fmaxf(fmaxf(qv.x, qv.y), fmaxf(qv.z, qv.w)));
            float4 e;
            e.x = __builtin_amdgcn_exp2f((qv.x - m) * L2E); e.y = __builtin_amdgcn_exp2f((qv.y - m) * L2E);
            e.z = __builtin_amdgcn_exp2f((qv.z - m) * L2E); e.w = __builtin_amdgcn_exp2f((qv.w - m) * L2E);
            const float inv = ok ? __builtin_amdgcn_rcpf(rowred_sum(e.x + e.y + e.z + e.w)) : 0.f;
            const int lo = row * SA_RS + pc * 8;
            *reinterpret_cast<uint2*>(Ki + lo) = make_uint2(pack_bf16x2(kv.x, kv.y), pack_bf16x2(kv.z, kv.w));
            *reinterpret_cast<uint2*>(Vi + lo) = make_uint2(pack_bf16x2(vv.x, vv.y), pack_bf16x2(vv.z, vv.w));
            *reinterpret_cast<uint2*>(Qi + lo) = make_uint2(pack_bf16x2(e.x * inv, e.y * inv), pack_bf16x2(e.z * inv, e.w * inv));
            *reinterpret_cast<uint2*>(Di + lo) = make_uint2(pack_bf16x2(dv.x, dv.y), pack_bf16x2(dv.z, dv.w));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        bf16x8 ka[2], qa[2], vb[2], db[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ka[i] = sa_frag(Ki + troff0 + i * 32, Ki + troff1 + i * 32);      // rows d of tile i, K = pixels
            qa[i] = sa_frag(Qi + troff0 + i * 32, Qi + troff1 + i * 32);
            vb[i] = sa_frag(Vi + troff0 + i * 32, Vi + troff1 + i * 32);      // K = pixels, columns e of tile i
            db[i] = sa_frag(Di + troff0 + i * 32, Di + troff1 + i * 32);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                cacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[i], vb[j], cacc[i][j], 0, 0, 0);
                dacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[i], db[j], dacc[i][j], 0, 0, 0);
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // ---- combine the 4 waves ----
    ks4.x = colred_sum(ks4.x); ks4.y = colred_sum(ks4.y); ks4.z = colred_sum(ks4.z); ks4.w = colred_sum(ks4.w);
    if (rg == 0) *reinterpret_cast<float4*>(&ksm[w][pc * 4]) = ks4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {                     // accumulator (col r, rows 4q+e) -> [d][e] row-major
                part[w][0][(i * 16 + 4 * q + e) * 32 + j * 16 + r] = cacc[i][j][e];
                part[w][1][(i * 16 + 4 * q + e) * 32 + j * 16 + r] = dacc[i][j][e];
            }
    __syncthreads();
    if (tid < 32) ksum[tid] = ksm[0][tid] + ksm[1][tid] + ksm[2][tid] + ksm[3][tid];
    __syncthreads();
    float* out = P.A + ((size_t)n * P.heads + h) * SLA_A;
    const int pd = tid >> 3, e0 = (tid & 7) * 4;              // this thread: ctx[pd][e0..e0+3]
    float tsum = 0.f;
    const float isum = 1.0f / ksum[pd];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = pd * 32 + e0 + e;
        const float c = (part[0][0][i] + part[1][0][i] + part[2][0][i] + part[3][0][i]) * isum;
        const float dc = part[0][1][i] + part[1][1][i] + part[2][1][i] + part[3][1][i];
        out[i] = c; out[1024 + i] = dc; tsum += c * dc;
    }
    tsum += __shfl_xor(tsum, 1); tsum += __shfl_xor(tsum, 2); tsum += __shfl_xor(tsum, 4);
    if ((tid & 7) == 0) out[2048 + 64 + pd] = tsum;
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
        const size_t go = ((size_t)n * P.N + p) * P.dstride + h * 32;          // dq / dk / dv rows (possibly one interleaved buffer)
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
            *reinterpret_cast<float4*>(P.dq + go + d) = make_float4(q[d] * (dqs[d] - dot), q[d + 1] * (dqs[d + 1] - dot), q[d + 2] * (dqs[d + 2] - dot), q[d + 3] * (dqs[d + 3] - dot));
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
            *reinterpret_cast<float4*>(P.dk + go + d) = make_float4(dqs[d], dqs[d + 1], dqs[d + 2], dqs[d + 3]);
            *reinterpret_cast<float4*>(P.dv + go + d) = make_float4(outv[d], outv[d + 1], outv[d + 2], outv[d + 3]);
        }
    }
}

// ---- bf16-mode form of pass B: the four 32x32 mat-vecs per (pixel, head) as MFMAs ------------------------------------------------
// Everything is computed TRANSPOSED so that lane (c, q) of an accumulator owns pixel c and channels 4q..4q+3 (+16): its inputs
// are two float4 per tensor, its outputs two float4 per tensor, and softmax_D(q) is an in-lane + permlane reduction.
//   O^T = ctx^T qsm^T,  dqs^T = ctx dOut^T,  dv^T = dctx^T ksm^T,  dks^T = dctx v^T      (A = the per-head 32x32 matrix, B = pixel data)
// The K slots of a lane are permuted consistently on both operands: slot j of lane group q is channel 4q+j (j<4) / 16+4q+j-4.
// The eight A fragments of a head are built once per wave from an LDS copy of ctx / dctx and reused for all its pixel tiles.
__device__ __forceinline__ bf16x8 pack8_bf16(const float4& a, const float4& b) {
    const uint4 u = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
    return __builtin_bit_cast(bf16x8, u);
}

template <bool IO16>
__global__ __launch_bounds__(256) void sla_bwd_b16_kernel(SlaBwdArgs P) {
    __shared__ float cm[32][33], dm[32][33];                 // ctx[d][e], dctx[d][e] of the current head
    __shared__ float km[32], ksu[32], Tv[32];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int chunks = (P.N + 255) / 256;
    const int n = blockIdx.x / chunks, p0 = (blockIdx.x % chunks) * 256;
    const float L2E = 1.44269504088896f;
    for (int h = 0; h < P.heads; ++h) {
        const float* A = P.A + ((size_t)n * P.heads + h) * SLA_A;
        __syncthreads();
        for (int i = tid; i < 1024; i += 256) { cm[i >> 5][i & 31] = A[i]; dm[i >> 5][i & 31] = A[1024 + i]; }
        if (tid < 32) { km[tid] = A[2048 + tid]; ksu[tid] = 1.0f / A[2048 + 32 + tid]; Tv[tid] = A[2048 + 64 + tid]; }
        __syncthreads();
        // A fragments: row = 16 i + c of the matrix, K slots = this lane group's 8 channels
        bf16x8 a_ctxT[2], a_ctx[2], a_dctxT[2], a_dctx[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 16 * i + c;
            float4 lo, hi;
            lo = make_float4(cm[4 * q][row], cm[4 * q + 1][row], cm[4 * q + 2][row], cm[4 * q + 3][row]);                          // ctx^T[e=row][d slots]
            hi = make_float4(cm[16 + 4 * q][row], cm[17 + 4 * q][row], cm[18 + 4 * q][row], cm[19 + 4 * q][row]);
            a_ctxT[i] = pack8_bf16(lo, hi);
            lo = make_float4(cm[row][4 * q], cm[row][4 * q + 1], cm[row][4 * q + 2], cm[row][4 * q + 3]);                          // ctx[d=row][e slots]
            hi = make_float4(cm[row][16 + 4 * q], cm[row][17 + 4 * q], cm[row][18 + 4 * q], cm[row][19 + 4 * q]);
            a_ctx[i] = pack8_bf16(lo, hi);
            lo = make_float4(dm[4 * q][row], dm[4 * q + 1][row], dm[4 * q + 2][row], dm[4 * q + 3][row]);
            hi = make_float4(dm[16 + 4 * q][row], dm[17 + 4 * q][row], dm[18 + 4 * q][row], dm[19 + 4 * q][row]);
            a_dctxT[i] = pack8_bf16(lo, hi);
            lo = make_float4(dm[row][4 * q], dm[row][4 * q + 1], dm[row][4 * q + 2], dm[row][4 * q + 3]);
            hi = make_float4(dm[row][16 + 4 * q], dm[row][17 + 4 * q], dm[row][18 + 4 * q], dm[row][19 + 4 * q]);
            a_dctx[i] = pack8_bf16(lo, hi);
        }
        const float4 km0 = *reinterpret_cast<const float4*>(&km[4 * q]), km1 = *reinterpret_cast<const float4*>(&km[16 + 4 * q]);
        const float4 ki0 = *reinterpret_cast<const float4*>(&ksu[4 * q]), ki1 = *reinterpret_cast<const float4*>(&ksu[16 + 4 * q]);
        const float4 T0 = *reinterpret_cast<const float4*>(&Tv[4 * q]), T1 = *reinterpret_cast<const float4*>(&Tv[16 + 4 * q]);
        for (int t = w; t < 16; t += 4) {                     // 16-pixel tiles of this workgroup's 256 pixels
            const int p = p0 + t * 16 + c;
            const bool ok = p < P.N;
            const size_t o = ((size_t)n * P.N + (ok ? p : 0)) * 256 + h * 32 + 4 * q;
            const size_t go = ((size_t)n * P.N + (ok ? p : 0)) * P.dstride + h * 32 + 4 * q;
            float4 q0 = load4_f32_or_bf16(P.q, o, IO16), q1 = load4_f32_or_bf16(P.q, o + 16, IO16);
            float4 k0 = load4_f32_or_bf16(P.k, o, IO16), k1 = load4_f32_or_bf16(P.k, o + 16, IO16);
            const float4 v0 = load4_f32_or_bf16(P.v, o, IO16), v1 = load4_f32_or_bf16(P.v, o + 16, IO16);
            const float4 d0 = load4_f32_or_bf16(P.dOut, o, IO16), d1 = load4_f32_or_bf16(P.dOut, o + 16, IO16);
            // softmax over the 32 channels of this pixel (8 in-lane, 4 lanes)
            const float mx = max_q(fmaxf(fmaxf(fmaxf(q0.x, q0.y), fmaxf(q0.z, q0.w)), fmaxf(fmaxf(q1.x, q1.y), fmaxf(q1.z, q1.w))));
            q0.x = __builtin_amdgcn_exp2f((q0.x - mx) * L2E); q0.y = __builtin_amdgcn_exp2f((q0.y - mx) * L2E);
            q0.z = __builtin_amdgcn_exp2f((q0.z - mx) * L2E); q0.w = __builtin_amdgcn_exp2f((q0.w - mx) * L2E);
            q1.x = __builtin_amdgcn_exp2f((q1.x - mx) * L2E); q1.y = __builtin_amdgcn_exp2f((q1.y - mx) * L2E);
            q1.z = __builtin_amdgcn_exp2f((q1.z - mx) * L2E); q1.w = __builtin_amdgcn_exp2f((q1.w - mx) * L2E);
            const float inv = __builtin_amdgcn_rcpf(reduce_q(q0.x + q0.y + q0.z + q0.w + q1.x + q1.y + q1.z + q1.w));
            q0.x *= inv; q0.y *= inv; q0.z *= inv; q0.w *= inv; q1.x *= inv; q1.y *= inv; q1.z *= inv; q1.w *= inv;      // qsm
            k0.x = __builtin_amdgcn_exp2f((k0.x - km0.x) * L2E) * ki0.x; k0.y = __builtin_amdgcn_exp2f((k0.y - km0.y) * L2E) * ki0.y;
            k0.z = __builtin_amdgcn_exp2f((k0.z - km0.z) * L2E) * ki0.z; k0.w = __builtin_amdgcn_exp2f((k0.w - km0.w) * L2E) * ki0.w;
            k1.x = __builtin_amdgcn_exp2f((k1.x - km1.x) * L2E) * ki1.x; k1.y = __builtin_amdgcn_exp2f((k1.y - km1.y) * L2E) * ki1.y;
            k1.z = __builtin_amdgcn_exp2f((k1.z - km1.z) * L2E) * ki1.z; k1.w = __builtin_amdgcn_exp2f((k1.w - km1.w) * L2E) * ki1.w;      // ksm
            const bf16x8 bq = pack8_bf16(q0, q1), bk = pack8_bf16(k0, k1), bv = pack8_bf16(v0, v1), bd = pack8_bf16(d0, d1);
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 oT[2], dqsT[2], dvT[2], dksT[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                oT[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_ctxT[i], bq, z, 0, 0, 0);
                dqsT[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_ctx[i], bd, z, 0, 0, 0);
                dvT[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_dctxT[i], bk, z, 0, 0, 0);
                dksT[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_dctx[i], bv, z, 0, 0, 0);
            }
            float dot = q0.x * dqsT[0][0] + q0.y * dqsT[0][1] + q0.z * dqsT[0][2] + q0.w * dqsT[0][3]
                      + q1.x * dqsT[1][0] + q1.y * dqsT[1][1] + q1.z * dqsT[1][2] + q1.w * dqsT[1][3];
            dot = reduce_q(dot);
            if (ok) {
                const int h16 = IO16;
                store4_f32_or_bf16(P.O, o, make_float4(oT[0][0], oT[0][1], oT[0][2], oT[0][3]), h16);
                store4_f32_or_bf16(P.O, o + 16, make_float4(oT[1][0], oT[1][1], oT[1][2], oT[1][3]), h16);
                store4_f32_or_bf16(P.dv, go, make_float4(dvT[0][0], dvT[0][1], dvT[0][2], dvT[0][3]), h16);
                store4_f32_or_bf16(P.dv, go + 16, make_float4(dvT[1][0], dvT[1][1], dvT[1][2], dvT[1][3]), h16);
                store4_f32_or_bf16(P.dq, go, make_float4(q0.x * (dqsT[0][0] - dot), q0.y * (dqsT[0][1] - dot), q0.z * (dqsT[0][2] - dot), q0.w * (dqsT[0][3] - dot)), h16);
                store4_f32_or_bf16(P.dq, go + 16, make_float4(q1.x * (dqsT[1][0] - dot), q1.y * (dqsT[1][1] - dot), q1.z * (dqsT[1][2] - dot), q1.w * (dqsT[1][3] - dot)), h16);
                store4_f32_or_bf16(P.dk, go, make_float4(k0.x * (dksT[0][0] - T0.x), k0.y * (dksT[0][1] - T0.y), k0.z * (dksT[0][2] - T0.z), k0.w * (dksT[0][3] - T0.w)), h16);
                store4_f32_or_bf16(P.dk, go + 16, make_float4(k1.x * (dksT[1][0] - T1.x), k1.y * (dksT[1][1] - T1.y), k1.z * (dksT[1][2] - T1.z), k1.w * (dksT[1][3] - T1.w)), h16);
            }
        }
    }
}

hipError_t launch_attn_core_bwd(const AttnBwdArgs& a, hipStream_t st) {
    if (a.bf16_mma && a.L <= 16) {
        const long blocks = (a.nseq + 3) / 4;
        if (a.io_bf16) hipLaunchKernelGGL(attn_core_bwd16_kernel<true>, dim3((unsigned)blocks), dim3(256), 4 * 4 * 16 * AB_RS, st, a);
        else hipLaunchKernelGGL(attn_core_bwd16_kernel<false>, dim3((unsigned)blocks), dim3(256), 4 * 4 * 16 * AB_RS, st, a);
        return hipGetLastError();
    }
    const size_t lds = ((size_t)4 * a.L * 33 + 2 * a.L * (a.L + 1)) * 4;
    auto kfn = attn_core_bwd_kernel;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kfn, dim3((unsigned)a.nseq, (unsigned)a.heads), dim3(256), lds, st, a);
    return hipGetLastError();
}

size_t sla_bwd_scratch_floats(int NF, int heads) { return (size_t)NF * heads * SLA_A; }

hipError_t launch_sla_bwd(const SlaBwdArgs& a, hipStream_t st) {
    if (a.bf16_mma && a.io_bf16) hipLaunchKernelGGL(sla_bwd_a16_kernel<true>, dim3(a.NF, a.heads), dim3(256), 0, st, a);
    else if (a.bf16_mma) hipLaunchKernelGGL(sla_bwd_a16_kernel<false>, dim3(a.NF, a.heads), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(sla_bwd_a_kernel, dim3(a.NF, a.heads), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int tiles = (a.N + 255) / 256;
    if (a.bf16_mma && a.io_bf16) hipLaunchKernelGGL(sla_bwd_b16_kernel<true>, dim3(a.NF * tiles), dim3(256), 0, st, a);
    else if (a.bf16_mma) hipLaunchKernelGGL(sla_bwd_b16_kernel<false>, dim3(a.NF * tiles), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(sla_bwd_b_kernel, dim3(a.NF * tiles), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace vdx
