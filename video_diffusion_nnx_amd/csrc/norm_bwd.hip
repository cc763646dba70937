// Backward of the fused normalisation/activation stages of ResnetBlock (gfx950), HBM-bound, float4-vectorised.
//
// Forward stage (reference modules.py:171-179, 233-243):   xh = (y - mu_g) * rstd_g ;  u = gamma * xh + beta ;
//     z = u * (1 + s_bc) + sh_bc ;  act = SiLU(z)          [s, sh = 0 for Block 2 / no time MLP]
// and, in the block tail, out = act + LayerNorm_C(r).
// Given dact (= dL/dact, or dL/dout for the tail):
//   reduce    R0[b,c] = sum_pix dz ,  R1[b,c] = sum_pix dz * xh   (dz = dact * SiLU'(z)); LN: dgamma_ln, dbeta_ln
//   finalize  dgamma, dbeta, (ds, dsh), per-group S1 = sum_c k R0 / n, S2 = sum_c k R1 / n   with k = (1+s) gamma
//   apply     dy = rstd_g * (k * dz - S1 - xh * S2) ;  LN: dr = rstd_p * (g*dout - mean_c(g*dout) - rh * mean_c(g*dout*rh))
// (jax.value_and_grad of the same expressions, reference trainer.py:361.)
#include "vdx_common.h"
#include <stdlib.h>
#include "vdx_internal.h"

namespace vdx {

__device__ __forceinline__ float dsilu_f(float z) {
    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
    return sg * (1.0f + z * (1.0f - sg));
}

// shared by reduce and apply: per-channel tables for sample b in LDS
struct NbTables { float* mu; float* rs; float* a; float* d; };

__device__ __forceinline__ void nb_build_tables(const NormBwdArgs& P, int b, float* gm, NbTables T) {
    const int tid = threadIdx.x, C = P.C;
    gn_mean_rstd_wg(P.stats, b, P.groups, (double)P.pix_per_sample * (C / P.groups), gm, tid, 256);
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int g = c / (C / P.groups);
        const float m = gm[2 * g], rs = gm[2 * g + 1];
        float sc = 1.f, sh = 0.f;
        if (P.ss) { sc = P.ss[(size_t)b * P.ss_stride + c] + 1.f; sh = P.ss[(size_t)b * P.ss_stride + C + c]; }
        const float a = rs * P.gamma[c] * sc;
        T.mu[c] = m; T.rs[c] = rs; T.a[c] = a; T.d[c] = (P.beta[c] - m * rs * P.gamma[c]) * sc + sh;
    }
    __syncthreads();
}

#ifndef VDX_NB_U
#define VDX_NB_U 4            // pixels per lane group and pass of the reduce kernel (VPL = 1; half of it for wider rows)
#endif
template <int VPL>
__global__ __launch_bounds__(256) void norm_bwd_reduce_kernel(NormBwdArgs P) {
    __shared__ float mu[1024], rsd[1024], ta[1024], td[1024];
    __shared__ float gm[64];
    __shared__ float red[256 * 4];
    const int tid = threadIdx.x, C = P.C, b = blockIdx.y;
    NbTables T{mu, rsd, ta, td};
    nb_build_tables(P, b, gm, T);
    const int LPP = P.lpp, ppb = 256 / LPP, sub = tid % LPP, pl = tid / LPP;
    const float invC = 1.0f / (float)C;
    float4 r0[VPL], r1[VPL], g0[VPL], g1[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) { r0[v] = r1[v] = g0[v] = g1[v] = make_float4(0.f, 0.f, 0.f, 0.f); }
    // U pixels per lane group and pass: every load of the pass (r, y, dact of U pixels) is issued before the first use, so a wave
    // has 3 * U * VPL 16-byte loads in flight (the launch has < 1 wave per SIMD: nothing else hides the latency)
    constexpr int U = VPL == 1 ? VDX_NB_U : VDX_NB_U / 2;
    const long pstride = (long)gridDim.x * ppb;
    for (long pix0 = (long)blockIdx.x * ppb + pl; pix0 < P.pix_per_sample; pix0 += pstride * U) {
        float4 rr[U][VPL], yv4[U][VPL], da4[U][VPL];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long pix = pix0 + u * pstride;
            const bool pv = pix < P.pix_per_sample;
            const size_t base = ((size_t)b * P.pix_per_sample + (pv ? pix : 0)) * C;
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                const int c = (v * LPP + sub) * 4;
                const size_t e = base + (c < C ? c : 0);
                rr[u][v] = P.r ? load4_f32_or_bf16(P.r, e, P.r_bf16) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (P.y_bf16) {
                    const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(P.y) + e * 2);
                    yv4[u][v] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), 0.f, 0.f);      // raw, widened below
                } else yv4[u][v] = *reinterpret_cast<const float4*>(P.y + e);
                da4[u][v] = *reinterpret_cast<const float4*>(P.dact + e);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long pix = pix0 + u * pstride;
            const bool pv = pix < P.pix_per_sample;                 // (uniform per lane group; masked lanes contribute zeros)
            float mean = 0.f, rstd = 0.f;
            if (P.r) {                                               // LayerNorm statistics of this pixel
                float s = 0.f, ss = 0.f;
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    const int c = (v * LPP + sub) * 4;
                    if (c < C) {
                        s += rr[u][v].x + rr[u][v].y + rr[u][v].z + rr[u][v].w;
                        ss += rr[u][v].x * rr[u][v].x + rr[u][v].y * rr[u][v].y + rr[u][v].z * rr[u][v].z + rr[u][v].w * rr[u][v].w;
                    }
                }
                for (int o = 1; o < LPP; o <<= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
                mean = s * invC;
                rstd = rsqrtf(fmaxf(ss * invC - mean * mean, 0.f) + NORM_EPS);
            }
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                const int c = (v * LPP + sub) * 4;
                if (c < C && pv) {
                    float4 y = yv4[u][v];
                    if (P.y_bf16) {
                        const unsigned t0 = __float_as_uint(y.x), t1 = __float_as_uint(y.y);
                        y = make_float4(__uint_as_float(t0 << 16), __uint_as_float(t0 & 0xFFFF0000u), __uint_as_float(t1 << 16), __uint_as_float(t1 & 0xFFFF0000u));
                    }
                    const float4 da = da4[u][v];
                    const float yv[4] = {y.x, y.y, y.z, y.w}, dv[4] = {da.x, da.y, da.z, da.w};
                    float* p0 = &r0[v].x; float* p1 = &r1[v].x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float z = fmaf(yv[e], ta[c + e], td[c + e]);
                        const float dz = dv[e] * dsilu_f(z);
                        p0[e] += dz;
                        p1[e] += dz * (yv[e] - mu[c + e]) * rsd[c + e];
                    }
                    if (P.r) {
                        const float rv[4] = {rr[u][v].x, rr[u][v].y, rr[u][v].z, rr[u][v].w};
                        float* q0 = &g0[v].x; float* q1 = &g1[v].x;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { q0[e] += dv[e] * (rv[e] - mean) * rstd; q1[e] += dv[e]; }
                    }
                }
            }
        }
    }
    // ---- reduce over the workgroup's pixel lanes, then ONE plain store per (channel, quantity) into this workgroup's row of the
    //      partials [B][workgroups][4][C]; the finalize pass sums the rows.  (Round-2 first form: fp32 atomics onto one [B][C] row per
    //      quantity -- contention-bound beyond ~64 workgroups per sample, so the pass ran on 192 workgroups, < 1 wave per SIMD, at
    //      1.3-1.7 TB/s; without the atomics it runs on up to 256 workgroups per sample.) ----
    auto flush = [&](const float4& val, int c, int qty) {
        __syncthreads();
        *reinterpret_cast<float4*>(red + tid * 4) = val;
        __syncthreads();
        if (pl == 0 && c < C) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < ppb; ++k) {
                const float4 u = *reinterpret_cast<const float4*>(red + (k * LPP + sub) * 4);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            *reinterpret_cast<float4*>(P.R + ((((size_t)b * gridDim.x + blockIdx.x) * 4 + qty) * C + c)) = t;
        }
    };
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const int c = (v * LPP + sub) * 4;
        flush(r0[v], c, 0);
        flush(r1[v], c, 1);
        if (P.r) { flush(g0[v], c, 2); flush(g1[v], c, 3); }
    }
}

// grid = (channel tiles, batch).  Sums the reduce pass's per-workgroup partials in a fixed order and turns them into parameter
// gradients, (ds, dsh) and the per-group correction terms G[b][g][2].  A workgroup owns CT channels = whole GroupNorm groups (CT is
// a multiple of C / groups), so its groups' sums need nobody else; its 256 threads are CT channel lanes x 256 / CT slices of the
// partial rows.  (With one workgroup per sample the pass took 78 us for 256 rows: 38 launches per train step.)
__global__ __launch_bounds__(256) void norm_bwd_finalize_kernel(NormBwdArgs P, int CT) {
    __shared__ float s1[32], s2[32];
    __shared__ float red[4][256];
    const int tid = threadIdx.x, C = P.C, b = blockIdx.y, nwg = P.nwg;
    const int cpg = C / P.groups;
    const int c0 = blockIdx.x * CT, g0 = c0 / cpg, ngl = CT / cpg;       // first channel / first group / groups of this workgroup
    if (tid < 32) { s1[tid] = 0.f; s2[tid] = 0.f; }
    const int CW = min(CT, 256), NS = 256 / CW;                         // channel lanes, row slices
    const int cl = tid % CW, sl = tid / CW;
    const int nq = P.r ? 4 : 2;
    for (int cc = 0; cc < CT; cc += CW) {                                // (one trip unless a group is wider than 256 channels)
        const int c = c0 + cc + cl;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const bool mine = c < C && cc + cl < CT;
        if (sl < NS && mine) {
            const float* p = P.R + (size_t)b * nwg * 4 * C + c;
            // 4 rows (16 loads) in flight per thread: the rows were written by other XCDs a moment ago, every load is a trip to the
            // memory side, and a plain row loop makes 8 dependent trips of it
            for (int w = sl; w < nwg; w += 4 * NS) {
                float t[4][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int wj = min(w + j * NS, nwg - 1);
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[j][k] = (k < nq) ? p[((size_t)wj * 4 + k) * C] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (w + j * NS < nwg) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) a[k] += t[j][k];
                    }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) red[k][tid] = a[k];
        __syncthreads();
        float v1 = 0.f, v2 = 0.f;
        if (sl == 0 && mine) {
            float R0 = 0.f, R1 = 0.f, q0 = 0.f, q1 = 0.f;
            for (int k = 0; k < NS; ++k) { R0 += red[0][k * CW + cl]; R1 += red[1][k * CW + cl]; q0 += red[2][k * CW + cl]; q1 += red[3][k * CW + cl]; }
            float sc = 1.f;
            if (P.ss) sc = P.ss[(size_t)b * P.ss_stride + c] + 1.f;
            const float ga = P.gamma[c], be = P.beta[c];
            if (P.dgp) {                                       // deterministic mode: per-sample rows, added over the samples in order by the apply pass
                float* d = P.dgp + (size_t)b * 4 * C;
                d[c] = sc * R1; d[C + c] = sc * R0;
                if (P.r) { d[2 * C + c] = q0; d[3 * C + c] = q1; }
            } else {
                atomicAdd(P.d_gamma + c, sc * R1);
                atomicAdd(P.d_beta + c, sc * R0);
                if (P.r) { atomicAdd(P.d_ln_gamma + c, q0); atomicAdd(P.d_ln_beta + c, q1); }
            }
            if (P.dss) { P.dss[(size_t)b * 2 * C + c] = ga * R1 + be * R0; P.dss[(size_t)b * 2 * C + C + c] = R0; }
            v1 = sc * ga * R0; v2 = sc * ga * R1;
            if (!P.dgp) {
                const int gl = c / cpg - g0;
                atomicAdd(&s1[gl], v1);
                atomicAdd(&s2[gl], v2);
            }
        }
        if (P.dgp) {                                           // (uniform) the groups' sums over their channels, one thread per group, in channel order
            __syncthreads();
            if (sl == 0 && mine) { red[0][cl] = v1; red[1][cl] = v2; }
            __syncthreads();
            if (tid < ngl) {
                const int lo = (g0 + tid) * cpg - c0 - cc;
                for (int i = 0; i < cpg; ++i) {
                    const int l = lo + i;
                    if (l >= 0 && l < CW && cc + l < CT && c0 + cc + l < C) { s1[tid] += red[0][l]; s2[tid] += red[1][l]; }
                }
            }
        }
    }
    __syncthreads();
    if (tid < ngl && g0 + tid < P.groups) {
        const float n = (float)((double)P.pix_per_sample * cpg);
        P.G[((size_t)b * P.groups + g0 + tid) * 2] = s1[tid] / n;
        P.G[((size_t)b * P.groups + g0 + tid) * 2 + 1] = s2[tid] / n;
    }
}

template <int VPL>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(NormBwdArgs P) {
    __shared__ float mu[1024], rsd[1024], ta[1024], td[1024];
    __shared__ float gm[64], gs[64];
    const int tid = threadIdx.x, C = P.C, b = blockIdx.y;
    NbTables T{mu, rsd, ta, td};
    nb_build_tables(P, b, gm, T);
    if (tid < 2 * P.groups) gs[tid] = P.G[(size_t)b * P.groups * 2 + tid];
    if (P.dgp && blockIdx.x == 0 && blockIdx.y == 0) {
        // deterministic mode: the parameter gradients = the finalize pass's per-sample rows added in sample order (one workgroup, once)
        const int nq = P.r ? 4 : 2;
        for (int i = tid; i < nq * C; i += 256) {
            const int qy = i / C, c = i - qy * C;
            float t = 0.f;
            for (int bb = 0; bb < P.batch; ++bb) t += P.dgp[((size_t)bb * 4 + qy) * C + c];
            float* dst = qy == 0 ? P.d_gamma : qy == 1 ? P.d_beta : qy == 2 ? P.d_ln_gamma : P.d_ln_beta;
            dst[c] += t;
        }
    }
    __syncthreads();
    const int cpg = C / P.groups;
    const int LPP = P.lpp, ppb = 256 / LPP, sub = tid % LPP, pl = tid / LPP;
    const float invC = 1.0f / (float)C;
    for (long pix = (long)blockIdx.x * ppb + pl; pix < P.pix_per_sample; pix += (long)gridDim.x * ppb) {
        const size_t base = ((size_t)b * P.pix_per_sample + pix) * C;
        float4 da[VPL], rr[VPL];
        float mean = 0.f, rstd = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int c = (v * LPP + sub) * 4;
            da[v] = rr[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < C) da[v] = *reinterpret_cast<const float4*>(P.dact + base + c);
        }
        if (P.r) {
            float s = 0.f, ss = 0.f;
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                const int c = (v * LPP + sub) * 4;
                if (c < C) {
                    rr[v] = load4_f32_or_bf16(P.r, base + c, P.r_bf16);
                    s += rr[v].x + rr[v].y + rr[v].z + rr[v].w;
                    ss += rr[v].x * rr[v].x + rr[v].y * rr[v].y + rr[v].z * rr[v].z + rr[v].w * rr[v].w;
                }
            }
            for (int o = 1; o < LPP; o <<= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
            mean = s * invC;
            rstd = rsqrtf(fmaxf(ss * invC - mean * mean, 0.f) + NORM_EPS);
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                const int c = (v * LPP + sub) * 4;
                if (c < C) {
                    const float4 lg = *reinterpret_cast<const float4*>(P.ln_gamma + c);
                    const float gd[4] = {lg.x * da[v].x, lg.y * da[v].y, lg.z * da[v].z, lg.w * da[v].w};
                    const float rv[4] = {rr[v].x, rr[v].y, rr[v].z, rr[v].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { m1 += gd[e]; m2 += gd[e] * (rv[e] - mean) * rstd; }
                }
            }
            for (int o = 1; o < LPP; o <<= 1) { m1 += __shfl_xor(m1, o); m2 += __shfl_xor(m2, o); }
            m1 *= invC; m2 *= invC;
        }
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int c = (v * LPP + sub) * 4;
            if (c < C) {
                const float4 y = load4_f32_or_bf16(P.y, base + c, P.y_bf16);
                const float yv[4] = {y.x, y.y, y.z, y.w}, dv[4] = {da[v].x, da[v].y, da[v].z, da[v].w};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int g = (c + e) / cpg;
                    const float z = fmaf(yv[e], ta[c + e], td[c + e]);
                    const float dz = dv[e] * dsilu_f(z);
                    const float xh = (yv[e] - mu[c + e]) * rsd[c + e];
                    o[e] = ta[c + e] * dz - rsd[c + e] * (gs[2 * g] + xh * gs[2 * g + 1]);     // ta = rstd * gamma * (1+s)
                }
                store4_f32_or_bf16(P.dy, base + c, make_float4(o[0], o[1], o[2], o[3]), P.dy_bf16);
                if (P.r) {
                    const float4 lg = *reinterpret_cast<const float4*>(P.ln_gamma + c);
                    const float lgv[4] = {lg.x, lg.y, lg.z, lg.w}, rv[4] = {rr[v].x, rr[v].y, rr[v].z, rr[v].w};
                    float d[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = rstd * (lgv[e] * dv[e] - m1 - (rv[e] - mean) * rstd * m2);
                    *reinterpret_cast<float4*>(P.dr + base + c) = make_float4(d[0], d[1], d[2], d[3]);
                }
            }
        }
    }
}

// workgroups per sample of the reduce pass: every workgroup walks >= 2 passes of U pixels per lane group, at most 256 per sample
// (measured at the N shape, B = 4: 128 per sample 55 us per launch, 256: 49-50, 512: 50)
static int norm_bwd_reduce_wgs(int C, long pix_per_sample) {
    const int quads = C / 4;
    int lpp = 1;
    while (lpp < quads && lpp < 64) lpp <<= 1;
    const int vpl = (quads + lpp - 1) / lpp, ppb = 256 / lpp, U = vpl == 1 ? VDX_NB_U : VDX_NB_U / 2;
    return (int)std::max<long>(1, std::min<long>(256, (pix_per_sample + (long)ppb * U * 2 - 1) / ((long)ppb * U * 2)));
}

// floats of NormBwdArgs::R (+ G behind it): [B][workgroups][4][C] partials + [B][32 groups][2]
size_t norm_bwd_scratch_floats(int C, int batch, long pix_per_sample) {
    return (size_t)batch * norm_bwd_reduce_wgs(C, pix_per_sample) * 4 * C + (size_t)batch * 64;
}

hipError_t launch_norm_bwd(NormBwdArgs a, hipStream_t st) {
    const int quads = a.C / 4;
    int lpp = 1;
    while (lpp < quads && lpp < 64) lpp <<= 1;
    a.lpp = lpp;
    const int vpl = (quads + lpp - 1) / lpp;
    const int ppb = 256 / lpp;
    const int gx = norm_bwd_reduce_wgs(a.C, a.pix_per_sample);
    a.nwg = gx;
    dim3 grid(gx, a.batch);
    switch (vpl) {
        case 1: hipLaunchKernelGGL(norm_bwd_reduce_kernel<1>, grid, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(norm_bwd_reduce_kernel<2>, grid, dim3(256), 0, st, a); break;
        case 3: case 4: hipLaunchKernelGGL(norm_bwd_reduce_kernel<4>, grid, dim3(256), 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    {   // channel tile of the finalize pass: whole groups, >= 8 channels (<= 32 groups of it in LDS), at most 256
        const int cpg = a.C / a.groups;
        int ct = cpg;
        while (ct < 8 && ct * 2 <= a.C && a.C % (ct * 2) == 0) ct *= 2;
        if (a.C % ct || ct / cpg > 32) return hipErrorInvalidValue;
        hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3(a.C / ct, a.batch), dim3(256), 0, st, a, ct);
    }
    const int gx2 = (int)std::max<long>(1, std::min<long>((a.pix_per_sample + ppb - 1) / ppb, 2048));
    dim3 grid2(gx2, a.batch);
    switch (vpl) {
        case 1: hipLaunchKernelGGL(norm_bwd_apply_kernel<1>, grid2, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(norm_bwd_apply_kernel<2>, grid2, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(norm_bwd_apply_kernel<4>, grid2, dim3(256), 0, st, a); break;
    }
    return hipGetLastError();
}

}  // namespace vdx
