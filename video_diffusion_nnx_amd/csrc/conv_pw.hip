// Pointwise (1x1) convolution of the wide levels for bf16 tensors (gfx950, bf16 MFMA operands):
//     y[p, :] = W . concat(x0[p, :], x1[p, :]) + b (+ res[p, :])
// = the `res_conv` of ResnetBlocks (modules.py:219-222), the attention `to_out` projections (modules.py:312-326) and the
// SpatialLinearAttention `to_out` (modules.py:121-129) of the levels with >= 128 channels in bf16 activation storage.
//
// The generic implicit-GEMM kernel stages a 256-pixel x 64-channel input tile in LDS per K chunk and synchronises the workgroup per
// chunk: with ONE tap that is 16 MFMAs per wave between two barriers (10 launches of 84 us per step at B = 64, ~240 TFLOP/s, for
// GEMMs that are HBM-bound at ~40 us).  Here the roles are those of the per-head attention kernels: a workgroup owns a tile of
// output channels and keeps its weight rows [rows][Cin] RESIDENT in LDS for a whole range of pixels (persistent: one 8-wave
// workgroup per CU); every wave walks its own groups of 32 pixels with no workgroup barrier: the x rows go from global memory
// straight into MFMA B fragments (lane (pixel, q) = 16 bytes of a 64-byte K chunk; 8 loads of one K block are in flight while the 4 K
// steps of the previous block are multiplied), weight fragments come from LDS and each feeds two MFMAs.  The weight image's rows are
// permuted as in resblock_tail_rc16_kernel so that a lane's accumulators are whole 16-byte pieces (8 consecutive channels) of the
// bf16 rows of y / res.  XCD-aware decode: the workgroups of the output-channel tiles of one pixel range are 8 ids apart (same XCD,
// same L2), so x is fetched from HBM once.
#include "vdx_common.h"
#include "vdx_internal.h"

namespace vdx {

typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));

// ROWS output channels per workgroup (64 or 128); K = Cin in blocks of KS steps of 32: KS = 4 (Cin a multiple of 128) or KS = 2 (Cin = 64:
// the q / k / v projections the SLA backward recomputes at the widest level -- one block per pixel group, so the NEXT GROUP's block is
// in flight during the MFMAs); OUT32: y and res are fp32 tensors (the dx projections of the attention / SLA backward: bf16 dq|dk|dv in,
// fp32 gradient out).
template <int ROWS, int KS = 4, bool OUT32 = false>
__global__ __launch_bounds__(512) void conv1x1_pw_kernel(const ConvArgs P, const int nct, const int groups_per_range, const int ngroups) {
    using M = Mma<MODE_BF16>;
    constexpr int TM = ROWS / 16, TN = 2;
    constexpr int KBB = KS * 64;                           // bytes of one K block of a pixel row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Cin = P.C0 + P.C1, nkb = Cin / (32 * KS);
    const int WRS = Cin * 2 + 16;                          // LDS row stride of the weight image (bytes)
    char* Wl = smem;                                       // [ROWS, A-tile order][WRS]
    float* bl = reinterpret_cast<float*>(Wl + ROWS * WRS); // [ROWS] bias, channel order

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lp = lane & 15, q = lane >> 4;
    const int ct = (blockIdx.x >> 3) % nct, range = ((blockIdx.x >> 3) / nct) * 8 + (blockIdx.x & 7);
    const int co0 = ct * ROWS;
    const int g0 = range * groups_per_range, g1 = min(g0 + groups_per_range, ngroups);
    if (g0 >= g1) return;

    {   // weight image: A-tile row (tm, r') = output channel co0 + (tm >> 1) * 32 + (r' >> 2) * 8 + (tm & 1) * 4 + (r' & 3)
        const int cpr = Cin >> 3;                          // 16-byte pieces per row
        const char* wsrc = reinterpret_cast<const char*>(P.wp);
        for (int i = tid; i < ROWS * cpr; i += 512) {
            const int row = i / cpr, c = i - row * cpr;
            const int tm = row >> 4, rr = row & 15;
            const int co = co0 + (tm >> 1) * 32 + (rr >> 2) * 8 + (tm & 1) * 4 + (rr & 3);
            *reinterpret_cast<uint4*>(Wl + row * WRS + c * 16) = *reinterpret_cast<const uint4*>(wsrc + ((size_t)(P.wrow0 + co) * P.CinPad + c * 8) * 2);
        }
        for (int i = tid; i < ROWS; i += 512) bl[i] = P.bias ? P.bias[co0 + i] : 0.f;
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x0), 0, P.x0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.C1 ? P.x1 : P.x0), 0, P.C1 ? P.x1_bytes : P.x0_bytes, 0x00020000);
    const int nkb0 = P.C0 / (32 * KS);                     // K blocks in x0 (C0 is a multiple of 128 when there is a second tensor)
    const unsigned rb0 = P.C0 * 2, rb1 = P.C1 * 2;

    // one K block (KS steps x TN pixel tiles) of B fragments: 2 KS buffer loads, all issued before the first use
    auto load_block = [&](int g, int kb, pu32x4 (&xf)[KS][TN]) {
        const bool second = kb >= nkb0;                    // (uniform)
        const unsigned rb = second ? rb1 : rb0;
        const unsigned base = (unsigned)(g * 32 + lp) * rb + (unsigned)((second ? kb - nkb0 : kb) * KBB + q * 16);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const unsigned off = base + tn * 16 * rb + s * 64;
                xf[s][tn] = second ? __builtin_amdgcn_raw_buffer_load_b128(rx1, off, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(rx0, off, 0, 0);
            }
    };
    f32x4 acc[TM][TN];
    auto mma_block = [&](int kb, const pu32x4 (&xf)[KS][TN]) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint4 bf[TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bf[tn] = make_uint4(xf[s][tn].x, xf[s][tn].y, xf[s][tn].z, xf[s][tn].w);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const uint4 af = *reinterpret_cast<const uint4*>(Wl + (tm * 16 + lp) * WRS + kb * KBB + s * 64 + q * 16);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) M::mma(acc[tm][tn], af, bf[tn]);
            }
        }
    };

    char* const yb = reinterpret_cast<char*>(P.y);
    const char* const resb = reinterpret_cast<const char*>(P.res);
    // ROWS = 64: two fragment sets, the next K block's loads in flight during the MFMAs of the current one.  ROWS = 128 (64
    // accumulator registers): one set -- the second would spill -- and the other wave of the SIMD covers the load latency.
    // KS = 2 (one block per group): two sets, the next GROUP's block in flight.
    constexpr bool PP = ROWS == 64 && KS == 4;
    auto init_acc = [&]() {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const float4 b4 = *reinterpret_cast<const float4*>(bl + (tm >> 1) * 32 + q * 8 + (tm & 1) * 4);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = f32x4{b4.x, b4.y, b4.z, b4.w};
        }
    };
    // epilogue: lane (pixel, q) holds channels co0 + 32 j + 8 q .. + 7 in acc[2 j][tn], acc[2 j + 1][tn]
    auto store_group = [&](int g) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const size_t pix = (size_t)g * 32 + tn * 16 + lp;
#pragma unroll
            for (int j = 0; j < TM / 2; ++j) {
                const size_t el = pix * P.Cout + co0 + j * 32 + q * 8;
                float o[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) { o[k] = acc[2 * j][tn][k]; o[4 + k] = acc[2 * j + 1][tn][k]; }
                if constexpr (OUT32) {
                    if (resb) {
                        const float4 r0 = *reinterpret_cast<const float4*>(resb + el * 4), r1 = *reinterpret_cast<const float4*>(resb + el * 4 + 16);
                        o[0] += r0.x; o[1] += r0.y; o[2] += r0.z; o[3] += r0.w; o[4] += r1.x; o[5] += r1.y; o[6] += r1.z; o[7] += r1.w;
                    }
                    *reinterpret_cast<float4*>(yb + el * 4) = make_float4(o[0], o[1], o[2], o[3]);
                    *reinterpret_cast<float4*>(yb + el * 4 + 16) = make_float4(o[4], o[5], o[6], o[7]);
                } else {
                    const size_t e = el * 2;
                    if (resb) {
                        const uint4 r = *reinterpret_cast<const uint4*>(resb + e);
                        o[0] += __uint_as_float(r.x << 16); o[1] += __uint_as_float(r.x & 0xFFFF0000u);
                        o[2] += __uint_as_float(r.y << 16); o[3] += __uint_as_float(r.y & 0xFFFF0000u);
                        o[4] += __uint_as_float(r.z << 16); o[5] += __uint_as_float(r.z & 0xFFFF0000u);
                        o[6] += __uint_as_float(r.w << 16); o[7] += __uint_as_float(r.w & 0xFFFF0000u);
                    }
                    *reinterpret_cast<uint4*>(yb + e) = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
                }
            }
        }
    };
    if constexpr (KS == 2) {
        pu32x4 xa[KS][TN], xb[KS][TN];
        int g = g0 + wave;
        if (g < g1) load_block(g, 0, xa);
        for (; g < g1; g += 16) {
            __asm__ volatile("" ::: "memory");             // bias / weight fragments are re-read from LDS per group, not hoisted
            if (g + 8 < g1) load_block(g + 8, 0, xb);
            init_acc(); mma_block(0, xa); store_group(g);
            if (g + 8 >= g1) break;
            if (g + 16 < g1) load_block(g + 16, 0, xa);
            init_acc(); mma_block(0, xb); store_group(g + 8);
        }
    } else {
        pu32x4 xa[KS][TN], xb[PP ? KS : 1][TN];
        for (int g = g0 + wave; g < g1; g += 8) {
            __asm__ volatile("" ::: "memory");             // bias / weight fragments are re-read from LDS per group, not hoisted
            init_acc();
            load_block(g, 0, xa);
            if constexpr (PP) {
                for (int kb = 0; kb < nkb; kb += 2) {
                    if (kb + 1 < nkb) load_block(g, kb + 1, xb);
                    mma_block(kb, xa);
                    if (kb + 1 >= nkb) break;
                    if (kb + 2 < nkb) load_block(g, kb + 2, xa);
                    mma_block(kb + 1, xb);
                }
            } else {
                for (int kb = 0; kb < nkb; ++kb) {
                    mma_block(kb, xa);
                    if (kb + 1 < nkb) load_block(g, kb + 1, xa);
                }
            }
            store_group(g);
        }
    }
}

int conv1x1_pw_rows(const ConvArgs& a) {
    const int cin = a.C0 + a.C1;
    // 128-row tiles when the weight image fits (rows x (2 Cin + 16) bytes <= ~136 KB) and Cout allows, else 64
    if (a.Cout % 128 == 0 && (size_t)128 * (cin * 2 + 16) + 128 * 4 <= 140 * 1024) return 128;
    return 64;
}

bool conv1x1_pw_eligible(int mode, const ConvArgs& a) {
    if (mode != MODE_BF16 || a.kind != 0 || a.kh != 1 || a.kw != 1 || a.stride != 1 || a.pad != 0 || a.pro || a.out_stats) return false;
    if (!a.x0_bf16 || (a.C1 && !a.x1_bf16)) return false;
    const bool out32 = !a.y_bf16;                          // fp32 y (+ fp32 res): the dx projections of the attention / SLA backward
    if (a.res && (a.res_bf16 != 0) == out32) return false;
    const int cin = a.C0 + a.C1;
    if ((cin % 128 && cin != 64) || cin > 1024 || a.Cout % 64 || a.Cout < 64) return false;
    if (out32 && cin == 64) return false;                   // (no such launch)
    if (!out32 && cin != 64 && a.Cout < 128) return false;  // (64-channel bf16 outputs of the forward stay where they were measured)
    if (a.C1 && (a.C0 % 128 || a.C1 % 128)) return false;
    if (a.CinPad != cin) return false;
    const size_t npix = (size_t)a.NF * a.H * a.W;
    if (npix % 32 || npix < 32 * 256) return false;       // whole 32-pixel groups, enough of them for a chip of persistent workgroups
    if (npix * (size_t)std::max(std::max(a.C0, a.C1), a.Cout) * 2 >= 0xFFFF0000ull) return false;
    const int rows = conv1x1_pw_rows(a);
    if ((size_t)rows * (cin * 2 + 16) + rows * 4 > 160 * 1024) return false;
    return true;
}

hipError_t launch_conv1x1_pw(const ConvArgs& a, hipStream_t st) {
    const int cin = a.C0 + a.C1, rows = conv1x1_pw_rows(a), nct = a.Cout / rows;
    const int ngroups = (int)((size_t)a.NF * a.H * a.W / 32);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    // ranges: a multiple of 8 (XCD decode), ~ cus / nct of them, each a multiple of 8 groups (one per wave and pass)
    const size_t lds = (size_t)rows * (cin * 2 + 16) + rows * 4;
    // two workgroups per CU when the weight image allows it (<= 80 KB, and the 128-row form needs < 128 registers): the kernel waits on
    // memory for most of a wave's life (79 % parked with one workgroup per CU, rocprofv3 PMC), so more waves = more loads in flight
    const int wg_per_cu = (lds <= 80 * 1024 && rows == 128) ? 2 : 1;
    int nranges = std::max(8, (cus * wg_per_cu / nct) / 8 * 8);
    int gpr = ((ngroups + nranges - 1) / nranges + 7) / 8 * 8;
    nranges = ((ngroups + gpr - 1) / gpr + 7) / 8 * 8;
    auto go = [&](auto kfn) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kfn, dim3(nranges * nct), dim3(512), lds, st, a, nct, gpr, ngroups);
        return hipGetLastError();
    };
    if (!a.y_bf16) return rows == 128 ? go(conv1x1_pw_kernel<128, 4, true>) : go(conv1x1_pw_kernel<64, 4, true>);
    if (cin == 64) return rows == 128 ? go(conv1x1_pw_kernel<128, 2>) : go(conv1x1_pw_kernel<64, 2>);
    return rows == 128 ? go(conv1x1_pw_kernel<128>) : go(conv1x1_pw_kernel<64>);
}

}  // namespace vdx
