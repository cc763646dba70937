// Implicit-GEMM (1,kh,kw) convolution over channel-last video tensors on MFMA (gfx950).
//
// Replaces, for the new path, what XLA did for the reference's nnx.Conv / nnx.ConvTranspose calls:
//   Block.proj (1,3,3) ............ /root/reference/modules.py:162-165,172
//   Downsample (1,4,4)/s2 ......... /root/reference/utils.py:115-125
//   Upsample ConvTranspose ........ /root/reference/utils.py:103-113 (4 output phases x 2x2 taps)
//   1x1 projections ............... /root/reference/modules.py:71-91,219-222
// and fuses the neighbouring elementwise work of Block (modules.py:171-179):
//   prologue  = GroupNorm-apply * (scale+1) + shift -> SiLU on the INPUT as it is staged to LDS
//   epilogue  = +bias, GroupNorm partial statistics (sum, sum^2 per (sample, group)) of the OUTPUT
// so a Block's activation makes one HBM round trip.
//
// GEMM view: D[cout, pixel] = sum_{tap, cin} Wp[tap][cout][cin] * X[pixel + tap][cin].
// A 256-thread workgroup owns 128 output pixels (NP patches of PH x PW of consecutive frames) x BC
// output channels.  Per K tile (128 B of cin per pixel) the (F,H,W) halo tile is staged ONCE in LDS
// (the 9 taps re-read it with shifted per-lane addresses); the weight tile of each tap streams through
// a 2-deep register->LDS ring.  Accumulators: lane (r,q) holds channels 4q..4q+3 of pixel r (vdx_common.h).
#include "vdx_common.h"
#include <type_traits>
#include "vdx_internal.h"
#include "vdx_glds.h"
#include "model.h"
#include <stdlib.h>
#include <string.h>

namespace vdx {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned OOB = 0xFFFFFFF0u;      // buffer-load offset beyond num_records: the hardware returns zeros

// TN = 16-pixel tiles per wave (4: 64 pixels, 2: 32 pixels), NW = waves.  Workgroup tile = BC channels x BM pixels.
// INF = storage of the input tensors, fixed at compile time so that the staging loop is a straight batch of loads with no
// format branch between them: 0 = fp32 (x0 and x1), 1 = bf16 x0 without concat, 2 = every input bf16 with channel counts that
// are multiples of 8 (16-byte pieces), 3 = anything else (formats read from the descriptor at run time).
template <int MODE, int BC, int TN, int NW, int INF>
__global__ __launch_bounds__(64 * NW, (NW == 8 && TN == 2) ? 4 : 2) void conv_igemm_kernel(const ConvArgs P) {
    using M = Mma<MODE>;
    constexpr int KT = M::KT;
    constexpr int RS = ROW_STRIDE;
    constexpr int WAVES_C = BC / 64;
    constexpr int WAVES_P = NW / WAVES_C;
    constexpr int NT = 64 * NW;                     // threads per workgroup
    constexpr int TM = 4;
    constexpr int APIECES = KT / 4;                 // float4 pieces per staged pixel row
    constexpr int WREGS = BC * 8 / NT;              // 16-byte weight pieces per thread per tap = LDS-DMA instructions per wave per slab
    constexpr int NSW = 3;                          // weight ring: slab t (MFMAs), t + 1 (in flight or landed), t + 2 (being issued)
    constexpr int SLAB = BC * 128;                  // one (tap, K tile) weight slab: BC rows x 128 bytes, 16-byte chunk c of row r at c ^ (r & 7)
    constexpr int AU = 4;                           // halo-tile loads in flight per thread (one or two batches per tile)

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lp = lane & 15, q = lane >> 4;
    const int wc = wave % WAVES_C, wpx = wave / WAVES_C;

    // ---- block decode (all wave-uniform) --------------------------------------------------------
    int bx = blockIdx.x;
    const int tx = bx % P.tiles_x; bx /= P.tiles_x;
    const int ty = bx % P.tiles_y;
    const int fg = bx / P.tiles_y;
    const int f0 = fg * P.NP;
    const int b = f0 / P.F;                          // NP divides F: one sample per workgroup
    const int c0 = blockIdx.y * BC;
    const int ry = P.kind ? (blockIdx.z >> 1) : 0, rx = P.kind ? (blockIdx.z & 1) : 0;
    const int KH = P.kind ? 2 : P.kh, KW = P.kind ? 2 : P.kw;
    const int ntaps = KH * KW;
    const int IH = (P.PH - 1) * P.stride + KH, IW = (P.PW - 1) * P.stride + KW;
    const int oy0 = ty * P.PH, ox0 = tx * P.PW;
    const int iy0 = oy0 * P.stride + (P.kind ? ry - 1 : -P.pad);
    const int ix0 = ox0 * P.stride + (P.kind ? rx - 1 : -P.pad);
    const int HPX = P.NP * IH * IW;
    const int Cin = P.C0 + P.C1;
    const int nchunks = P.CinPad / KT;

    // ---- LDS carve (every offset a multiple of 16 bytes) ------------------------------------------
    size_t off = 0;
    // fp32 partial sums meet in f64: adding a few dozen floats in double is exact, so neither these LDS atomics nor the f64 global
    // atomics of the slab depend on the order the waves arrive in (bf16 sampling is bit-reproducible run to run)
    double* chs = reinterpret_cast<double*>(smem + off); off += 2 * BC * 8;        // [2][BC] channel sum / sumsq
    int* hp_pix = reinterpret_cast<int*>(smem + off); off += ((HPX * 4 + 15) / 16) * 16;   // [HPX] global pixel or -1
    float* coefA = nullptr; float* coefD = nullptr; float* gmean = nullptr;
    if (P.pro) {
        coefA = reinterpret_cast<float*>(smem + off); off += (size_t)P.CinPad * 4;
        coefD = reinterpret_cast<float*>(smem + off); off += (size_t)P.CinPad * 4;
        gmean = reinterpret_cast<float*>(smem + off); off += 64 * 4;               // [groups<=32][mean, rstd]
    }
    char* As = smem + off; off += (size_t)HPX * RS;
    char* Ws = smem + off;                                                          // [NSW][BC rows][128 B] (swizzled, filled by LDS-DMA)

    for (int i = tid; i < 2 * BC; i += NT) chs[i] = 0.0;
    for (int hp = tid; hp < HPX; hp += NT) {
        const int patch = div_magic(hp, P.m_ihiw);
        const int r = hp - patch * (IH * IW);
        const int iy = div_magic(r, P.m_iw), ix = r - iy * IW;
        const int gy = iy0 + iy, gx = ix0 + ix, f = f0 + patch;
        hp_pix[hp] = (f < P.NF && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W) ? ((f * P.H + gy) * P.W + gx) : -1;
    }
    if (P.pro) {
        // per-channel affine of GroupNorm-apply (+ time scale/shift):  x_hat = x * a + d
        gn_mean_rstd_wg(P.in_stats, b, P.groups, (double)P.F * P.H * P.W * (Cin / P.groups), gmean, tid, NT);
        __syncthreads();
        for (int c = tid; c < P.CinPad; c += NT) {
            float a = 0.f, d = 0.f;
            if (c < Cin) {
                const int g = c / (Cin / P.groups);
                const float m = gmean[2 * g], rs = gmean[2 * g + 1];
                const float ga = P.gamma[c], be = P.beta[c];
                float sc = 1.f, sh = 0.f;
                if (P.ss) { sc = P.ss[(size_t)b * P.ss_stride + c] + 1.f; sh = P.ss[(size_t)b * P.ss_stride + Cin + c]; }
                a = rs * ga * sc;
                d = (be - m * rs * ga) * sc + sh;
            }
            coefA[c] = a; coefD[c] = d;
        }
    }
    __syncthreads();

    // ---- buffer descriptors: 32-bit offsets, out-of-range reads return 0 (no bounds branches) ------
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x0), 0, P.x0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x1 ? P.x1 : P.x0), 0, P.x1 ? P.x1_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.wp), 0, P.w_bytes, 0x00020000);

    // ---- per-lane activation-fragment offsets ----------------------------------------------------
    int pixoff[TN];
    int gout[TN];                                    // output pixel index (into y, channel-last) or -1
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int p = wpx * (TN * 16) + tn * 16 + lp;
        const int patch = div_magic(p, P.m_phpw);
        const int r = p - patch * (P.PH * P.PW);
        const int py = div_magic(r, P.m_pw), px = r - py * P.PW;
        pixoff[tn] = ((patch * IH + py * P.stride) * IW + px * P.stride) * RS + q * 16;
        const int oy = oy0 + py, ox = ox0 + px, f = f0 + patch;
        const bool ok = (f < P.NF) && (oy < P.Ho) && (ox < P.Wo);
        const int yy = P.kind ? (2 * oy + ry) : oy, xx = P.kind ? (2 * ox + rx) : ox;
        gout[tn] = ok ? ((f * P.Hy + yy) * P.Wy + xx) : -1;
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- weight stream: the [tap][BC rows][KT] slabs of the packing flow through a 3-deep LDS ring filled by LDS-DMA
    //      (global_load_lds_dwordx4, no VGPRs, no ds_write pass), two taps ahead of the MFMAs; ONE raw s_barrier per tap behind a
    //      counted s_waitcnt vmcnt (conv_ws.hip, round 2; round 1: global -> registers -> LDS one tap ahead + __syncthreads, whose
    //      vmcnt(0) exposed an L2 round trip per tap: 0.5 us per tap on the 64-256-workgroup grids of training) -----------------
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned wsrc[WREGS];                            // this lane's source byte offsets inside a slab's rows (swizzle on the SOURCE side)
#pragma unroll
    for (int k = 0; k < WREGS; ++k) {
        const int i = (k * NW + wave) * 64 + lane, row = i >> 3, pos = i & 7;
        const int co = min(c0 + row, P.Cout - 1);     // rows past Cout re-read the last row; their outputs are never stored
        wsrc[k] = (unsigned)((co + P.wrow0) * P.CinPad) * M::ES + (unsigned)((pos ^ (row & 7)) << 4);
    }
    const size_t tap_stride = (size_t)P.wrows * P.CinPad * M::ES;
    const char* wbase = reinterpret_cast<const char*>(P.wp);
    const unsigned ws_a = lds_addr(Ws);
    const int cc_lo = 0, cc_hi = nchunks;
    int pdy = 0, pdx = 0, pcc = cc_lo, pslot = 0;     // prefetch cursor; wraps at the end (two harmless slabs past the last tap keep the wait counts uniform)
    auto issue_w = [&]() {
        const int widx = P.kind ? ((2 * pdy + ry) * 4 + (2 * pdx + rx)) : (pdy * KW + pdx);
        const char* src = wbase + (size_t)widx * tap_stride + (size_t)(pcc * KT) * M::ES;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ws_a + pslot * SLAB + wave_u * 1024);
#pragma unroll
        for (int k = 0; k < WREGS; ++k) glds16(src + wsrc[k], dst + k * NW * 1024);
        if (++pdx == KW) { pdx = 0; if (++pdy == KH) { pdy = 0; if (++pcc == cc_hi) pcc = cc_lo; } }
        if (++pslot == NSW) pslot = 0;
    };
    // A fragments: row wc*64 + tm*16 + lp of the slab, chunk 4 ch + q at position chunk ^ (row & 7); (row & 7) = (lp & 7)
    const int aoff = (wc * 64 + lp) * 128 + ((q ^ (lp & 7)) << 4);

    // ---- main loop: K tiles of cin x taps --------------------------------------------------------
    const int total = HPX * APIECES;
    constexpr bool RT = (INF == 3);
    const bool x0b = RT ? (P.x0_bf16 != 0) : (INF == 1);
    const bool x1b = RT ? (P.x1_bf16 != 0) : false;
    const bool cat = (INF == 1) ? false : (P.C1 != 0);
    int cslot = 0;
    issue_w(); issue_w();
    for (int cc = cc_lo; cc < cc_hi; ++cc) {
        if (cc != cc_lo) __builtin_amdgcn_s_barrier();   // every wave is done reading the previous halo tile (its fragment reads are consumed: MFMAs issued)
        if constexpr (MODE == MODE_BF16 && INF == 2) {
            // every input tensor is bf16: 16-byte pieces of 8 channels, half the loads / address math / LDS writes, and a
            // plain copy into the bf16 tile when there is no prologue
            constexpr int BP = KT / 8;
            const int total8 = HPX * BP;
            for (int i0 = tid; i0 < total8; i0 += NT * AU) {
                // loads only in this loop (nothing consumes a loaded register before every load of the batch is issued: a use here
                // would put an s_waitcnt vmcnt(0) after each load and serialise the L2 round trips)
                u32x4 v[AU], v1[AU];
#pragma unroll
                for (int u = 0; u < AU; ++u) {
                    const int i = i0 + NT * u;
                    const int hp = i / BP, pc = i % BP;
                    const int c = cc * KT + pc * 8;
                    const int pix = (i < total8) ? hp_pix[hp] : -1;
                    const unsigned o0 = (pix >= 0 && c < P.C0) ? (unsigned)(pix * P.C0 + c) * 2u : OOB;
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs0, o0, 0, 0);
                }
                if (P.C1) {                            // (one uniform branch around the whole second batch, none between loads)
#pragma unroll
                    for (int u = 0; u < AU; ++u) {
                        const int i = i0 + NT * u;
                        const int hp = i / BP, pc = i % BP;
                        const int c = cc * KT + pc * 8;
                        const int pix = (i < total8) ? hp_pix[hp] : -1;
                        const unsigned o1 = (pix >= 0 && c >= P.C0 && c < Cin) ? (unsigned)(pix * P.C1 + (c - P.C0)) * 2u : OOB;
                        v1[u] = __builtin_amdgcn_raw_buffer_load_b128(rs1, o1, 0, 0);
                    }
                }
#pragma unroll
                for (int u = 0; u < AU; ++u) {
                    const int i = i0 + NT * u;
                    if (P.C1) v[u] |= v1[u];
                    if (i < total8) {
                        const int hp = i / BP, pc = i % BP;
                        if (P.pro) {
                            const int c = cc * KT + pc * 8;
                            const bool ok = hp_pix[hp] >= 0 && c < Cin;       // zero padding stays zero AFTER the activation
                            const unsigned w4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                            unsigned o4[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float2 a = *reinterpret_cast<const float2*>(coefA + c + 2 * k);
                                const float2 d = *reinterpret_cast<const float2*>(coefD + c + 2 * k);
                                const float lo = silu_f(fmaf(__uint_as_float(w4[k] << 16), a.x, d.x));
                                const float hi = silu_f(fmaf(__uint_as_float(w4[k] & 0xFFFF0000u), a.y, d.y));
                                o4[k] = ok ? pack_bf16x2(lo, hi) : 0u;
                            }
                            v[u] = u32x4{o4[0], o4[1], o4[2], o4[3]};
                        }
                        *reinterpret_cast<u32x4*>(As + hp * RS + pc * 16) = v[u];
                    }
                }
            }
        } else {
        for (int i0 = tid; i0 < total; i0 += NT * AU) {
            u32x4 v[AU], v1[AU];                       // raw loads first (see above), widened / merged when stored
            u32x2 h[AU], h1[AU];                       // (bf16 sources: 8 bytes; copying them into v here would already be a use)
#pragma unroll
            for (int u = 0; u < AU; ++u) {
                const int i = i0 + NT * u;
                const int hp = i / APIECES, pc = i % APIECES;
                const int c = cc * KT + pc * 4;
                const int pix = (i < total) ? hp_pix[hp] : -1;
                if (x0b) {                              // bf16-stored input (intra-ResnetBlock tensor): 4 channels = 8 bytes
                    const unsigned o0 = (pix >= 0 && c < P.C0) ? (unsigned)(pix * P.C0 + c) * 2u : OOB;
                    h[u] = __builtin_amdgcn_raw_buffer_load_b64(rs0, o0, 0, 0);
                } else {
                    const unsigned o0 = (pix >= 0 && c < P.C0) ? (unsigned)(pix * P.C0 + c) * 4u : OOB;
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs0, o0, 0, 0);
                }
            }
            if (cat) {
#pragma unroll
                for (int u = 0; u < AU; ++u) {
                    const int i = i0 + NT * u;
                    const int hp = i / APIECES, pc = i % APIECES;
                    const int c = cc * KT + pc * 4;
                    const int pix = (i < total) ? hp_pix[hp] : -1;
                    if (x1b) {
                        const unsigned o1 = (pix >= 0 && c >= P.C0 && c < Cin) ? (unsigned)(pix * P.C1 + (c - P.C0)) * 2u : OOB;
                        h1[u] = __builtin_amdgcn_raw_buffer_load_b64(rs1, o1, 0, 0);
                    } else {
                        const unsigned o1 = (pix >= 0 && c >= P.C0 && c < Cin) ? (unsigned)(pix * P.C1 + (c - P.C0)) * 4u : OOB;
                        v1[u] = __builtin_amdgcn_raw_buffer_load_b128(rs1, o1, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < AU; ++u) {
                const int i = i0 + NT * u;
                if (x0b) v[u] = u32x4{h[u].x << 16, h[u].x & 0xFFFF0000u, h[u].y << 16, h[u].y & 0xFFFF0000u};
                if (cat) {
                    if (x1b) v[u] |= u32x4{h1[u].x << 16, h1[u].x & 0xFFFF0000u, h1[u].y << 16, h1[u].y & 0xFFFF0000u};
                    else v[u] |= v1[u];
                }
                if (i < total) {
                    const int hp = i / APIECES, pc = i % APIECES;
                    float4 f = make_float4(__uint_as_float(v[u].x), __uint_as_float(v[u].y), __uint_as_float(v[u].z), __uint_as_float(v[u].w));
                    if (P.pro) {
                        const int c = cc * KT + pc * 4;
                        const bool ok = hp_pix[hp] >= 0 && c < Cin;       // zero padding stays zero AFTER the activation
                        const float4 a = *reinterpret_cast<const float4*>(coefA + c);
                        const float4 d = *reinterpret_cast<const float4*>(coefD + c);
                        f.x = ok ? silu_f(fmaf(f.x, a.x, d.x)) : 0.f; f.y = ok ? silu_f(fmaf(f.y, a.y, d.y)) : 0.f;
                        f.z = ok ? silu_f(fmaf(f.z, a.z, d.z)) : 0.f; f.w = ok ? silu_f(fmaf(f.w, a.w, d.w)) : 0.f;
                    }
                    M::store4(As + hp * RS, pc * 4, f);
                }
            }
        }
        }
        __syncthreads();                                  // the halo tile is complete (LDS stores of every wave)
        int dy = 0, dx = 0;
        for (int tap = 0; tap < ntaps; ++tap) {
            // this tap's slab has landed in every wave's part (only the next slab's WREGS LDS-DMAs may still be in flight), and every
            // wave is done with the slab of the previous tap, whose ring slot the new issue overwrites
            wait_vm<WREGS>();
            __builtin_amdgcn_s_barrier();
            issue_w();
            const int tapoff = (dy * IW + dx) * RS;
            const char* wt = Ws + cslot * SLAB;
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                uint4 bf[TN], af[TM];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bf[tn] = *reinterpret_cast<const uint4*>(As + pixoff[tn] + tapoff + ch * CHUNK_BYTES);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) af[tm] = *reinterpret_cast<const uint4*>(wt + ((aoff + tm * 2048) ^ (ch * 64)));
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) M::mma(acc[tm][tn], af[tm], bf[tn]);
            }
            if (++dx == KW) { dx = 0; ++dy; }
            if (++cslot == NSW) cslot = 0;
        }
    }
    wait_vm_lgkm0<0>();                                   // the two slabs issued past the end must land before the LDS is released

    // ---- epilogue: bias, store, GroupNorm partial statistics --------------------------------------
    const int st_cpg = P.out_stats ? P.Cout / P.out_groups : 4;
    const int st_glo = c0 / st_cpg;
    const bool grp4 = (st_cpg & 3) == 0;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int co = c0 + wc * 64 + tm * 16 + q * 4;
        const bool cvalid = co < P.Cout;
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cvalid && P.bias) bias = *reinterpret_cast<const float4*>(P.bias + co);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f), ss = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            float4 v;
            v.x = acc[tm][tn][0] + bias.x; v.y = acc[tm][tn][1] + bias.y;
            v.z = acc[tm][tn][2] + bias.z; v.w = acc[tm][tn][3] + bias.w;
            if (cvalid && gout[tn] >= 0) {
                if (P.res) {
                    const float4 r4 = load4_f32_or_bf16(P.res, (size_t)gout[tn] * P.Cout + co, P.res_bf16);
                    v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                }
                if (P.y_bf16) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(P.y) + ((size_t)gout[tn] * P.Cout + co) * 2) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                else *reinterpret_cast<float4*>(P.y + (size_t)gout[tn] * P.Cout + co) = v;
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                ss.x += v.x * v.x; ss.y += v.y * v.y; ss.z += v.z * v.z; ss.w += v.w * v.w;
            }
        }
        if (P.out_stats && grp4) {
            // a lane's 4 channels belong to one group (channels per group % 4 == 0): sum them in the lane first, then ONE pair of
            // 16-lane reductions and LDS atomics per channel tile instead of four, straight into the group's slot
            const float s4 = reduce16((s.x + s.y) + (s.z + s.w)), q4 = reduce16((ss.x + ss.y) + (ss.z + ss.w));
            if (lp == 0 && cvalid) {
                const int gl = co / st_cpg - st_glo;
                unsafeAtomicAdd(&chs[gl], (double)s4); unsafeAtomicAdd(&chs[BC + gl], (double)q4);
            }
        } else if (P.out_stats) {
            s.x = reduce16(s.x); s.y = reduce16(s.y); s.z = reduce16(s.z); s.w = reduce16(s.w);
            ss.x = reduce16(ss.x); ss.y = reduce16(ss.y); ss.z = reduce16(ss.z); ss.w = reduce16(ss.w);
            if (lp == 0) {
                const int lc = wc * 64 + tm * 16 + q * 4;
                unsafeAtomicAdd(&chs[lc + 0], (double)s.x); unsafeAtomicAdd(&chs[lc + 1], (double)s.y);
                unsafeAtomicAdd(&chs[lc + 2], (double)s.z); unsafeAtomicAdd(&chs[lc + 3], (double)s.w);
                unsafeAtomicAdd(&chs[BC + lc + 0], (double)ss.x); unsafeAtomicAdd(&chs[BC + lc + 1], (double)ss.y);
                unsafeAtomicAdd(&chs[BC + lc + 2], (double)ss.z); unsafeAtomicAdd(&chs[BC + lc + 3], (double)ss.w);
            }
        }
    }
    if (P.out_stats) {
        __syncthreads();
        const int cpg = P.Cout / P.out_groups;               // channels per group
        const int g_lo = c0 / cpg;
        const int c_hi = min(c0 + BC, P.Cout);
        const int g_hi = (c_hi + cpg - 1) / cpg;              // groups touched by this channel tile
        const int ng = g_hi - g_lo;
        if (tid < 2 * ng) {
            const int g = g_lo + (tid >> 1), which = tid & 1;
            const int lo = max(g * cpg, c0), hi = min((g + 1) * cpg, c_hi);
            double t = 0.0;
            if (grp4) t = chs[which * BC + (tid >> 1)];
            else for (int c = lo; c < hi; ++c) t += chs[which * BC + (c - c0)];
            const int slot = (blockIdx.x + blockIdx.z * 7) % GN_SLOTS;
            unsafeAtomicAdd(P.out_stats + (((size_t)b * GN_SLOTS + slot) * P.out_groups + g) * 2 + which, t);
        }
    }
}

// ---- persistent specialisation: 3x3 / stride 1, Cin = Cout = 64, bf16 mode (the 8 level-0 convs of the N shape) -----------
// One 512-thread workgroup per CU walks a contiguous range of 16x16-pixel tiles.  The whole weight set (9 taps x 64 x 64 bf16
// = 72 KB) is loaded into LDS once and stays there: no per-tap weight staging and no barrier inside the 144-MFMA tap loop.
// The halo tile is double buffered: the next tile's global loads are issued before the MFMAs of the current one and written
// (with the GroupNorm/SiLU prologue applied) after its epilogue; one barrier per tile.  GroupNorm partial sums of the output stay
// in registers across tiles and are flushed when the sample changes.  LDS rows are 128 bytes, unpadded, with the 16-byte chunk
// index XOR-ed by (row & 7): conflict-free for the ds_read_b128 lane groups of 16 consecutive rows (MI355X_MICROARCH.md, LDS).
#ifndef VDX_C32_PRO_WAVES
#define VDX_C32_PRO_WAVES 2     // waves per SIMD the C = 32 prologue form is compiled for: 2 = one workgroup per CU, no scratch (Y-shape step 9.66 ms); 4 = two per CU with 156 bytes of scratch (10.03 ms)
#endif
#ifndef VDX_C64P_EARLY
#define VDX_C64P_EARLY 0      // 1: issue the loads of tile t + 2 inside the tap loop of tile t (measured SLOWER: 495 vs 413 us, r03)
#endif
#ifndef VDX_C64P_DIAG
#define VDX_C64P_DIAG 0      // knock-out switches of conv64p_kernel for timing experiments; none in the product build
#endif
constexpr int C64_HALO = 18 * 18;
__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + 16 * (chunk ^ (row & 7)); }
// the same for rows of C bf16 channels: C = 32 -> 64-byte rows of 4 chunks, chunk ^= 2 on rows with bit 2 set (conflict-free for the
// ds_read_b128 lane groups of 16 consecutive rows and for 8-lane ds_write_b128 groups; brute-forced against the LDS banking table)
template <int C> __device__ __forceinline__ int swz_c(int row, int chunk) {
    if constexpr (C == 64) return row * 128 + 16 * (chunk ^ (row & 7));
    else return row * 64 + 16 * (chunk ^ ((row >> 1) & 2));
}

// PRO / OUT16 / RES: prologue present, bf16 output, residual epilogue (plain form only: the data gradient of a ResnetBlock's first
// conv) -- compile-time, so the unused path costs no registers or issue slots
// C = Cin = Cout: 64 (levels 0 / 1 of the N shape) or 32 (level 0 of the YAML-literal config_v2_2: one 32-deep K chunk, two output-channel
// tiles, 64-byte LDS rows -- memory-bound, 36 MFMAs per wave and tile)
template <bool IN16, bool PRO, bool OUT16, bool RES = false, int C = 64>
__global__ __launch_bounds__(512, C == 32 ? (PRO ? VDX_C32_PRO_WAVES : 4) : 2) void conv64p_kernel(const ConvArgs P, const int tiles_per_block, const int total_tiles) {
    using M = Mma<MODE_BF16>;
    constexpr int PCH = IN16 ? 8 : 4;                 // channels per 16-byte global piece
    constexpr int PPR = C / PCH;                      // pieces per pixel
    constexpr int RB = C * 2;                         // LDS row bytes (bf16)
    constexpr int NTM = C / 16, NCH = C / 32;         // output-channel tiles, 32-deep K chunks
    static_assert(C == 64 || (C == 32 && IN16 && !RES), "C = 32: bf16 tensors, no residual epilogue");
    auto swz = [](int row, int chunk) __attribute__((always_inline)) { return swz_c<C>(row, chunk); };
    constexpr int NPIECE = C64_HALO * PPR;
    constexpr int NU = (NPIECE + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Wl = smem;                                  // [9 * C rows][RB]
    char* Al = Wl + 9 * C * RB;                       // [2][324 rows][RB]
    float* coefA = reinterpret_cast<float*>(Al + 2 * C64_HALO * RB);
    float* coefD = coefA + 64;
    float* gmean = coefD + 64;                        // [32][mean, rstd]
    double* chs = reinterpret_cast<double*>(gmean + 64);   // [2][64] channel sum / sumsq of the current sample (f64: order-independent; [which * 64 + c])
    float* biasl = reinterpret_cast<float*>(chs + 128);    // [64] bias: read per tile in the epilogue (16 registers less to hold across the tap loop:
                                                           // the prologue form sat on the 256-register limit and spilled)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int lp = lane & 15, q = lane >> 4;
    const int tiles_x = P.W >> 4, tiles_pf = tiles_x * (P.H >> 4);
    const int t0 = blockIdx.x * tiles_per_block, t1 = min(t0 + tiles_per_block, total_tiles);
    if (t0 >= t1) return;

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x0), 0, P.x0_bytes, 0x00020000);
    // output through a buffer descriptor too: 32-bit offsets instead of a 64-bit address per store (launcher: the tensor is < 4 GB)
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(P.y, 0, (unsigned)P.NF * P.H * P.W * (unsigned)C * (OUT16 ? 2u : 4u), 0x00020000);
    for (int i = tid; i < 9 * C * (RB / 16); i += 512) {     // packed [tap][wrows][CinPad = 64 ci] bf16: 128-byte rows; this conv's C rows start at wrow0
        const int row = i / (RB / 16), c = i % (RB / 16);    // (a slice of a wider packing: the data gradient of one half of a concat input)
        const size_t srow = (size_t)(row / C) * P.wrows + P.wrow0 + (row % C);
        *reinterpret_cast<uint4*>(Wl + swz(row, c)) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(P.wp) + srow * 128 + c * 16);
    }
    if (tid < 128) chs[tid] = 0.0;
    if (tid < C) biasl[tid] = P.bias ? P.bias[tid] : 0.f;

    // per-thread staging pieces (constant over tiles): halo position, channel piece, LDS byte offset
    // (register budget: 256 at 2 waves per SIMD and the prologue forms sit on it -- the halo row / column share one register, the
    // channel piece is the same for every piece of a thread since 512 % PPR == 0)
    static_assert(512 % PPR == 0, "pieces of a thread share their channel piece");
    int pyx[NU], ploff[NU];                           // (halo row << 5) | halo column; row 100 = not a piece: never in range
    const int pch0 = (tid % PPR) * PCH;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int i = tid + 512 * u;
        const int hp = min(i / PPR, C64_HALO - 1), pc = i % PPR;
        pyx[u] = (((i < NPIECE) ? hp / 18 : 100) << 5) | (hp % 18);
        ploff[u] = IN16 ? swz(hp, pc) : (swz(hp, pc >> 1) + 8 * (pc & 1));
    }
    auto decode = [&](int t, int& f, int& ty, int& tx) { f = t / tiles_pf; const int r = t - f * tiles_pf; ty = r / tiles_x; tx = r - ty * tiles_x; };

    u32x4 sreg[NU];
    unsigned okmask = 0;
    auto stage_load = [&](int t) {
        int f, ty, tx; decode(t, f, ty, tx);
        okmask = 0;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int gy = ty * 16 - 1 + (pyx[u] >> 5), gx = tx * 16 - 1 + (pyx[u] & 31);
            const bool ok = gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
            okmask |= ok ? (1u << u) : 0u;
            const unsigned off = ok ? (unsigned)(((f * P.H + gy) * P.W + gx) * C + pch0) * (IN16 ? 2u : 4u) : OOB;
            sreg[u] = __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0);
        }
    };
    auto stage_store = [&](int buf, int u0 = 0, int u1 = 64) {
        char* dst = Al + buf * (C64_HALO * RB);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (u < u0 || u >= u1) continue;
            if (wave_u * 64 + 512 * u >= NPIECE) continue;               // wave-uniform: the last round of pieces (2592 = 5 x 512 + 32) is wave 0's only --
            if ((pyx[u] >> 5) > 17) continue;                            // a per-lane test alone still issues its SiLU block in all 8 waves
            const bool ok = (okmask >> u) & 1u;
            if (IN16) {
                u32x4 v = sreg[u];
                if (PRO && !(VDX_C64P_DIAG & 2)) {
                    float ca[PCH], cd[PCH];                           // (LDS broadcast-free reads: 4 x 16 B per piece; registers spilled, r02)
#pragma unroll
                    for (int k = 0; k < PCH; k += 4) {
                        const float4 a4 = *reinterpret_cast<const float4*>(coefA + pch0 + k), d4 = *reinterpret_cast<const float4*>(coefD + pch0 + k);
                        ca[k] = a4.x; ca[k + 1] = a4.y; ca[k + 2] = a4.z; ca[k + 3] = a4.w; cd[k] = d4.x; cd[k + 1] = d4.y; cd[k + 2] = d4.z; cd[k + 3] = d4.w;
                    }
                    const unsigned w4[4] = {v.x, v.y, v.z, v.w};
                    unsigned o4[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float lo = silu_f(fmaf(__uint_as_float(w4[k] << 16), ca[2 * k], cd[2 * k]));
                        const float hi = silu_f(fmaf(__uint_as_float(w4[k] & 0xFFFF0000u), ca[2 * k + 1], cd[2 * k + 1]));
                        o4[k] = ok ? pack_bf16x2(lo, hi) : 0u;        // zero padding stays zero AFTER the activation
                    }
                    v = u32x4{o4[0], o4[1], o4[2], o4[3]};
                }
                *reinterpret_cast<u32x4*>(dst + ploff[u]) = v;
            } else {
                float4 f = make_float4(__uint_as_float(sreg[u].x), __uint_as_float(sreg[u].y), __uint_as_float(sreg[u].z), __uint_as_float(sreg[u].w));
                if (PRO) {
                    const float4 a4 = *reinterpret_cast<const float4*>(coefA + pch0), d4 = *reinterpret_cast<const float4*>(coefD + pch0);
                    const float ca[4] = {a4.x, a4.y, a4.z, a4.w}, cd[4] = {d4.x, d4.y, d4.z, d4.w};
                    f.x = ok ? silu_f(fmaf(f.x, ca[0], cd[0])) : 0.f; f.y = ok ? silu_f(fmaf(f.y, ca[1], cd[1])) : 0.f;
                    f.z = ok ? silu_f(fmaf(f.z, ca[2], cd[2])) : 0.f; f.w = ok ? silu_f(fmaf(f.w, ca[3], cd[3])) : 0.f;
                }
                *reinterpret_cast<uint2*>(dst + ploff[u]) = make_uint2(pack_bf16x2(f.x, f.y), pack_bf16x2(f.z, f.w));
            }
        }
    };
    // GroupNorm-apply coefficients of sample b (all threads call; ends with a barrier)
    auto make_coef = [&](int b) {
        if (!PRO) return;
        gn_mean_rstd_wg(P.in_stats, b, P.groups, (double)P.F * P.H * P.W * (C / P.groups), gmean, tid, 512);
        __syncthreads();
        int tt = tid;
        asm volatile("" : "+v"(tt));                  // opaque copy: the addresses below are formed here, per call, instead of being held (and spilled) across the tile loop
        if (tt < C) {
            const int g = tt / (C / P.groups);
            const float m = gmean[2 * g], rsd = gmean[2 * g + 1];
            float sc = 1.f, sh = 0.f;
            if (P.ss) { sc = P.ss[(size_t)b * P.ss_stride + tt] + 1.f; sh = P.ss[(size_t)b * P.ss_stride + C + tt]; }
            coefA[tt] = rsd * P.gamma[tt] * sc;
            coefD[tt] = (P.beta[tt] - m * rsd * P.gamma[tt]) * sc + sh;
        }
        __syncthreads();
    };
    // flush the register partial sums of sample b into out_stats (all threads call)
    f32x4 ssum[NTM], ssq[NTM];
#pragma unroll
    for (int i = 0; i < NTM; ++i) { ssum[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ssq[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    auto flush_stats = [&](int b) {
        if (!P.out_stats) return;
#pragma unroll
        for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s1 = reduce16(ssum[tm][e]), s2 = reduce16(ssq[tm][e]);
                if (lp == 0) { unsafeAtomicAdd(&chs[tm * 16 + 4 * q + e], (double)s1); unsafeAtomicAdd(&chs[64 + tm * 16 + 4 * q + e], (double)s2); }
                ssum[tm][e] = 0.f; ssq[tm][e] = 0.f;
            }
        __syncthreads();
        const int cpg = C / P.out_groups;
        int tt = tid;
        asm volatile("" : "+v"(tt));
        if (tt < 2 * P.out_groups) {
            const int g = tt >> 1, which = tt & 1;
            double t = 0.0;
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) t += chs[which * 64 + c];
            unsafeAtomicAdd(P.out_stats + (((size_t)b * GN_SLOTS + (blockIdx.x % GN_SLOTS)) * P.out_groups + g) * 2 + which, t);
        }
        __syncthreads();
        if (tid < 128) chs[tid] = 0.0;
        __syncthreads();
    };

    // fragment addressing: wave owns output rows py = 2 * wave + tn of the tile, pixel px = lp
    int hpb[2];
    hpb[0] = (2 * wave) * 18 + lp; hpb[1] = hpb[0] + 18;

    int fcur, tyc, txc;
    decode(t0, fcur, tyc, txc);
    int bcur = fcur / P.F;
    __syncthreads();                                  // weights + chs visible
    make_coef(bcur);
    stage_load(t0);
    stage_store(0);
    __syncthreads();
    int bcoef = bcur;                                 // sample the prologue coefficients in LDS belong to
    // The global loads of tile t + 1 are issued as soon as the staging registers are free -- right after tile t has been written to LDS,
    // i.e. inside the tap loop of tile t - 1 -- not at the top of tile t: the epilogue, the barrier and ~2 taps more of latency slack (the
    // knock-out timings of round 3: tap loop and memory pipeline did not overlap, 224 us of 420 was the load -> LDS -> store skeleton alone)
    bool preloaded = false;
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        const bool more = t + 1 < t1;
        int fn = fcur, tyn = tyc, txn = txc, bn = bcur;
        if (more) {
            if (!preloaded) stage_load(t + 1);
            preloaded = false;
            decode(t + 1, fn, tyn, txn);
            bn = fn / P.F;
            if (bn != bcoef) { make_coef(bn); bcoef = bn; }          // uniform; nobody reads the coefficients during the MFMAs
        }
        // ---- 9 taps x 2 chunks x (4 x 2) MFMAs, no barrier; the next tile is written to the other buffer half way through,
        //      so its prologue arithmetic and LDS writes sit between MFMAs instead of after them ----
        f32x4 acc[NTM][2];
#pragma unroll
        for (int i = 0; i < NTM; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // residual epilogue (the data gradient of a ResnetBlock's first conv: + the gradient of the skip path), plain form only:
        // fetched before the MFMAs so that the epilogue does not wait for it
        float4 rpre[RES ? 4 : 1][2];
        if constexpr (RES) {
            const int oy0 = tyc * 16 + 2 * wave, ox = txc * 16 + lp;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    rpre[tm][tn] = *reinterpret_cast<const float4*>(P.res + ((size_t)(fcur * P.H + oy0 + tn) * P.W + ox) * 64 + tm * 16 + 4 * q);   // fp32 res only: 8 plain loads, no format branch between them
        }
        const char* At = Al + buf * (C64_HALO * RB);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // the next tile goes to the other buffer between the taps (loading two tiles ahead instead was measured slower: r02).
            // With a prologue the write is VALU work (SiLU): two pieces per tap over three taps, and the two waves of a SIMD (w, w + 4)
            // take DIFFERENT taps, so one wave's VALU block runs under the other's MFMAs instead of both stalling the matrix pipe at once
            if (more) {
                if (!PRO) { if (tap == 5) { stage_store(buf ^ 1); if (VDX_C64P_EARLY && t + 2 < t1) { stage_load(t + 2); preloaded = true; } } }
                else {
                    const int first = wave_u < 4 ? 4 : 6;
                    if (tap >= 4 && tap - first >= 0 && tap - first < 3) {
                        stage_store(buf ^ 1, 2 * (tap - first), tap - first == 2 ? NU : 2 * (tap - first) + 2);
                        if (VDX_C64P_EARLY && tap - first == 2 && t + 2 < t1) { stage_load(t + 2); preloaded = true; }      // (wave-uniform)
                    }
                }
            }
            const int dy = tap / 3, dx = tap % 3;
            int boff[2];
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) { const int hp = hpb[tn] + dy * 18 + dx; boff[tn] = swz(hp, q); }
            const int woff = swz(tap * C + lp, q);
#pragma unroll
            for (int ch = 0; ch < ((VDX_C64P_DIAG & 4) ? 0 : NCH); ++ch) {
                uint4 af[NTM], bf[2];
#pragma unroll
                for (int tm = 0; tm < NTM; ++tm) af[tm] = *reinterpret_cast<const uint4*>(Wl + ((woff + tm * 16 * RB) ^ (ch * 64)));
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) bf[tn] = *reinterpret_cast<const uint4*>(At + (boff[tn] ^ (ch * 64)));
#pragma unroll
                for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) M::mma(acc[tm][tn], af[tm], bf[tn]);
            }
        }
        // ---- epilogue of tile t ----
#if VDX_C64P_DIAG & 1
        if (acc[0][0][0] == 12345.678f && acc[1][1][1] == 3.f && acc[2][0][2] == 1.f && acc[3][1][3] == 7.f)     // diagnostic: no epilogue (never true on real data)
#endif
        {
            const int oy0 = tyc * 16 + 2 * wave, ox = txc * 16 + lp;
            float4 bias4[NTM];
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm) bias4[tm] = *reinterpret_cast<const float4*>(biasl + tm * 16 + 4 * q);
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const unsigned gout = (unsigned)(((fcur * P.H + oy0 + tn) * P.W + ox) * C + 4 * q);
#pragma unroll
                for (int tm = 0; tm < NTM; ++tm) {
                    float4 v = make_float4(acc[tm][tn][0] + bias4[tm].x, acc[tm][tn][1] + bias4[tm].y, acc[tm][tn][2] + bias4[tm].z, acc[tm][tn][3] + bias4[tm].w);
                    if constexpr (RES) { const float4 r4 = rpre[tm][tn]; v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
                    if constexpr (OUT16) __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)}, rsy, (gout + tm * 16) * 2u, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, rsy, (gout + tm * 16) * 4u, 0, 0);
                    ssum[tm][0] += v.x; ssum[tm][1] += v.y; ssum[tm][2] += v.z; ssum[tm][3] += v.w;
                    ssq[tm][0] += v.x * v.x; ssq[tm][1] += v.y * v.y; ssq[tm][2] += v.z * v.z; ssq[tm][3] += v.w * v.w;
                }
            }
        }
        if (more && bn != bcur) { flush_stats(bcur); bcur = bn; }    // uniform: the next tile belongs to another sample
        fcur = fn; tyc = tyn; txc = txn;
        __syncthreads();
    }
    flush_stats(bcur);
}

// ---- the same persistent 64 -> 64 conv for bf16 INPUT tensors, input staged by LDS-DMA ("conv64d") -------------------------------
// conv64p_kernel fetches the next tile into registers (24 VGPRs of staging + per-piece LDS offsets), waits for them half way
// through the tap loop, applies the prologue and writes the tile with ds_write_b128; its prologue form sits on the 256-register limit
// with a 60-byte spill and 39 % of its wave life is spent parked (rocprofv3 PMC, profiles/r02_pmc_step.md).  Here the next tile lands
// in the other LDS buffer by global_load_lds (lane l fetches the chunk that belongs at its swizzled position, out-of-image pieces
// read a zero page: conv128x64p_kernel's scheme) -- no staging registers, no LDS store instructions.  Plain form only (no prologue;
// optional residual epilogue): 254 -> 245 us per launch at B = 64.  The prologue form was built the same way (every thread
// transforming the pieces it issued in place behind counted vmcnt waits, coefficients in registers) and measured SLOWER than the
// register-staged conv64p_kernel (340 vs 323 us: the in-place pass adds a ds_read_b128 per piece and its waits), so it is not kept.
#ifndef VDX_C64D_LATEWAIT
#define VDX_C64D_LATEWAIT 1
#endif
#ifndef VDX_C64D_SWP
#define VDX_C64D_SWP 0
#endif
#ifndef VDX_C64D_DIAG
#define VDX_C64D_DIAG 0      // knock-out switches for timing experiments (tools/mkvariant.sh); the product build has none
#endif
constexpr int C64D_AROWS = 328;                       // 324 halo rows padded to 41 DMA instructions of 8 rows
constexpr int C64D_APL = C64D_AROWS * 128;
__device__ __attribute__((aligned(16))) unsigned g_zero_page_c64d[4];

template <bool OUT16, bool RES>
__global__ __launch_bounds__(512) void conv64d_kernel(const ConvArgs P, const int tiles_per_block, const int total_tiles) {
    using M = Mma<MODE_BF16>;
    constexpr int NDMA = 41, NK = (NDMA + 7) / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Wl = smem;                                  // [9 * 64 rows][128 B]
    char* Al = Wl + 9 * 64 * 128;                     // [2][328 rows][128 B]
    double* chs = reinterpret_cast<double*>(Al + 2 * C64D_APL);   // [2][64] channel sum / sumsq of the current sample (f64: order-independent)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int lp = lane & 15, q = lane >> 4;
    const int tiles_x = P.W >> 4, tiles_pf = tiles_x * (P.H >> 4);
    const int t0 = blockIdx.x * tiles_per_block, t1 = min(t0 + tiles_per_block, total_tiles);
    if (t0 >= t1) return;

    for (int i = tid; i < 9 * 64 * 8; i += 512) {     // packed [tap][wrows][64 ci] bf16: 128-byte rows; this conv's 64 rows start at wrow0
        const int row = i >> 3, c = i & 7;
        const size_t srow = (size_t)(row >> 6) * P.wrows + P.wrow0 + (row & 63);
        *reinterpret_cast<uint4*>(Wl + swz(row, c)) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(P.wp) + srow * 128 + c * 16);
    }
    if (tid < 128) chs[tid] = 0.0;

    // DMA pieces of this lane: instruction u = wave + 8 k covers halo rows 8 u .. 8 u + 7; the lane's row is 8 u + (lane >> 3) and its
    // 16 bytes are global chunk cch = (lane & 7) ^ (lane >> 3) of that row -- the same 8 channels for every piece of the lane
    int pyx[NK];                                      // (halo row << 5) | halo column; row 100 = padding row / no instruction
    const int cch = (lane & 7) ^ (lane >> 3);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int u = wave + 8 * k, hp = u * 8 + (lane >> 3);
        pyx[k] = (((u < NDMA && hp < C64_HALO) ? hp / 18 : 100) << 5) | (hp % 18);
    }
    const char* const zero_page = reinterpret_cast<const char*>(g_zero_page_c64d);
    const char* const xb = reinterpret_cast<const char*>(P.x0);
    const unsigned al_base = lds_addr(Al);
    auto decode = [&](int t, int& f, int& ty, int& tx) { f = t / tiles_pf; const int r = t - f * tiles_pf; ty = r / tiles_x; tx = r - ty * tiles_x; };
    auto dma = [&](int t, int buf) {
        int f, ty, tx; decode(t, f, ty, tx);
        const unsigned dst = al_base + buf * C64D_APL + wave_u * 1024;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if (wave_u + 8 * k >= NDMA) continue;     // (uniform)
            const int gy = ty * 16 - 1 + (pyx[k] >> 5), gx = tx * 16 - 1 + (pyx[k] & 31);
            const bool ok = gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
            const size_t off = (size_t)((f * P.H + gy) * P.W + gx) * 128 + cch * 16;
            glds16(ok ? static_cast<const void*>(xb + off) : static_cast<const void*>(zero_page), dst + k * 8 * 1024);
        }
    };
    f32x4 ssum[4], ssq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ssum[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ssq[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    auto flush_stats = [&](int b) {
        if (!P.out_stats) return;
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s1 = reduce16(ssum[tm][e]), s2 = reduce16(ssq[tm][e]);
                if (lp == 0) { unsafeAtomicAdd(&chs[tm * 16 + 4 * q + e], (double)s1); unsafeAtomicAdd(&chs[64 + tm * 16 + 4 * q + e], (double)s2); }
                ssum[tm][e] = 0.f; ssq[tm][e] = 0.f;
            }
        __syncthreads();
        const int cpg = 64 / P.out_groups;
        if (tid < 2 * P.out_groups) {
            const int g = tid >> 1, which = tid & 1;
            double t = 0.0;
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) t += chs[which * 64 + c];
            unsafeAtomicAdd(P.out_stats + (((size_t)b * GN_SLOTS + (blockIdx.x % GN_SLOTS)) * P.out_groups + g) * 2 + which, t);
        }
        __syncthreads();
        if (tid < 128) chs[tid] = 0.0;
        __syncthreads();
    };

    int hpb[2];
    hpb[0] = (2 * wave) * 18 + lp; hpb[1] = hpb[0] + 18;
    float4 bias4[4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) bias4[tm] = P.bias ? *reinterpret_cast<const float4*>(P.bias + tm * 16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);

    int fcur, tyc, txc;
    decode(t0, fcur, tyc, txc);
    int bcur = fcur / P.F;
    __syncthreads();                                  // weights + chs visible
    dma(t0, 0);
    wait_vm<0>();
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        const bool more = t + 1 < t1;
        int fn = fcur, tyn = tyc, txn = txc, bn = bcur;
        if (more) {
            decode(t + 1, fn, tyn, txn);
            bn = fn / P.F;
#if !(VDX_C64D_DIAG & 2)
            dma(t + 1, buf ^ 1);                      // the other buffer was last read during tile t - 1 (barrier since)
#endif
        }
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        float4 rpre[RES ? 4 : 1][2];
        if constexpr (RES) {                          // (fp32 residual: issued after the DMA, so the waits below cover it)
            const int oy0 = tyc * 16 + 2 * wave, ox = txc * 16 + lp;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    rpre[tm][tn] = *reinterpret_cast<const float4*>(P.res + ((size_t)(fcur * P.H + oy0 + tn) * P.W + ox) * 64 + tm * 16 + 4 * q);
        }
        const char* At = Al + buf * C64D_APL;
#if VDX_C64D_DIAG & 4
        uint4 af_once[4], bf_once[2];
#endif
#if VDX_C64D_SWP
        // software pipeline over the 18 (tap, K chunk) steps: the 6 fragment reads of step s + 1 are issued BEFORE the 8 MFMAs of step s
        // (two register sets), so an MFMA never waits a whole LDS round trip for reads issued right in front of it.  The compiler's own
        // order puts each pair of reads 2-4 MFMAs ahead of its use with an s_waitcnt behind it: MFMA busy 45 % (profiles/r02_pmc_step.md)
        auto frag_load = [&](int st, uint4 (&af)[4], uint4 (&bf)[2]) __attribute__((always_inline)) {
            const int tap = st >> 1, ch = st & 1, dy = tap / 3, dx = tap % 3;
            const int woff = swz(tap * 64 + lp, q);
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) af[tm] = *reinterpret_cast<const uint4*>(Wl + ((woff + tm * 16 * 128) ^ (ch * 64)));
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) bf[tn] = *reinterpret_cast<const uint4*>(At + (swz(hpb[tn] + dy * 18 + dx, q) ^ (ch * 64)));
        };
        uint4 af[2][4], bf[2][2];
        frag_load(0, af[0], bf[0]);
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            if (st + 1 < 18) frag_load(st + 1, af[(st + 1) & 1], bf[(st + 1) & 1]);
#if VDX_C64D_SWP == 1
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) M::mma(acc[tm][tn], af[st & 1][tm], bf[st & 1][tn]);
#if VDX_C64D_SWP == 1
            __builtin_amdgcn_sched_barrier(0);
#else
            if (st + 1 < 18) {                                     // one read behind each of the first six MFMAs of the step
#pragma unroll
                for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            } else __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
#endif
        }
#else
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            int boff[2];
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) { const int hp = hpb[tn] + dy * 18 + dx; boff[tn] = swz(hp, q); }
            const int woff = swz(tap * 64 + lp, q);
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
#if VDX_C64D_DIAG & 4
                uint4 (&af)[4] = af_once; uint4 (&bf)[2] = bf_once;                     // diagnostic: no fragment reads inside the tap loop
                if (tap == 0 && ch == 0) {
#else
                uint4 af[4], bf[2];
                {
#endif
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) af[tm] = *reinterpret_cast<const uint4*>(Wl + ((woff + tm * 16 * 128) ^ (ch * 64)));
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) bf[tn] = *reinterpret_cast<const uint4*>(At + (boff[tn] ^ (ch * 64)));
                }
#if VDX_C64D_DIAG & 8
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) acc[tm][0][0] += __uint_as_float((af[tm].x ^ bf[0].y ^ bf[1].z) & 0x3F800000u);      // diagnostic: reads only, no MFMA
#else
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) M::mma(acc[tm][tn], af[tm], bf[tn]);
#endif
            }
        }
#endif
#if !VDX_C64D_LATEWAIT
        wait_vm<0>();                                 // the next tile has landed (before the stores below: the wait covers the DMA only)
#endif
#if VDX_C64D_DIAG & 1
        if (acc[0][0][0] == 12345.678f && acc[1][1][1] == 3.f && acc[2][0][2] == 1.f && acc[3][1][3] == 7.f)     // diagnostic: no epilogue (never true on real data)
#endif
        {
            const int oy0 = tyc * 16 + 2 * wave, ox = txc * 16 + lp;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const size_t gout = ((size_t)(fcur * P.H + oy0 + tn) * P.W + ox) * 64;
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) {
                    float4 v = make_float4(acc[tm][tn][0] + bias4[tm].x, acc[tm][tn][1] + bias4[tm].y, acc[tm][tn][2] + bias4[tm].z, acc[tm][tn][3] + bias4[tm].w);
                    if constexpr (RES) { const float4 r4 = rpre[tm][tn]; v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
                    store4_f32_or_bf16(P.y, gout + tm * 16 + 4 * q, v, OUT16 ? 1 : 0);
                    ssum[tm][0] += v.x; ssum[tm][1] += v.y; ssum[tm][2] += v.z; ssum[tm][3] += v.w;
                    ssq[tm][0] += v.x * v.x; ssq[tm][1] += v.y * v.y; ssq[tm][2] += v.z * v.z; ssq[tm][3] += v.w * v.w;
                }
            }
        }
#if VDX_C64D_LATEWAIT
        wait_vm<8>();                                 // everything older than this tile's 8 stores -- the next tile's DMA -- has landed: the
                                                      // epilogue's arithmetic and store issue overlap the tail of the DMA instead of following it
#endif
        if (more && bn != bcur) { flush_stats(bcur); bcur = bn; }    // uniform: the next tile belongs to another sample
        fcur = fn; tyc = tyn; txc = txn;
        __syncthreads();                              // next tile landed everywhere; everybody is done reading this one
    }
    flush_stats(bcur);
}

static hipError_t launch_conv64d(const ConvArgs& a, hipStream_t st) {
    const int total = a.NF * (a.H >> 4) * (a.W >> 4);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    const int grid = std::min(total, cus);
    const int tpb = (total + grid - 1) / grid;
    const int nblocks = (total + tpb - 1) / tpb;
    const size_t lds = 9 * 64 * 128 + 2 * (size_t)C64D_APL + 128 * 8;
    auto launch = [&](auto kfn) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kfn, dim3(nblocks), dim3(512), lds, st, a, tpb, total);
        return hipGetLastError();
    };
    if (a.pro || !a.x0_bf16) return hipErrorInvalidValue;
    if (a.res) {
        if (a.res_bf16) return hipErrorInvalidValue;
        return a.y_bf16 ? launch(conv64d_kernel<true, true>) : launch(conv64d_kernel<false, true>);
    }
    return a.y_bf16 ? launch(conv64d_kernel<true, false>) : launch(conv64d_kernel<false, false>);
}

// ---- weights-in-registers form of the persistent 64 -> 64 conv (bf16 tensors, no prologue): "conv64r" ---------------------------------
// conv64d_kernel keeps the 72 KB weight image in LDS beside TWO input tiles: one 41 KB tile in flight per CU against a ~3 us DMA round
// trip in a 4.9 us tile period, and every MFMA step reads 4 weight + 2 pixel fragments from LDS (knock-out timings of round 3: the
// load -> LDS -> store skeleton and the tap loop do not overlap).  Here a wave owns 32 output channels x 64 pixels (4 rows of the 16 x 16
// tile) and keeps ITS weights -- 9 taps x 2 K chunks x 2 channel tiles = 36 A fragments, 144 registers -- for the whole tile walk:
// no weight image in LDS, so THREE input tiles fit (two tiles of LDS-DMA in flight: a tile has two periods to land), and a step reads 4
// pixel fragments for 8 MFMAs (72 reads per wave and tile instead of 108).  Swizzle key of a halo row = its COLUMN, so a fragment address
// is a register per (dx, K chunk) + the wave's row base + an immediate.
#ifndef VDX_C64R_DIAG
#define VDX_C64R_DIAG 0            // timing knock-outs (wrong results): 1 = no MFMAs, 2 = no fragment reads
#endif
template <bool OUT16>
__global__ __launch_bounds__(512) void conv64r_kernel(const ConvArgs P, const int tiles_per_block, const int total_tiles) {
    using M = Mma<MODE_BF16>;
    constexpr int NDMA = 41, NK = (NDMA + 7) / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Al = smem;                                  // [3][328 rows][128 B]
    double* chs = reinterpret_cast<double*>(Al + 3 * C64D_APL);   // [2][64] channel sum / sumsq of the current sample (f64: order-independent)
    float* biasl = reinterpret_cast<float*>(chs + 128);           // [64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave_u & 1, wp = wave_u >> 1;     // output-channel half, pixel quarter (rows 4 wp .. 4 wp + 3)
    const int lp = lane & 15, q = lane >> 4;
    const int tiles_x = P.W >> 4, tiles_pf = tiles_x * (P.H >> 4);
    const int t0 = blockIdx.x * tiles_per_block, t1 = min(t0 + tiles_per_block, total_tiles);
    if (t0 >= t1) return;

    // this wave's weights: packed [tap][wrows][64 ci] bf16, 16 bytes = ci 32 ch + 8 q .. + 7.  Row lp of channel tile tm is output channel
    // wc * 32 + 8 (lp >> 2) + 4 tm + (lp & 3): the accumulator rows 4 q .. 4 q + 3 of the two tiles are then 8 CONSECUTIVE channels of a lane
    // (one 16-byte store per pixel in the bf16 form)
    uint4 wf[9][2][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
                wf[tap][ch][tm] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(P.wp) +
                    ((size_t)tap * P.wrows + P.wrow0 + wc * 32 + 8 * (lp >> 2) + 4 * tm + (lp & 3)) * 128 + ch * 64 + q * 16);
    if (tid < 128) chs[tid] = 0.0;
    if (tid < 64) biasl[tid] = P.bias ? P.bias[tid] : 0.f;

    // DMA pieces of this lane: instruction u = wave + 8 k covers halo rows 8 u .. 8 u + 7; the lane's row is hp = 8 u + (lane >> 3), its LDS
    // position lane & 7 holds source chunk (lane & 7) ^ (hx & 7) with hx = hp % 18 (column key).  The piece geometry is recomputed per
    // tile from an OPAQUE copy of the lane id: held in registers (or hoisted by the compiler) it is what spills next to 144 weight registers.
    const char* const zero_page = reinterpret_cast<const char*>(g_zero_page_c64d);
    const char* const xb = reinterpret_cast<const char*>(P.x0);
    const unsigned al_base = lds_addr(Al);
    auto decode = [&](int t, int& f, int& ty, int& tx) { f = t / tiles_pf; const int r = t - f * tiles_pf; ty = r / tiles_x; tx = r - ty * tiles_x; };
    auto dma = [&](int t, int buf) {
        int f, ty, tx; decode(t, f, ty, tx);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const unsigned dst = al_base + buf * C64D_APL + wave_u * 1024;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if (wave_u + 8 * k >= NDMA) continue;     // (uniform)
            const int hp = (wave_u + 8 * k) * 8 + (ln >> 3);
            const int hy = (hp * 3641) >> 16, hx = hp - hy * 18;          // hp / 18, hp % 18 for hp < 328
            const int gy = ty * 16 - 1 + hy, gx = tx * 16 - 1 + hx;
            const bool ok = hp < C64_HALO && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
            const size_t off = (size_t)((f * P.H + gy) * P.W + gx) * 128 + (((ln & 7) ^ (hx & 7)) << 4);
            glds16(ok ? static_cast<const void*>(xb + off) : static_cast<const void*>(zero_page), dst + k * 8 * 1024);
        }
    };
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(P.y, 0, (unsigned)P.NF * P.H * P.W * 64u * (OUT16 ? 2u : 4u), 0x00020000);
    f32x4 ssum[2], ssq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { ssum[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ssq[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    auto flush_stats = [&](int b) {
        if (!P.out_stats) return;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s1 = reduce16(ssum[tm][e]), s2 = reduce16(ssq[tm][e]);
                const int c = wc * 32 + 8 * q + 4 * tm + e;
                if (lp == 0) { unsafeAtomicAdd(&chs[c], (double)s1); unsafeAtomicAdd(&chs[64 + c], (double)s2); }
                ssum[tm][e] = 0.f; ssq[tm][e] = 0.f;
            }
        __syncthreads();
        const int cpg = 64 / P.out_groups;
        int tt = tid;
        asm volatile("" : "+v"(tt));
        if (tt < 2 * P.out_groups) {
            const int g = tt >> 1, which = tt & 1;
            double t = 0.0;
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) t += chs[which * 64 + c];
            unsafeAtomicAdd(P.out_stats + (((size_t)b * GN_SLOTS + (blockIdx.x % GN_SLOTS)) * P.out_groups + g) * 2 + which, t);
        }
        __syncthreads();
        if (tid < 128) chs[tid] = 0.0;
        __syncthreads();
    };

    // fragment addressing (see the header): B fragment of (tn, dy, dx, ch) = At + rowbase + bdx[dx][ch] + (tn + dy) * 18 * 128
    int bdx[3][2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) { const int hx = lp + dx; bdx[dx][ch] = hx * 128 + 16 * ((q + 4 * ch) ^ (hx & 7)); }
    const int rowbase = (4 * wp) * 18 * 128;           // (wave-uniform)

    int fcur, tyc, txc;
    decode(t0, fcur, tyc, txc);
    int bcur = fcur / P.F;
    __syncthreads();                                  // chs + bias visible
    dma(t0, 0);
    if (t0 + 1 < t1) { dma(t0 + 1, 1); wait_vm<5>(); } else wait_vm<0>();      // tile t0 has landed (a wave issues 5 or 6 instructions per tile)
    __syncthreads();
    int buf = 0;
    for (int t = t0; t < t1; ++t) {
        const bool more = t + 1 < t1, more2 = t + 2 < t1;
        int fn = fcur, tyn = tyc, txn = txc, bn = bcur;
        if (more) { decode(t + 1, fn, tyn, txn); bn = fn / P.F; }
        if (more2) dma(t + 2, buf >= 1 ? buf - 1 : 2);    // (buf + 2) % 3: the buffer of tile t - 1, last read before the previous barrier
        const char* At = Al + buf * C64D_APL + rowbase;
        const int oy0 = tyc * 16 + 4 * wp, ox = txc * 16 + lp;
        // Sliding window over the wave's 6 halo rows: a fragment of halo row hr serves output rows hr, hr - 1, hr - 2 (taps dy = 0, 1, 2: all
        // weights are in registers), so a tile costs 36 fragment reads per wave; at most three rows of accumulators are live, a row is
        // finished (bias, statistics, store) as soon as halo row hr = row + 2 has been consumed.
        f32x4 acc[2][4];
        uint4 bf[2][3];                                  // fragments of step s = 2 hr + ch in bf[s & 1]: step s + 1 is read before step s's MFMAs
        auto frag_read = [&](uint4 (&d)[3], int hr, int ch) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                if constexpr (VDX_C64R_DIAG & 2) d[dx] = uint4{(unsigned)lane, (unsigned)dx, (unsigned)t, (unsigned)ch};
                else d[dx] = *reinterpret_cast<const uint4*>(At + bdx[dx][ch] + hr * (18 * 128));
            }
        };
        frag_read(bf[0], 0, 0);
#pragma unroll
        for (int hr = 0; hr < 6; ++hr) {
            if (hr < 4) { acc[0][hr] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1][hr] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int s_ = 2 * hr + ch;
                if (s_ + 1 < 12) frag_read(bf[(s_ + 1) & 1], (s_ + 1) >> 1, (s_ + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int tn = hr - dy;
                        if (tn < 0 || tn > 3) continue;
#pragma unroll
                        for (int tm = 0; tm < 2; ++tm) {
                            if constexpr (VDX_C64R_DIAG & 1) acc[tm][tn][0] += __uint_as_float(bf[s_ & 1][dx].x ^ wf[dy * 3 + dx][ch][tm].x);
                            else M::mma(acc[tm][tn], wf[dy * 3 + dx][ch][tm], bf[s_ & 1][dx]);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (hr >= 2) {
                const int tn = hr - 2;
                float4 bias4[2];
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) bias4[tm] = *reinterpret_cast<const float4*>(biasl + wc * 32 + 8 * q + 4 * tm);
                const unsigned gout = (unsigned)(((fcur * P.H + oy0 + tn) * P.W + ox) * 64 + wc * 32 + 8 * q);
                float4 v[2];
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    v[tm] = make_float4(acc[tm][tn][0] + bias4[tm].x, acc[tm][tn][1] + bias4[tm].y, acc[tm][tn][2] + bias4[tm].z, acc[tm][tn][3] + bias4[tm].w);
                    ssum[tm][0] += v[tm].x; ssum[tm][1] += v[tm].y; ssum[tm][2] += v[tm].z; ssum[tm][3] += v[tm].w;
                    ssq[tm][0] += v[tm].x * v[tm].x; ssq[tm][1] += v[tm].y * v[tm].y; ssq[tm][2] += v[tm].z * v[tm].z; ssq[tm][3] += v[tm].w * v[tm].w;
                }
                if constexpr (OUT16)
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{pack_bf16x2(v[0].x, v[0].y), pack_bf16x2(v[0].z, v[0].w), pack_bf16x2(v[1].x, v[1].y), pack_bf16x2(v[1].z, v[1].w)}, rsy, gout * 2u, 0, 0);
                else {
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[tm].x), __float_as_uint(v[tm].y), __float_as_uint(v[tm].z), __float_as_uint(v[tm].w)}, rsy, (gout + 4 * tm) * 4u, 0, 0);
                }
            }
        }
        // tile t + 1 has landed: everything older than this iteration's DMA (5 or 6 instructions, when issued) and its 4 or 8 stores
        constexpr int NST = OUT16 ? 4 : 8;
        if (more2) wait_vm<5 + NST>(); else wait_vm<NST>();
        if (more && bn != bcur) { flush_stats(bcur); bcur = bn; }    // uniform: the next tile belongs to another sample
        fcur = fn; tyc = tyn; txc = txn;
        buf = buf == 2 ? 0 : buf + 1;
        __syncthreads();                              // tile t + 1 landed everywhere; everybody is done reading tile t
    }
    flush_stats(bcur);
}

static hipError_t launch_conv64r(const ConvArgs& a, hipStream_t st) {
    const int total = a.NF * (a.H >> 4) * (a.W >> 4);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    const int grid = std::min(total, cus);
    const int tpb = (total + grid - 1) / grid;
    const int nblocks = (total + tpb - 1) / tpb;
    const size_t lds = 3 * (size_t)C64D_APL + 128 * 8 + 64 * 4;
    auto launch = [&](auto kfn) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kfn, dim3(nblocks), dim3(512), lds, st, a, tpb, total);
        return hipGetLastError();
    };
    if (a.pro || !a.x0_bf16 || a.res) return hipErrorInvalidValue;
    return a.y_bf16 ? launch(conv64r_kernel<true>) : launch(conv64r_kernel<false>);
}

// ---- "conv64q": the weights-in-registers scheme with 16 output channels x 128 pixels per wave ------------------------------------------
// conv64r_kernel's waves own 32 channels x 64 pixels: 144 weight registers, which leaves the prologue form no room (256 registers + scratch,
// every coefficient re-read from LDS per piece: 449 us against conv64p_kernel's 422) and does not fit 128 input channels at all.  Here wave
// (wc, wp) owns output channels 16 wc .. 16 wc + 15 of pixel rows 8 wp .. 8 wp + 7: 9 taps x CIN / 32 K chunks = 18 A fragments (72
// registers) for CIN = 64, 36 (144) for CIN = 128.  Same three-deep LDS-DMA ring of 64-channel PLANES (a 128-channel tile = two plane
// passes over persistent accumulators: the two tensors of a concat input, or the two halves of one), same column-keyed swizzle, same
// sliding window: a fragment of halo row hr serves output rows hr, hr - 1, hr - 2; 60 fragment reads for 144 MFMAs per wave and pass.
// Two finished rows are exchanged between the lane quads (v_permlane16_swap) so that a lane stores 8 consecutive channels = 16 bytes.
// PRO (CIN = 64): GroupNorm-apply . (scale + 1) + shift -> SiLU applied IN PLACE to tile t + 1 while the MFMAs of tile t run; every thread
// owns one 8-channel chunk of rows (tid >> 3) + 64 k, so its 16 coefficients stay in registers, and the piece of slot k + 1 is read from
// LDS before the arithmetic of slot k.
#ifndef VDX_C64Q_VPM
#define VDX_C64Q_VPM 6        // VALU instructions of the prologue behind each MFMA of a transform step
#endif
// COUT = 64, or 32 (level 0 / 1 of dim-32 networks: wave (wc, wp) = channels 16 wc.., pixel rows 4 wp..4 wp + 3, 6 halo rows, 12 steps); the 64 input
// channels of the COUT = 32 form are the two 32-channel tensors of a concat input (P.C1 == 32) or one 64-channel tensor
template <int CIN, bool PRO, bool OUT16, int COUT = 64>
__global__ __launch_bounds__(512) void conv64q_kernel(const ConvArgs P, const int tiles_per_block, const int total_tiles) {
    using M = Mma<MODE_BF16>;
    static_assert(CIN == 64 || (CIN == 128 && !PRO), "prologue form: 64 input channels");
    static_assert(COUT == 64 || (COUT == 32 && !PRO), "COUT");
    constexpr int NWC = COUT / 16, NWP = 8 / NWC, RW = 16 / NWP, NHR = RW + 2, NSTEP = 2 * NHR;   // channel quarters, pixel parts, rows per wave, halo rows, steps per pass
    constexpr int NPL = CIN / 64;                     // 64-channel planes per tile
    constexpr int NDMA = 41, NK = (NDMA + 7) / 8;
    constexpr int NST = (OUT16 ? 1 : 2) * (RW / 2);   // row stores of a wave per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Al = smem;                                  // [3][328 rows][128 B]
    double* chs = reinterpret_cast<double*>(Al + 3 * C64D_APL);   // [2][64] channel sum / sumsq of the current sample (f64: order-independent)
    float* biasl = reinterpret_cast<float*>(chs + 128);           // [64]
    float* coefA = biasl + 64;                                    // PRO: [64] x -> silu(x * coefA + coefD)
    float* coefD = coefA + 64;
    float* gmean = coefD + 64;                                    // [32][mean, rstd]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave_u % NWC, wp = wave_u / NWC;   // output-channel quarter, pixel part (rows RW wp .. RW wp + RW - 1)
    const int lp = lane & 15, q = lane >> 4;
    const int tiles_x = P.W >> 4, tiles_pf = tiles_x * (P.H >> 4);
    const int t0 = blockIdx.x * tiles_per_block, t1 = min(t0 + tiles_per_block, total_tiles);
    if (t0 >= t1) return;

    // this wave's weights: packed [tap][wrows][CIN ci] bf16; fragment (tap, K chunk c) of lane (lp, q) = ci 32 c + 8 q .. + 7 of row 16 wc + lp
    uint4 wf[9][CIN / 32];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int c = 0; c < CIN / 32; ++c)
            wf[tap][c] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(P.wp) +
                ((size_t)tap * P.wrows + P.wrow0 + wc * 16 + lp) * (CIN * 2) + c * 64 + q * 16);
    if (tid < 128) chs[tid] = 0.0;
    if (tid < 64) biasl[tid] = (P.bias && tid < COUT) ? P.bias[tid] : 0.f;

    const char* const zero_page = reinterpret_cast<const char*>(g_zero_page_c64d);
    const char* const xb0 = reinterpret_cast<const char*>(P.x0);
    const char* const xb1 = (CIN == 128) ? (P.C1 ? reinterpret_cast<const char*>(P.x1) : xb0 + 128) : xb0;   // plane 1: second tensor, or channels 64..127
    const int rowb = (CIN == 128 && !P.C1) ? 256 : 128;          // bytes per pixel of a plane's tensor
    const bool split32 = CIN == 64 && P.C1 == 32;                // the plane = two 32-channel tensors (64-byte pixels): chunks 0..3 from x0, 4..7 from x1
    const char* const xs1 = split32 ? reinterpret_cast<const char*>(P.x1) : nullptr;
    const unsigned al_base = lds_addr(Al);
    auto decode = [&](int t, int& f, int& ty, int& tx) { f = t / tiles_pf; const int r = t - f * tiles_pf; ty = r / tiles_x; tx = r - ty * tiles_x; };
    // plane pass j = t * NPL + plane lives in ring buffer j % 3
    auto dma = [&](int t, int plane, int buf) {
        int f, ty, tx; decode(t, f, ty, tx);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const char* const xb = plane ? xb1 : xb0;
        const unsigned dst = al_base + buf * C64D_APL + wave_u * 1024;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if (wave_u + 8 * k >= NDMA) continue;     // (uniform)
            const int hp = (wave_u + 8 * k) * 8 + (ln >> 3);
            const int hy = (hp * 3641) >> 16, hx = hp - hy * 18;          // hp / 18, hp % 18 for hp < 328
            const int gy = ty * 16 - 1 + hy, gx = tx * 16 - 1 + hx;
            const bool ok = hp < C64_HALO && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
            const int cch = (ln & 7) ^ (hx & 7);
            const size_t pixi = (size_t)((f * P.H + gy) * P.W + gx);
            const char* src = split32 ? (cch < 4 ? xb0 : xs1) + pixi * 64 + ((cch & 3) << 4) : xb + pixi * rowb + (cch << 4);
            glds16(ok ? static_cast<const void*>(src) : static_cast<const void*>(zero_page), dst + k * 8 * 1024);
        }
    };
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(P.y, 0, (unsigned)P.NF * P.H * P.W * (unsigned)COUT * (OUT16 ? 2u : 4u), 0x00020000);
    // statistics of this lane's 8 channels 16 wc + 8 (q >> 1) + i (rows of both parities)
    float ssum[8], ssq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { ssum[i] = 0.f; ssq[i] = 0.f; }
    auto flush_stats = [&](int b) {
        if (!P.out_stats) return;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float s1 = reduce16(ssum[i]), s2 = reduce16(ssq[i]);
            const int c = wc * 16 + 8 * (q >> 1) + i;
            if (lp == 0) { unsafeAtomicAdd(&chs[c], (double)s1); unsafeAtomicAdd(&chs[64 + c], (double)s2); }
            ssum[i] = 0.f; ssq[i] = 0.f;
        }
        __syncthreads();
        const int cpg = COUT / P.out_groups;
        int tt = tid;
        asm volatile("" : "+v"(tt));
        if (tt < 2 * P.out_groups) {
            const int g = tt >> 1, which = tt & 1;
            double t = 0.0;
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) t += chs[which * 64 + c];
            unsafeAtomicAdd(P.out_stats + (((size_t)b * GN_SLOTS + (blockIdx.x % GN_SLOTS)) * P.out_groups + g) * 2 + which, t);
        }
        __syncthreads();
        if (tid < 128) chs[tid] = 0.0;
        __syncthreads();
    };
    // PRO: this thread's channel chunk and coefficients
    const int tch = tid & 7, trow = tid >> 3;         // chunk 8 tch .. 8 tch + 7 of halo rows trow + 64 k
    float ca[PRO ? 8 : 1], cd[PRO ? 8 : 1];
    auto make_coef = [&](int b) {                     // (all threads call; contains barriers)
        if constexpr (PRO) {
            gn_mean_rstd_wg(P.in_stats, b, P.groups, (double)P.F * P.H * P.W * (64 / P.groups), gmean, tid, 512);
            __syncthreads();
            int tt = tid;
            asm volatile("" : "+v"(tt));
            if (tt < 64) {
                const int g = tt / (64 / P.groups);
                const float m = gmean[2 * g], rsd = gmean[2 * g + 1];
                float sc = 1.f, sh = 0.f;
                if (P.ss) { sc = P.ss[(size_t)b * P.ss_stride + tt] + 1.f; sh = P.ss[(size_t)b * P.ss_stride + 64 + tt]; }
                coefA[tt] = rsd * P.gamma[tt] * sc;
                coefD[tt] = (P.beta[tt] - m * rsd * P.gamma[tt]) * sc + sh;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) { ca[i] = coefA[8 * tch + i]; cd[i] = coefD[8 * tch + i]; }
        }
    };
    // LDS byte offset (inside a plane buffer) of this thread's piece of round k, and whether the piece is inside the image
    auto piece_off = [&](int k) __attribute__((always_inline)) -> int {
        const int hp = trow + 64 * k;
        const int hy = (hp * 3641) >> 16, hx = hp - hy * 18;
        return hp * 128 + ((tch ^ (hx & 7)) << 4);
    };
    auto piece_ok = [&](int k, int ty, int tx) __attribute__((always_inline)) -> bool {
        const int hp = trow + 64 * k;
        const int hy = (hp * 3641) >> 16, hx = hp - hy * 18;
        const int gy = ty * 16 - 1 + hy, gx = tx * 16 - 1 + hx;
        return hp < C64_HALO && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
    };
    auto piece_math = [&](u32x4 v, bool ok) __attribute__((always_inline)) -> u32x4 {
        const unsigned w4[4] = {v.x, v.y, v.z, v.w};
        unsigned o4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = silu_f(fmaf(__uint_as_float(w4[j] << 16), ca[PRO ? 2 * j : 0], cd[PRO ? 2 * j : 0]));
            const float hi = silu_f(fmaf(__uint_as_float(w4[j] & 0xFFFF0000u), ca[PRO ? 2 * j + 1 : 0], cd[PRO ? 2 * j + 1 : 0]));
            o4[j] = ok ? pack_bf16x2(lo, hi) : 0u;    // zero padding stays zero AFTER the activation
        }
        return u32x4{o4[0], o4[1], o4[2], o4[3]};
    };
    constexpr int NPK = 6;                            // rounds of 64 rows: 324 = 5 x 64 + 4 (the last round is wave 0's lanes 0..31 only)
    auto piece_live = [&](int k) __attribute__((always_inline)) -> bool { return k < NPK - 1 || wave_u == 0; };   // (uniform)

    // fragment addressing: B fragment of (halo row hr of this wave, dx, K chunk ch) = At + rowbase + bdx[dx][ch] + hr * 18 * 128
    int bdx[3][2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) { const int hx = lp + dx; bdx[dx][ch] = hx * 128 + 16 * ((q + 4 * ch) ^ (hx & 7)); }
    const int rowbase = (RW * wp) * 18 * 128;         // (wave-uniform)

    int fcur, tyc, txc;
    decode(t0, fcur, tyc, txc);
    int bcur = fcur / P.F;
    __syncthreads();                                  // chs + bias visible
    // passes in flight: j = 0 (and 1) issued here, j + 2 at the top of pass j
    const int npass = (t1 - t0) * NPL;
    dma(t0, 0, 0);
    if (npass > 1) dma(NPL == 1 ? t0 + 1 : t0, NPL == 1 ? 0 : 1, 1);
    // pass 0 has landed (a wave issues 5 or 6 instructions per pass); PRO: pass 1 too -- it is transformed during pass 0, so a pass's DMA
    // has ONE period to land (buffers: computing / being transformed / landing), the plain form's has two
    if (npass > 1 && !PRO) wait_vm<5>(); else wait_vm<0>();
    int bcoef = bcur;
    if constexpr (PRO) {
        make_coef(bcur);                              // (its barriers also make pass 0 visible to every thread)
#pragma unroll
        for (int k = 0; k < NPK; ++k) {
            if (!piece_live(k)) continue;
            char* pp = Al + piece_off(k);
            if (trow + 64 * k < C64D_AROWS) *reinterpret_cast<u32x4*>(pp) = piece_math(*reinterpret_cast<const u32x4*>(pp), piece_ok(k, tyc, txc));
        }
    }
    __syncthreads();
    int buf = 0, j = 0;                               // ring buffer and index of the current pass
    f32x4 acc[RW];
    for (int t = t0; t < t1; ++t) {
        const bool more = t + 1 < t1;
        int fn = fcur, tyn = tyc, txn = txc, bn = bcur;
        if (more) { decode(t + 1, fn, tyn, txn); bn = fn / P.F; }
        const int oy0 = tyc * 16 + RW * wp, ox = txc * 16 + lp;
#pragma unroll
        for (int plane = 0; plane < NPL; ++plane, ++j) {
            if (j + 2 < npass) {                      // (buf + 2) % 3: the buffer of pass j - 1, last read before the previous barrier
                const int j2 = j + 2;
                dma(t0 + j2 / NPL, NPL == 1 ? 0 : (j2 & 1), buf >= 1 ? buf - 1 : 2);
            }
            const int bufn = buf == 2 ? 0 : buf + 1;
            const char* At = Al + buf * C64D_APL + rowbase;
            char* const An = Al + bufn * C64D_APL;    // PRO: the tile being transformed
            // The hr / K-chunk steps of the pass.  XF (PRO, a next tile exists): the prologue of tile t + 1 rides in the MFMA shadows of
            // steps 4..13 (9 MFMAs each): half a piece (4 channels: 32 VALU) per step, laid out MFMA, 4 VALU, MFMA, 4 VALU, ... by
            // sched_group_barrier -- behind the step's MFMAs as a block (first form) the matrix pipe idled while both waves of a SIMD ran
            // their SiLU at the same time, and the prologue form cost its whole VALU time on top of the plain form (410-446 vs 300 us).
            auto steps = [&](auto xf_tag) __attribute__((always_inline)) {
                constexpr bool XF = decltype(xf_tag)::value;
                uint4 bf[2][3];                       // fragments of step s = 2 hr + ch in bf[s & 1]: step s + 1 is read before step s's MFMAs
                auto frag_read = [&](uint4 (&d)[3], int hr, int ch) __attribute__((always_inline)) {
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) d[dx] = *reinterpret_cast<const uint4*>(At + bdx[dx][ch] + hr * (18 * 128));
                };
                auto half_math = [&](unsigned w0, unsigned w1, int cb, bool ok, unsigned& o0, unsigned& o1) __attribute__((always_inline)) {
                    const float a0 = silu_f(fmaf(__uint_as_float(w0 << 16), ca[PRO ? cb : 0], cd[PRO ? cb : 0]));
                    const float a1 = silu_f(fmaf(__uint_as_float(w0 & 0xFFFF0000u), ca[PRO ? cb + 1 : 0], cd[PRO ? cb + 1 : 0]));
                    const float a2 = silu_f(fmaf(__uint_as_float(w1 << 16), ca[PRO ? cb + 2 : 0], cd[PRO ? cb + 2 : 0]));
                    const float a3 = silu_f(fmaf(__uint_as_float(w1 & 0xFFFF0000u), ca[PRO ? cb + 3 : 0], cd[PRO ? cb + 3 : 0]));
                    const unsigned keep = ok ? 0xFFFFFFFFu : 0u;       // zero padding stays zero AFTER the activation (a mask, not a select around
                    o0 = pack_bf16x2(a0, a1) & keep;                   // the arithmetic: the compiler turns that into a branch over it, which
                    o1 = pack_bf16x2(a2, a3) & keep;                   // cuts the MFMA / VALU interleave region in two)
                };
                u32x4 pv = u32x4{0u, 0u, 0u, 0u}, cur = pv;   // XF: the piece of the next slot (read one slot ahead) / of this slot
                bool pok = false, okc = false;
                unsigned oh0 = 0u, oh1 = 0u;
                if constexpr (XF) { pv = *reinterpret_cast<const u32x4*>(An + piece_off(0)); pok = piece_ok(0, tyn, txn); }
                frag_read(bf[0], 0, 0);
#pragma unroll
                for (int hr = 0; hr < NHR; ++hr) {
                    if (plane == 0 && hr < RW) acc[hr] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        const int s_ = 2 * hr + ch;
                        if (s_ + 1 < NSTEP) frag_read(bf[(s_ + 1) & 1], (s_ + 1) >> 1, (s_ + 1) & 1);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                            for (int dy = 0; dy < 3; ++dy) {
                                const int tn = hr - dy;
                                if (tn < 0 || tn > RW - 1) continue;
                                M::mma(acc[tn], wf[dy * 3 + dx][plane * 2 + ch], bf[s_ & 1][dx]);
                            }
                        if constexpr (XF) {
                            const int c = s_ - 4;     // half-piece slot: rounds k = 0..4 of 64 rows
                            if (c >= 0 && c < 10) {
                                const int k = c >> 1;
                                if ((c & 1) == 0) {
                                    cur = pv; okc = pok;
                                    if (k + 1 < NPK - 1) { pv = *reinterpret_cast<const u32x4*>(An + piece_off(k + 1)); pok = piece_ok(k + 1, tyn, txn); }
                                    half_math(cur.x, cur.y, 0, okc, oh0, oh1);
                                } else {
                                    unsigned o2, o3;
                                    half_math(cur.z, cur.w, 4, okc, o2, o3);
                                    *reinterpret_cast<u32x4*>(An + piece_off(k)) = u32x4{oh0, oh1, o2, o3};
                                }
#pragma unroll
                                for (int i = 0; i < 9; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, VDX_C64Q_VPM, 0); }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (XF) {
                            if (s_ == 14 && wave_u == 0 && trow + 64 * (NPK - 1) < C64D_AROWS) {     // the last 4 (+ 4 padding) rows: wave 0's lanes 0..31
                                char* pp = An + piece_off(NPK - 1);
                                *reinterpret_cast<u32x4*>(pp) = piece_math(*reinterpret_cast<const u32x4*>(pp), piece_ok(NPK - 1, tyn, txn));
                            }
                        }
                    }
                    // rows hr - 3 (even) and hr - 2 are finished once halo row hr has been consumed in the LAST plane: exchange the lane quads so
                    // that lane (px, q) holds row (hr - 3) + (q & 1), channels 16 wc + 8 (q >> 1) .. + 7, add the bias, take the statistics, store
                    if (plane == NPL - 1 && hr >= 3 && (hr & 1)) {
                        const int ta = hr - 3;
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[ta][r]), __float_as_uint(acc[ta + 1][r]), false, false);
                            v[r] = __uint_as_float(sw[0]); v[4 + r] = __uint_as_float(sw[1]);
                        }
                        const float4 b0 = *reinterpret_cast<const float4*>(biasl + wc * 16 + 8 * (q >> 1)), b1 = *reinterpret_cast<const float4*>(biasl + wc * 16 + 8 * (q >> 1) + 4);
                        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
#pragma unroll
                        for (int i = 0; i < 8; ++i) { ssum[i] += v[i]; ssq[i] += v[i] * v[i]; }
                        const unsigned gout = (unsigned)(((fcur * P.H + oy0 + ta + (q & 1)) * P.W + ox) * COUT + wc * 16 + 8 * (q >> 1));
                        if constexpr (OUT16)
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])}, rsy, gout * 2u, 0, 0);
                        else {
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, rsy, gout * 4u, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, rsy, (gout + 4) * 4u, 0, 0);
                        }
                    }
                }
            };
            if (PRO && more) steps(std::true_type{}); else steps(std::false_type{});
            // pass j + 1 has landed: everything older than this pass's DMA (5 or 6 instructions, when issued) and the row stores of the last plane
            if constexpr (PRO) wait_vm<NST>();        // (the DMA issued at the top of this pass: the next pass transforms it)
            else {
                const bool issued = j + 2 < npass;
                if (plane == NPL - 1) { if (issued) wait_vm<5 + NST>(); else wait_vm<NST>(); }
                else { if (issued) wait_vm<5>(); else wait_vm<0>(); }
            }
            if (plane == NPL - 1) {
                if (more && bn != bcur) { flush_stats(bcur); bcur = bn; }    // uniform: the next tile belongs to another sample
                if constexpr (PRO) {
                    // coefficients of the sample tile t + 2 belongs to, before its transform starts (make_coef begins with a barrier behind
                    // every transform of this iteration)
                    if (t + 2 < t1) { int f2, ty2, tx2; decode(t + 2, f2, ty2, tx2); const int b2 = f2 / P.F; if (b2 != bcoef) { make_coef(b2); bcoef = b2; } }
                }
            }
            buf = bufn;
            __syncthreads();                          // pass j + 1 landed everywhere (PRO: tile t + 1 transformed); everybody is done reading pass j
        }
        fcur = fn; tyc = tyn; txc = txn;
    }
    flush_stats(bcur);
}

static hipError_t launch_conv64q(const ConvArgs& a, hipStream_t st) {
    const int total = a.NF * (a.H >> 4) * (a.W >> 4);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    const int grid = std::min(total, cus);
    const int tpb = (total + grid - 1) / grid;
    const int nblocks = (total + tpb - 1) / tpb;
    const size_t lds = 3 * (size_t)C64D_APL + 128 * 8 + 64 * 4 + (64 + 64 + 64) * 4;
    auto launch = [&](auto kfn) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kfn, dim3(nblocks), dim3(512), lds, st, a, tpb, total);
        return hipGetLastError();
    };
    if (!a.x0_bf16 || a.res || (a.C1 && !a.x1_bf16)) return hipErrorInvalidValue;
    if (a.Cout == 32) {                               // dim-32 networks: 64 (= 32 + 32 or 64) or 128 (= 64 + 64) input channels, bf16 output
        if (a.pro || !a.y_bf16) return hipErrorInvalidValue;
        return a.C0 + a.C1 == 128 ? launch(conv64q_kernel<128, false, true, 32>) : launch(conv64q_kernel<64, false, true, 32>);
    }
    if (a.C0 + a.C1 == 128) {
        if (a.pro) return hipErrorInvalidValue;
        return a.y_bf16 ? launch(conv64q_kernel<128, false, true>) : launch(conv64q_kernel<128, false, false>);
    }
    if (a.pro) return a.y_bf16 ? launch(conv64q_kernel<64, true, true>) : launch(conv64q_kernel<64, true, false>);
    return a.y_bf16 ? launch(conv64q_kernel<64, false, true>) : launch(conv64q_kernel<64, false, false>);
}

static hipError_t launch_conv64p(const ConvArgs& a, hipStream_t st) {
    const int total = a.NF * (a.H >> 4) * (a.W >> 4);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    const int grid = std::min(total, cus);
    const int tpb = (total + grid - 1) / grid;
    const int nblocks = (total + tpb - 1) / tpb;
    const size_t lds = 9 * 64 * 128 + 2 * (size_t)C64_HALO * 128 + (64 + 64 + 64) * 4 + 128 * 8 + 64 * 4;      // weights, 2 halo tiles, coefA / coefD / gmean, chs, bias
    auto launch = [&](auto kfn) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kfn, dim3(nblocks), dim3(512), lds, st, a, tpb, total);
        return hipGetLastError();
    };
    const int v = (a.x0_bf16 ? 4 : 0) | (a.pro ? 2 : 0) | (a.y_bf16 ? 1 : 0);
    if (a.Cout == 32) {                               // C = 32 form: bf16 tensors only; 62 KB of LDS, two workgroups per CU
        if (!a.x0_bf16 || !a.y_bf16 || a.res) return hipErrorInvalidValue;
        const int grid2 = std::min(total, ((a.pro ? VDX_C32_PRO_WAVES : 4) / 2) * cus), tpb2 = (total + grid2 - 1) / grid2, nb2 = (total + tpb2 - 1) / tpb2;
        const size_t lds2 = 9 * 32 * 64 + 2 * (size_t)C64_HALO * 64 + (64 + 64 + 64) * 4 + 128 * 8 + 64 * 4;
        auto launch2 = [&](auto kfn) -> hipError_t {
            hipLaunchKernelGGL(kfn, dim3(nb2), dim3(512), lds2, st, a, tpb2, total);
            return hipGetLastError();
        };
        return a.pro ? launch2(conv64p_kernel<true, true, true, false, 32>) : launch2(conv64p_kernel<true, false, true, false, 32>);
    }
    if (a.res) {
        if (a.pro || !a.x0_bf16 || a.res_bf16) return hipErrorInvalidValue;      // (the fp32-input form with 32 more registers would spill)
        return a.y_bf16 ? launch(conv64p_kernel<true, false, true, true>) : launch(conv64p_kernel<true, false, false, true>);
    }
    switch (v) {
        case 0: return launch(conv64p_kernel<false, false, false>);
        case 1: return launch(conv64p_kernel<false, false, true>);
        case 2: return launch(conv64p_kernel<false, true, false>);
        case 3: return launch(conv64p_kernel<false, true, true>);
        case 4: return launch(conv64p_kernel<true, false, false>);
        case 5: return launch(conv64p_kernel<true, false, true>);
        case 6: return launch(conv64p_kernel<true, true, false>);
        default: return launch(conv64p_kernel<true, true, true>);
    }
}

// ---- persistent 3x3 conv with Cin = 128 (two-pointer concat of 64 + 64, or one 128-channel tensor), Cout = 64, bf16 inputs ----
// Same scheme as conv64p_kernel, but a workgroup owns HALF of the output channels (32 x 128 x 9 bf16 weights = 72 KB resident);
// two workgroups on the same XCD walk the same tile range, so the second reader of a tile finds it in L2.
// The 128-channel halo tile does not fit twice beside the weights, so the pipeline runs in PLANE steps (round 2, second form): the
// two 64-channel planes of the input have one LDS buffer each; while the 72 MFMAs of plane 0 of tile t run, plane 1 of tile t
// lands in the other buffer by LDS-DMA (global_load_lds, no staging registers, no LDS store instructions), and while plane 1 runs,
// plane 0 of tile t + 1 lands in the first.  One s_waitcnt vmcnt(0) + barrier per plane step.  (First form: both planes single-
// buffered, next tile prefetched into 44 registers and written between two barriers with the matrix pipe idle: 758 us per launch at
// B = 64; this form 626 us.  The MFMA phase itself is LDS-read bound -- one ds_read_b128 per MFMA with 32 x 32 wave tiles.  Measured
// without gain on top of it: fragment reads software-pipelined two groups ahead of their MFMAs (626 us: LDS latency is not what is
// exposed); the LDS-DMA pieces issued one per MFMA group instead of all after the barrier (667 us).)
// A wave-instruction of the DMA covers 8 halo rows x 128 B; lane l fetches the global chunk (l & 7) ^ (l >> 3) of row l >> 3 so that
// it lands at the swizzled position.  Out-of-image pieces read a zero page.  No prologue (these are the first convs of
// ResnetBlocks whose input is a concat).
constexpr int C128_WPL = 9 * 32 * 128;                // bytes per weight plane [9 taps x 32 rows][128 B]
constexpr int C128_AROWS = 328;                       // 324 halo rows padded to 41 DMA instructions of 8 rows
constexpr int C128_APL = C128_AROWS * 128;            // bytes per activation plane buffer
__device__ __attribute__((aligned(16))) unsigned g_zero_page_c128[4];

__global__ __launch_bounds__(512) void conv128x64p_kernel(const ConvArgs P, const int tiles_per_block, const int total_tiles) {
    using M = Mma<MODE_BF16>;
    constexpr int NDMA = 41, NK = (NDMA + 7) / 8;      // DMA instructions per plane; per wave at most NK
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Wl = smem;                                   // [2 planes][288 rows][128 B]
    char* Al = Wl + 2 * C128_WPL;                      // [2 planes][328 rows][128 B]
    double* chs = reinterpret_cast<double*>(Al + 2 * C128_APL); // [2][32] (f64: order-independent)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int lp = lane & 15, q = lane >> 4;
    // workgroup ids are dealt round-robin to the 8 XCDs: ids i and i + 8 (same XCD, same L2, dispatched together) are the two
    // channel halves of one tile range, so the second reader of a tile finds it in L2 (ids 2r / 2r + 1 sit on different XCDs
    // and both fetched every tile from HBM: 1.49 GB per launch measured against 0.81 GB algorithmic)
    const int half = (blockIdx.x >> 3) & 1, co0 = half * 32;
    const int range = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7);
    const int tiles_x = P.W >> 4, tiles_pf = tiles_x * (P.H >> 4);
    const int t0 = range * tiles_per_block, t1 = min(t0 + tiles_per_block, total_tiles);
    if (t0 >= t1) return;

    for (int i = tid; i < 9 * 32 * 16; i += 512) {     // packed [tap][64 co][128 ci] bf16: 256-byte rows
        const int row = i >> 4, c = i & 15;            // row = tap * 32 + local co
        const int tap = row >> 5, col = row & 31;
        const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(P.wp) + ((size_t)(tap * 64 + co0 + col) * 256) + c * 16);
        *reinterpret_cast<uint4*>(Wl + (c >> 3) * C128_WPL + swz(row, c & 7)) = v;
    }
    if (tid < 64) chs[tid] = 0.0;

    // DMA pieces of this lane: instruction u = wave + 8 k covers halo rows 8 u .. 8 u + 7; the lane's row is 8 u + (lane >> 3)
    int pyx[NK];                                       // (halo row << 5) | halo column; row 100 = padding row / no instruction
    const int cbyte = ((lane & 7) ^ (lane >> 3)) * 16; // global chunk that lands at LDS position lane & 7 of a row with key lane >> 3
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int u = wave + 8 * k, hp = u * 8 + (lane >> 3);
        pyx[k] = (((u < NDMA && hp < C64_HALO) ? hp / 18 : 100) << 5) | (hp % 18);
    }
    const char* const zero_page = reinterpret_cast<const char*>(g_zero_page_c128);
    const char* const xb0 = reinterpret_cast<const char*>(P.x0);
    const char* const xb1 = P.C1 ? reinterpret_cast<const char*>(P.x1) : xb0 + 128;       // plane 1: second tensor, or channels 64..127
    const int rowb = P.C1 ? 128 : 256;                 // bytes per pixel row of a plane's tensor
    const unsigned al_base = lds_addr(Al);
    auto decode = [&](int t, int& f, int& ty, int& tx) { f = t / tiles_pf; const int r = t - f * tiles_pf; ty = r / tiles_x; tx = r - ty * tiles_x; };
    auto dma = [&](int t, int pl) {                    // plane pl of tile t -> buffer pl
        int f, ty, tx; decode(t, f, ty, tx);
        const char* xb = pl ? xb1 : xb0;
        const unsigned dst = al_base + pl * C128_APL + wave_u * 1024;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if (wave_u + 8 * k >= NDMA) continue;      // (uniform)
            const int gy = ty * 16 - 1 + (pyx[k] >> 5), gx = tx * 16 - 1 + (pyx[k] & 31);
            const bool ok = gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
            const size_t off = (size_t)((f * P.H + gy) * P.W + gx) * rowb + cbyte;
            glds16(ok ? static_cast<const void*>(xb + off) : static_cast<const void*>(zero_page), dst + k * 8 * 1024);
        }
    };
    f32x4 ssum[2], ssq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { ssum[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ssq[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    auto flush_stats = [&](int b) {
        if (!P.out_stats) return;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s1 = reduce16(ssum[tm][e]), s2 = reduce16(ssq[tm][e]);
                if (lp == 0) { unsafeAtomicAdd(&chs[tm * 16 + 4 * q + e], (double)s1); unsafeAtomicAdd(&chs[32 + tm * 16 + 4 * q + e], (double)s2); }
                ssum[tm][e] = 0.f; ssq[tm][e] = 0.f;
            }
        __syncthreads();
        const int cpg = 64 / P.out_groups, ng = 32 / cpg;       // groups inside this half
        if (tid < 2 * ng) {
            const int g = tid >> 1, which = tid & 1;
            double t = 0.0;
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) t += chs[which * 32 + c];
            unsafeAtomicAdd(P.out_stats + (((size_t)b * GN_SLOTS + ((blockIdx.x >> 1) % GN_SLOTS)) * P.out_groups + half * ng + g) * 2 + which, t);
        }
        __syncthreads();
        if (tid < 64) chs[tid] = 0.0;
        __syncthreads();
    };
    int hpb[2];
    hpb[0] = (2 * wave) * 18 + lp; hpb[1] = hpb[0] + 18;
    float4 bias4[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) bias4[tm] = P.bias ? *reinterpret_cast<const float4*>(P.bias + co0 + tm * 16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);

    f32x4 acc[2][2];
    auto plane_mma = [&](int pl) {                     // 9 taps x 2 chunks x (2 x 2) MFMAs on plane pl
        const char* Wp = Wl + pl * C128_WPL;
        const char* Ap = Al + pl * C128_APL;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            int boff[2];
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) { const int hp = hpb[tn] + dy * 18 + dx; boff[tn] = swz(hp, q); }
            const int woff = swz(tap * 32 + lp, q);
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                uint4 af[2], bf[2];
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) af[tm] = *reinterpret_cast<const uint4*>(Wp + ((woff + tm * 16 * 128) ^ (ch * 64)));
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) bf[tn] = *reinterpret_cast<const uint4*>(Ap + (boff[tn] ^ (ch * 64)));
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) M::mma(acc[tm][tn], af[tm], bf[tn]);
            }
        }
    };

    int fcur, tyc, txc;
    decode(t0, fcur, tyc, txc);
    int bcur = fcur / P.F;
    dma(t0, 0);
    wait_vm<0>();
    __syncthreads();                                   // weights, chs and plane 0 of the first tile visible
    for (int t = t0; t < t1; ++t) {
        const bool more = t + 1 < t1;
        int fn = fcur, tyn = tyc, txn = txc, bn = bcur;
        if (more) { decode(t + 1, fn, tyn, txn); bn = fn / P.F; }
        dma(t, 1);                                     // buffer 1 was last read during plane 1 of tile t - 1 (barrier since)
#pragma unroll
        for (int i = 0; i < 2; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        plane_mma(0);
        wait_vm<0>();                                  // this wave's pieces of plane 1 (and the stores of tile t - 1) are done
        __syncthreads();                               // everybody's are; every wave is done reading buffer 0
        if (more) dma(t + 1, 0);
        plane_mma(1);
        wait_vm<0>();                                  // before the stores below: the wait covers the DMA only
        {
            const int oy0 = tyc * 16 + 2 * wave, ox = txc * 16 + lp;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const size_t gout = ((size_t)(fcur * P.H + oy0 + tn) * P.W + ox) * 64 + co0;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    const float4 v = make_float4(acc[tm][tn][0] + bias4[tm].x, acc[tm][tn][1] + bias4[tm].y, acc[tm][tn][2] + bias4[tm].z, acc[tm][tn][3] + bias4[tm].w);
                    store4_f32_or_bf16(P.y, gout + tm * 16 + 4 * q, v, P.y_bf16);
                    ssum[tm][0] += v.x; ssum[tm][1] += v.y; ssum[tm][2] += v.z; ssum[tm][3] += v.w;
                    ssq[tm][0] += v.x * v.x; ssq[tm][1] += v.y * v.y; ssq[tm][2] += v.z * v.z; ssq[tm][3] += v.w * v.w;
                }
            }
        }
        if (more && bn != bcur) { flush_stats(bcur); bcur = bn; }
        fcur = fn; tyc = tyn; txc = txn;
        __builtin_amdgcn_s_barrier();                  // plane 0 of tile t + 1 landed everywhere; every wave is done reading buffer 1
    }
    flush_stats(bcur);
}

static hipError_t launch_conv128x64p(const ConvArgs& a, hipStream_t st) {
    const int total = a.NF * (a.H >> 4) * (a.W >> 4);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    const int ranges = std::max(1, std::min(total, cus / 2));
    const int tpb = (total + ranges - 1) / ranges;
    const int nranges = (total + tpb - 1) / tpb;
    const size_t lds = 2 * (size_t)C128_WPL + 2 * (size_t)C128_APL + 64 * 8;
    auto kfn = conv128x64p_kernel;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3((nranges + 7) / 8 * 16), dim3(512), lds, st, a, tpb, total);
    return hipGetLastError();
}

// Flax kernel [taps][Cin][Cout] fp32  ->  packed [taps][Cout][CinPad] in the MMA element type, zero padded.
template <int MODE>
__global__ void pack_weights_kernel(const float* __restrict__ src, void* __restrict__ dst, int taps, int Cin, int Cout, int CinPad) {
    const size_t n = (size_t)taps * Cout * CinPad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % CinPad);
        const size_t r = i / CinPad;
        const int co = (int)(r % Cout);
        const int t = (int)(r / Cout);
        const float v = (ci < Cin) ? src[((size_t)t * Cin + ci) * Cout + co] : 0.f;
        store_operand<MODE>(dst, i, v);
    }
}

// Transposed + tap-reversed packing for the data gradient: Flax kernel [taps][Cin][Cout] fp32 ->
// [taps][Cin (rows)][CoutPad (K)], tap t' = taps-1-t.  conv(dy, this) == d/dx of conv(x, kernel) for stride 1,
// and it turns Downsample's dgrad into the ConvTranspose kernel and vice versa (backward).
template <int MODE>
__global__ void pack_weights_t_kernel(const float* __restrict__ src, void* __restrict__ dst, int taps, int Cin, int Cout, int CoutPad) {
    const size_t n = (size_t)taps * Cin * CoutPad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % CoutPad);
        const size_t r = i / CoutPad;
        const int ci = (int)(r % Cin);
        const int t = (int)(r / Cin);
        const float v = (co < Cout) ? src[((size_t)(taps - 1 - t) * Cin + ci) * Cout + co] : 0.f;
        store_operand<MODE>(dst, i, v);
    }
}

// every tensor of a parameter (re)packing in one launch: blockIdx.y = job (model.h PackJob); a workgroup walks destination rows
// (one scalar division per row), its threads the contiguous K of the row -- no per-element division
template <int MODE>
__global__ __launch_bounds__(256) void pack_jobs_kernel(const float* __restrict__ params, char* __restrict__ dst_base, const PackJob* __restrict__ jobs) {
    const PackJob J = jobs[blockIdx.y];
    const float* src = params + J.src;
    char* dstc = dst_base + J.dst;
    if (J.kind == 2) {                                        // plain copy
        for (int i = blockIdx.x * 256 + threadIdx.x; i < (int)J.n; i += gridDim.x * 256) reinterpret_cast<float*>(dstc)[i] = src[i];
        return;
    }
    if (J.kind == 0) {                                        // [tap][Cin][Cout] -> [tap][Cout][Pad]: 32 x 32 tiles transposed through LDS
        __shared__ float tile[32][33];
        const int tk = (J.Pad + 31) / 32, tc = (J.Cout + 31) / 32, per_tap = tk * tc, ntiles = J.taps * per_tap;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int id = blockIdx.x; id < ntiles; id += gridDim.x) {
            const int t = id / per_tap, rem = id - t * per_tap, kb = rem / tc, cb = rem - kb * tc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kk = kb * 32 + ty + 8 * j, co = cb * 32 + tx;
                tile[ty + 8 * j][tx] = (kk < J.Cin && co < J.Cout) ? src[((size_t)t * J.Cin + kk) * J.Cout + co] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = cb * 32 + ty + 8 * j, kk = kb * 32 + tx;
                if (co < J.Cout && kk < J.Pad) {
                    const size_t di = ((size_t)t * J.Cout + co) * J.Pad + kk;
                    const float v = tile[tx][ty + 8 * j];
                    store_operand<MODE>(dstc, di, v);
                }
            }
            __syncthreads();
        }
        return;
    }
    const int rows = (int)(J.n / J.Pad);
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const size_t drow = (size_t)r * J.Pad;
        if (false) {
        } else if (J.kind == 1) {                             // row = (reversed tap, input channel), k = output channel
            const int t = r / J.Cin, ci = r - t * J.Cin;
            const float* srow = src + ((size_t)(J.taps - 1 - t) * J.Cin + ci) * J.Cout;
            for (int k = threadIdx.x; k < J.Pad; k += 256) {
                const float v = (k < J.Cout) ? srow[k] : 0.f;
                store_operand<MODE>(dstc, drow + k, v);
            }
        } else {                                              // kind 3: row = input channel, k = (tensor, output channel) of three kernels
            for (int k = threadIdx.x; k < J.Pad; k += 256) {
                const int sel = k / J.Cout, kk = k - sel * J.Cout;
                const float* sp = params + (sel == 0 ? J.src : (sel == 1 ? J.src1 : J.src2));
                const float v = (sel < 3) ? sp[(size_t)r * J.Cout + kk] : 0.f;
                store_operand<MODE>(dstc, drow + k, v);
            }
        }
    }
}

// ---- host-side launchers ----------------------------------------------------------------------------

static void choose_patch(int BM, int NF, int F, int Ho, int Wo, int stride, int K, int& PH, int& PW, int& NP) {
    int pw = 4;
    while (pw < Wo && pw < 16) pw *= 2;
    long best = -1;
    PH = BM / pw; PW = pw; NP = 1;
    for (int np = 1; np <= 16; np *= 2) {
        if (F % np) continue;
        const int ph = BM / (pw * np);
        if (ph < 1) continue;
        const int IH = (ph - 1) * stride + K, IW = (pw - 1) * stride + K;
        if ((long)np * IH * IW > 400) continue;                  // keep the halo tile within the LDS budget
        const long blocks = (long)(NF / np) * ((Ho + ph - 1) / ph) * ((Wo + pw - 1) / pw);
        if (best < 0 || blocks < best) { best = blocks; PH = ph; PW = pw; NP = np; }
    }
}

size_t conv_packed_bytes(int mode, int taps, int Cin, int Cout) {
    const int KT = mode == MODE_F32 ? Mma<MODE_F32>::KT : Mma<MODE_BF16>::KT;
    const int ES = mode == MODE_F32 ? 4 : 2;
    const int CinPad = (Cin + KT - 1) / KT * KT;
    return (size_t)taps * Cout * CinPad * ES;
}

int conv_cin_pad(int mode, int Cin) {
    const int KT = mode == MODE_F32 ? Mma<MODE_F32>::KT : Mma<MODE_BF16>::KT;
    return (Cin + KT - 1) / KT * KT;
}

hipError_t launch_pack_weights(int mode, const float* src, void* dst, int taps, int Cin, int Cout, hipStream_t st) {
    const int CinPad = conv_cin_pad(mode, Cin);
    const size_t n = (size_t)taps * Cout * CinPad;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 2048);
    if (mode == MODE_F32) hipLaunchKernelGGL(pack_weights_kernel<MODE_F32>, dim3(blocks), dim3(256), 0, st, src, dst, taps, Cin, Cout, CinPad);
    else if (mode == MODE_F16) hipLaunchKernelGGL(pack_weights_kernel<MODE_F16>, dim3(blocks), dim3(256), 0, st, src, dst, taps, Cin, Cout, CinPad);
    else hipLaunchKernelGGL(pack_weights_kernel<MODE_BF16>, dim3(blocks), dim3(256), 0, st, src, dst, taps, Cin, Cout, CinPad);
    return hipGetLastError();
}

hipError_t launch_pack_jobs(int mode, const float* params, void* dst_base, const PackJob* d_jobs, int njobs, hipStream_t st) {
    if (njobs <= 0) return hipSuccess;
    dim3 grid(96, njobs);
    if (mode == MODE_F32) hipLaunchKernelGGL(pack_jobs_kernel<MODE_F32>, grid, dim3(256), 0, st, params, reinterpret_cast<char*>(dst_base), d_jobs);
    else if (mode == MODE_F16) hipLaunchKernelGGL(pack_jobs_kernel<MODE_F16>, grid, dim3(256), 0, st, params, reinterpret_cast<char*>(dst_base), d_jobs);
    else hipLaunchKernelGGL(pack_jobs_kernel<MODE_BF16>, grid, dim3(256), 0, st, params, reinterpret_cast<char*>(dst_base), d_jobs);
    return hipGetLastError();
}

hipError_t launch_pack_weights_t(int mode, const float* src, void* dst, int taps, int Cin, int Cout, hipStream_t st) {
    const int CoutPad = conv_cin_pad(mode, Cout);
    const size_t n = (size_t)taps * Cin * CoutPad;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 2048);
    if (mode == MODE_F32) hipLaunchKernelGGL(pack_weights_t_kernel<MODE_F32>, dim3(blocks), dim3(256), 0, st, src, dst, taps, Cin, Cout, CoutPad);
    else if (mode == MODE_F16) hipLaunchKernelGGL(pack_weights_t_kernel<MODE_F16>, dim3(blocks), dim3(256), 0, st, src, dst, taps, Cin, Cout, CoutPad);
    else hipLaunchKernelGGL(pack_weights_t_kernel<MODE_BF16>, dim3(blocks), dim3(256), 0, st, src, dst, taps, Cin, Cout, CoutPad);
    return hipGetLastError();
}

// instrumentation (vdx.h: vdx_set_launch_hook): algorithmic work of one conv launch -- 2 * pixels_out * Cin * Cout * taps FLOP (ConvTranspose:
// 4 effective taps per output pixel) and input + output (+ residual) tensors in their storage type + the packed weights
namespace {
struct ConvWork { double flops, bytes; char shape[128]; };
ConvWork conv_work(int mode, const ConvArgs& a) {
    ConvWork w;
    const double es = mode == MODE_F32 ? 4.0 : 2.0;
    const int taps = a.kind ? 16 : a.kh * a.kw;
    const double ho = a.kind ? 2.0 * a.H : (a.H + a.stride - 1) / a.stride, wo = a.kind ? 2.0 * a.W : (a.W + a.stride - 1) / a.stride;
    const double cin = a.C0 + a.C1;
    w.flops = 2.0 * a.NF * ho * wo * cin * a.Cout * (a.kind ? 4 : taps);
    w.bytes = (double)a.NF * a.H * a.W * (a.C0 * (a.x0_bf16 ? 2.0 : 4.0) + a.C1 * (a.x1_bf16 ? 2.0 : 4.0)) + a.NF * ho * wo * a.Cout * (a.y_bf16 ? 2.0 : 4.0)
            + (a.res ? a.NF * ho * wo * a.Cout * (a.res_bf16 ? 2.0 : 4.0) : 0.0) + es * taps * cin * a.Cout;
    snprintf(w.shape, sizeof(w.shape), "%s%dx%d %d->%d %dx%dx%d%s%s%s%s", a.kind ? "convT" : "conv", a.kind ? 4 : a.kh, a.kind ? 4 : a.kw, a.C0 + a.C1, a.Cout,
             a.NF, a.H, a.W, a.stride == 2 ? " s2" : "", a.pro ? " +prologue" : "", a.res ? " +res" : "", a.C1 ? " concat" : "");
    return w;
}
}  // namespace

hipError_t launch_conv(int mode, ConvArgs a, hipStream_t st) {
    // geometry completion
    const int K = a.kind ? 2 : a.kh;
    if (a.kind) { a.stride = 1; a.Ho = a.H; a.Wo = a.W; a.Hy = 2 * a.H; a.Wy = 2 * a.W; }
    else { a.Ho = (a.H + a.stride - 1) / a.stride; a.Wo = (a.W + a.stride - 1) / a.stride; a.Hy = a.Ho; a.Wy = a.Wo; }
    a.CinPad = conv_cin_pad(mode, a.C0 + a.C1);
    const int ES = mode == MODE_F32 ? 4 : 2;
    const size_t npix = (size_t)a.NF * a.H * a.W;
    if (a.wrows <= 0) { a.wrows = a.Cout; a.wrow0 = 0; }
    const size_t b0 = npix * a.C0 * (a.x0_bf16 ? 2 : 4), b1 = npix * a.C1 * (a.x1_bf16 ? 2 : 4), bw = (size_t)(a.kind ? 16 : a.kh * a.kw) * a.wrows * a.CinPad * ES;
    if (b0 >= 0xFFFFFFF0ull || b1 >= 0xFFFFFFF0ull || bw >= 0xFFFFFFF0ull) return hipErrorInvalidValue;   // 32-bit buffer offsets
    a.x0_bytes = (unsigned)b0; a.x1_bytes = (unsigned)b1; a.w_bytes = (unsigned)bw;
    if (conv3x3_ws_eligible(mode, a)) {                                      // wide levels: persistent weight-streaming kernel
        const ConvWork cw = conv_work(mode, a);
        LaunchScope ls(st, "conv3x3_ws_kernel", cw.flops, cw.bytes, "<%d, %s> %s", conv3x3_ws_geo(a), a.pro ? "true" : "false", cw.shape);
        return launch_conv3x3_ws(a, st);
    }
    if (conv4x4_ws_eligible(mode, a)) {                                      // 4x4 resampling convs of the wide levels
        const ConvWork cw = conv_work(mode, a);
        LaunchScope ls(st, "conv4x4_ws_kernel", cw.flops, cw.bytes, "<%d, %d, %d> %s", a.kind == 1 ? a.H : a.H / 2, a.kind == 1 ? 2 : 1, a.Cout == 64 ? 64 : 128, cw.shape);
        return launch_conv4x4_ws(a, st);
    }
    if (resample32_eligible(mode, a)) {                                      // 32-channel resampling convs (level 0 of dim-32 networks)
        const ConvWork cw = conv_work(mode, a);
        LaunchScope ls(st, "resample32_kernel", cw.flops, cw.bytes, "<%d> %s", a.kind, cw.shape);
        return launch_resample32(a, st);
    }
    if (conv1x1_pw_eligible(mode, a)) {                                      // 1x1 convs of the wide levels (bf16 tensors)
        const ConvWork cw = conv_work(mode, a);
        LaunchScope ls(st, "conv1x1_pw_kernel", cw.flops, cw.bytes, "<%d> %s", conv1x1_pw_rows(a), cw.shape);
        return launch_conv1x1_pw(a, st);
    }
    {   // persistent specialisation for the level-0 shape (see conv64p_kernel)
        const int use64p = 1;
        const long tiles = (long)a.NF * (a.H / 16) * (a.W / 16);
        if (use64p && mode == MODE_BF16 && a.kind == 0 && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.C0 == 64 && a.C1 == 0 && a.Cout == 64 &&
            a.wrows >= 64 && a.wrow0 >= 0 && a.wrow0 + 64 <= a.wrows && (!a.res || (!a.pro && a.x0_bf16 && !a.res_bf16)) && a.H % 16 == 0 && a.W % 16 == 0 && tiles >= 1024 && npix * 64 * (a.y_bf16 ? 2 : 4) < 0xFFFFFFF0ull && (!a.pro || (a.groups <= 32 && 64 % a.groups == 0)) &&
            (!a.out_stats || (a.out_groups <= 32 && 64 % a.out_groups == 0))) {
#ifndef VDX_C64_NODMA
#define VDX_C64_NODMA 0
#endif
            const bool dma_form = a.x0_bf16 && !a.pro && !VDX_C64_NODMA;               // bf16 input, no prologue: input staged by LDS-DMA (conv64d_kernel)
#ifndef VDX_C64R
#define VDX_C64R 1
#endif
#ifndef VDX_C64Q
#define VDX_C64Q 13            // conv64q_kernel (16 channels x 128 pixels per wave) for: 1 = the prologue form, 2 = the plain form, 4 = 128 input channels, 8 = 32 output channels
#endif
            if (a.x0_bf16 && !a.res && !VDX_C64_NODMA && (VDX_C64Q & (a.pro ? 1 : 2))) {
                const ConvWork cw = conv_work(mode, a);
                LaunchScope ls(st, "conv64q_kernel", cw.flops, cw.bytes, "<cin 64, pro %d, y16 %d> %s", a.pro, a.y_bf16, cw.shape);
                return launch_conv64q(a, st);
            }
            if (VDX_C64R && a.x0_bf16 && !a.pro && !a.res && !VDX_C64_NODMA) {      // weights in registers, three-deep tile ring (conv64r_kernel; its prologue form -- in-place transform with
                                                                                       // the coefficients re-read per piece, 256 registers + scratch -- measured 449 us against conv64p_kernel's 422 and was removed)
                const ConvWork cw = conv_work(mode, a);
                LaunchScope ls(st, "conv64r_kernel", cw.flops, cw.bytes, "<pro 0, y16 %d> %s", a.y_bf16, cw.shape);
                return launch_conv64r(a, st);
            }
            const ConvWork cw = conv_work(mode, a);
            LaunchScope ls(st, dma_form ? "conv64d_kernel" : "conv64p_kernel", cw.flops, cw.bytes, "<x16 %d, pro %d, y16 %d, res %d> %s", a.x0_bf16, a.pro, a.y_bf16, a.res ? 1 : 0, cw.shape);
            return dma_form ? launch_conv64d(a, st) : launch_conv64p(a, st);
        }
        // level 0 of dim-32 networks (the YAML-literal config_v2_2): the same kernel with 32-channel tiles, bf16 tensors
        if (use64p && mode == MODE_BF16 && a.kind == 0 && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.C0 == 32 && a.C1 == 0 && a.Cout == 32 && a.wrows == 32 &&
            a.wrow0 == 0 && a.CinPad == 64 && a.x0_bf16 && a.y_bf16 && !a.res && a.H % 16 == 0 && a.W % 16 == 0 && tiles >= 1024 &&
            (!a.pro || (a.groups <= 32 && 32 % a.groups == 0)) && (!a.out_stats || (a.out_groups <= 32 && 32 % a.out_groups == 0))) {
            const ConvWork cw = conv_work(mode, a);
            LaunchScope ls(st, "conv64p_kernel", cw.flops, cw.bytes, "<x16 1, pro %d, y16 1, res 0, C 32> %s", a.pro, cw.shape);
            return launch_conv64p(a, st);
        }
        const bool in16c = a.x0_bf16 && (!a.C1 || a.x1_bf16);
        // dim-32 networks (the YAML-literal config_v2_2): the 3x3 convs on a concat input with 32 output channels (64 = 32 + 32 -> 32 at level 0,
        // 128 = 64 + 64 -> 32 at level 1) ran on the generic kernel (278 / 117 us at B = 64)
        if ((VDX_C64Q & 8) && use64p && mode == MODE_BF16 && a.kind == 0 && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.Cout == 32 && in16c && !a.pro && a.y_bf16 &&
            ((a.C0 == 32 && a.C1 == 32) || (a.C0 == 64 && a.C1 == 64)) && a.CinPad == a.C0 + a.C1 && a.wrows == 32 && a.wrow0 == 0 && !a.res && a.H % 16 == 0 && a.W % 16 == 0 &&
            tiles >= 1024 && (!a.out_stats || (a.out_groups <= 32 && 32 % a.out_groups == 0 && (32 / a.out_groups) <= 8 && 8 % (32 / a.out_groups) == 0))) {
            const ConvWork cw = conv_work(mode, a);
            LaunchScope ls(st, "conv64q_kernel", cw.flops, cw.bytes, "<cin %d, pro 0, y16 1, cout 32> %s", a.C0 + a.C1, cw.shape);
            return launch_conv64q(a, st);
        }
        if (use64p && mode == MODE_BF16 && a.kind == 0 && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.Cout == 64 && in16c && !a.pro &&
            ((a.C0 == 64 && a.C1 == 64) || (a.C0 == 128 && a.C1 == 0)) && a.wrows == 64 && a.wrow0 == 0 && !a.res && a.H % 16 == 0 && a.W % 16 == 0 &&
            tiles >= 1024 && (!a.out_stats || (a.out_groups <= 32 && 32 % (64 / a.out_groups) == 0 && 64 % a.out_groups == 0))) {
            const ConvWork cw = conv_work(mode, a);
            if ((VDX_C64Q & 4) && (long)a.NF * a.H * a.W * 256 < 0xFFFFFFF0l) {
                LaunchScope ls(st, "conv64q_kernel", cw.flops, cw.bytes, "<cin 128, pro 0, y16 %d> %s", a.y_bf16, cw.shape);
                return launch_conv64q(a, st);
            }
            LaunchScope ls(st, "conv128x64p_kernel", cw.flops, cw.bytes, "%s", cw.shape);
            return launch_conv128x64p(a, st);
        }
    }
    // variant: 64-channel tiles take 256 pixels per workgroup (stride 1) so every wave owns a 64x64 tile
    const int BC = a.Cout <= 64 ? 64 : 128;
    // 8-wave workgroups (each wave 32 pixels x 64 channels of the same workgroup tile): 4 waves per SIMD instead of 2;
    // (small launches -- the deep levels at the training batch, 128 tiles of 128 x 128 -- as 256 four-wave workgroups of 128 x 64 were
    // measured slower: 16.95 vs 16.75 ms per train step; the halo tile is staged twice as often)
    const int TN = (BC == 64 && a.stride == 2) ? 2 : 4;
    const int BM = 16 * TN * (4 / (BC / 64));
    choose_patch(BM, a.NF, a.F, a.Ho, a.Wo, a.stride, K, a.PH, a.PW, a.NP);
    a.tiles_y = (a.Ho + a.PH - 1) / a.PH;
    a.tiles_x = (a.Wo + a.PW - 1) / a.PW;
    const int IH = (a.PH - 1) * a.stride + K, IW = (a.PW - 1) * a.stride + K;
    const int HPX = a.NP * IH * IW;
    auto magic = [](int d) { return (unsigned)((1ull << 32) / (unsigned)d) + 1u; };      // every divisor here is >= 2
    a.m_ihiw = magic(IH * IW); a.m_iw = magic(IW); a.m_phpw = magic(a.PH * a.PW); a.m_pw = magic(a.PW);
    if (IW < 2 || a.PW < 2) return hipErrorInvalidValue;
    size_t lds = 2 * BC * 8 + ((HPX * 4 + 15) / 16) * 16 + (a.pro ? (2 * a.CinPad * 4 + 64 * 4) : 0)
               + (size_t)HPX * ROW_STRIDE + 3 * (size_t)BC * 128;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    dim3 grid((a.NF / a.NP) * a.tiles_y * a.tiles_x, (a.Cout + BC - 1) / BC, a.kind ? 4 : 1);
    const bool in16 = mode == MODE_BF16 && a.x0_bf16 && (!a.C1 || a.x1_bf16) && (a.C0 % 8 == 0) && (a.C1 % 8 == 0);
    const int inf = in16 ? 2 : (!a.x0_bf16 && !(a.C1 && a.x1_bf16)) ? 0 : (mode == MODE_BF16 && a.x0_bf16 && !a.C1) ? 1 : 3;
    const ConvWork cw = conv_work(mode, a);
    LaunchScope ls(st, "conv_igemm_kernel", cw.flops, cw.bytes, "<%d, %d, 2, %d, %d> %s", mode, BC, (BC == 128 || TN == 4) ? 8 : 4, (mode == MODE_BF16 || inf == 0) ? inf : 3, cw.shape);
#define VDX_LAUNCH_CONV_K(KFN_, NTH_)                                                                    \
    do {                                                                                                  \
        auto kfn = KFN_;                                                                                  \
        if (lds > 64 * 1024) {                                                                            \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                \
        }                                                                                                 \
        hipLaunchKernelGGL(kfn, grid, dim3(NTH_), lds, st, a);                                            \
    } while (0)
    // 64-pixel-per-wave shapes run as 8 waves x 32 pixels (measured better than 4 waves x 64 pixels); stride-2 64-channel tiles as 4 x 32
#define VDX_LAUNCH_CONV_T(MODE_, INF_)                                                                   \
    do {                                                                                                  \
        if (BC == 128) VDX_LAUNCH_CONV_K((conv_igemm_kernel<MODE_, 128, 2, 8, INF_>), 512);              \
        else if (TN == 4) VDX_LAUNCH_CONV_K((conv_igemm_kernel<MODE_, 64, 2, 8, INF_>), 512);            \
        else VDX_LAUNCH_CONV_K((conv_igemm_kernel<MODE_, 64, 2, 4, INF_>), 256);                          \
    } while (0)
    if (mode == MODE_F32) {
        if (inf == 0) VDX_LAUNCH_CONV_T(MODE_F32, 0); else VDX_LAUNCH_CONV_T(MODE_F32, 3);
    } else if (mode == MODE_F16) {
        if (inf != 0) return hipErrorInvalidValue;           // fp16 operand mode keeps every tensor fp32 in HBM
        VDX_LAUNCH_CONV_T(MODE_F16, 0);
    } else {
        if (inf == 0) VDX_LAUNCH_CONV_T(MODE_BF16, 0); else if (inf == 1) VDX_LAUNCH_CONV_T(MODE_BF16, 1);
        else if (inf == 2) VDX_LAUNCH_CONV_T(MODE_BF16, 2); else VDX_LAUNCH_CONV_T(MODE_BF16, 3);
    }
#undef VDX_LAUNCH_CONV_T
#undef VDX_LAUNCH_CONV_K
    return hipGetLastError();
}

}  // namespace vdx
