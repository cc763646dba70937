// Persistent, weight-streaming (1,3,3) convolution for the wide levels (Cout >= 128) of the bf16 path (gfx950).
//
// Replaces conv_igemm_kernel<1,128,2,8,2> for Block.proj (reference modules.py:162-172) where it dominated the sampling step
// (r01: 0.32 of the bf16 MFMA peak, 43 % of wave life parked on waits, ~1800 non-MFMA instructions of per-workgroup set-up per
// 288 MFMAs).  Same math, same fusions (prologue = GroupNorm-apply * (scale+1) + shift -> SiLU on the input, epilogue = +bias and
// GroupNorm partial statistics of the output), different structure:
//
//   * PERSISTENT: one 8-wave workgroup per CU walks a contiguous range of 256-pixel tiles for one 128-channel output tile; every
//     per-lane table (halo piece -> pixel, fragment offsets, weight-row offsets, bias) is computed once per workgroup, not per tile.
//   * WEIGHT STREAM: the [tap][128 couts][64 cin] slabs (16 KB) of the packed weights flow through an NS-deep LDS ring filled by
//     LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write pass), NS - 1 slabs ahead of the MFMAs, across tile boundaries; ONE raw
//     s_barrier per slab, guarded by a COUNTED s_waitcnt vmcnt(N) so that the younger slabs stay in flight across it.
//   * HALO DOUBLE BUFFER: the (frames x (PH+2) x (PW+2)) x 64-channel input tile of the NEXT K chunk / pixel tile is fetched by
//     LDS-DMA while the 9 taps of the current one run; out-of-image pixels read a zero page, so the 3x3 window needs no bounds
//     logic.  With a prologue the issuing thread rewrites its own pieces in place (LDS -> registers -> LDS) between the taps.
//   * WAVE TILE 64 couts x 64 pixels (4 x 4 MFMA tiles of 16x16x32): 8 fragment reads per 16 MFMAs instead of 6 per 8.
//   * XCD-aware decode: the workgroups that share a pixel range (one per 128-channel output tile) are 8 ids apart -> same XCD,
//     same L2, so an input tile leaves HBM once.
//
// LDS rows are 128 bytes (64 bf16 of K); 16-byte chunk k of row r sits at position k ^ (r & 7) (conflict-free ds_read_b128 for the
// lane groups of 16 consecutive rows: MI355X_MICROARCH.md, LDS).  LDS-DMA writes lane-linear (base + 16 * lane), so the swizzle is
// applied to the SOURCE address of each lane and again when reading (cdna_hip_programming.md rule 21).
#include "vdx_common.h"
#include "vdx_internal.h"
#include <type_traits>

namespace vdx {

namespace {

__device__ __attribute__((aligned(16))) unsigned g_zero_page[4];      // source of every out-of-image halo piece

// 16-byte LDS-DMA: lane l's 16 bytes at `src` land at LDS byte address lds_wave_base + 16 l (wave-uniform base in M0).
// Inline asm on purpose: hipcc's wait-count pass puts an s_waitcnt vmcnt(0) in front of the next ds_read after the BUILTIN form
// (it cannot prove that the LDS-DMA destination and the read do not alias), which drains the ring every tap; the asm form is
// invisible to that pass and every wait on these loads is the hand-counted one at the tap's sync (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16(const void* src, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_wave_base) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const char* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// position-swizzled byte offset of 16-byte chunk (4 ks + q) of LDS row `row`, for ks = 0; ks = 1 is this ^ 64
__device__ __forceinline__ int frag_off(int row, int q) { return row * 128 + (((q ^ (row & 3)) | (row & 4)) << 4); }

constexpr int WS_SLAB = 128 * 128;            // one (tap, K chunk) weight slab: 128 couts x 64 cin bf16
constexpr int WS_STORES = 16;                 // global stores per wave in the tile epilogue (4 x 4 MFMA tiles, unconditional)

template <int N> __device__ __forceinline__ void wait_vm_lgkm0() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(N) : "memory");
}

}  // namespace

// PW = tile width in pixels: 16 -> 16 x 16 pixels of one frame, 8 -> four whole 8 x 8 frames.  NS = ring depth.
template <int PW, int NS, bool PRO>
__global__ __launch_bounds__(512, 2) void conv3x3_ws_kernel(const ConvArgs P, const int tiles_per_range, const int total_tiles, const int nct) {
    using M = Mma<MODE_BF16>;
    constexpr int PH = PW == 16 ? 16 : 8;
    constexpr int NP = 256 / (PW * PH);
    constexpr int IW = PW + 2, IH = PH + 2, HPX = NP * IH * IW;
    constexpr int NPIECE = HPX * 8;                       // 16-byte pieces of one halo buffer
    constexpr int NU = (NPIECE + 511) / 512;
    constexpr int NUMIN = NPIECE / 512;                   // halo LDS-DMA instructions EVERY wave issues per buffer
    constexpr int HBUF = HPX * 128;
    constexpr int WIN = 2 * (NS - 2);                     // weight LDS-DMA instructions younger than the slab a sync waits for

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;                                    // [NS][128 rows][128 B]
    char* halo = ring + NS * WS_SLAB;                     // [2][HPX rows][128 B]
    float* coefA = reinterpret_cast<float*>(halo + 2 * HBUF);   // PRO: [Cin] x_hat = x * a + d
    float* coefD = coefA + P.CinPad;
    float* gmean = coefD + P.CinPad;                      // [32][mean, rstd]

    const unsigned ring_a = lds_addr(ring), halo_a = lds_addr(halo);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int wc = wave & 1, wpx = wave >> 1;
    // ids i, i + 8, ... share an XCD: the nct output-channel tiles of one pixel range are consecutive slots of one XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int j = slot % nct;
    const int range = (slot / nct) * 8 + xcd;
    const int t0 = range * tiles_per_range, t1 = min(t0 + tiles_per_range, total_tiles);
    if (t0 >= t1) return;
    const int tiles_x = P.W / PW, tiles_pf = tiles_x * (P.H / PH);
    const int nchunks = P.CinPad >> 6;
    const int Cin = P.C0 + P.C1;

    // ---- per-thread constants -----------------------------------------------------------------------------------------
    // weight stream: this thread's two 16-byte pieces of a slab
    unsigned wsrc[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int i2 = (v * 8 + wave) * 64 + lane, row = i2 >> 3, pos = i2 & 7;
        wsrc[v] = (unsigned)((j * 128 + row) * P.CinPad) * 2u + (unsigned)((pos ^ (row & 7)) << 4);
    }
    const size_t tap_stride = (size_t)P.Cout * P.CinPad * 2;
    const char* wbase = reinterpret_cast<const char*>(P.wp);
    // halo pieces: piece i = (u * 8 + wave) * 64 + lane -> LDS row i >> 3, position i & 7 (holds source chunk pos ^ (row & 7))
    int hinfo[NU];                                        // iy | ix << 8 | patch << 16 | source chunk << 24 ; < 0: no piece
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int i = (u * 8 + wave) * 64 + lane, row = i >> 3, pos = i & 7;
        const int patch = row / (IH * IW), rr = row - patch * (IH * IW);
        const int iy = rr / IW, ix = rr - iy * IW;
        hinfo[u] = (i < NPIECE) ? (iy | (ix << 8) | (patch << 16) | ((pos ^ (row & 7)) << 24)) : -1;
    }
    // fragment offsets: A = weight rows wc * 64 + tm * 16 + r; B = halo rows of this wave's 4 x 16 pixels
    const int aoff = frag_off(wc * 64 + r, q);
    int hpb[4], opix[4];                                  // halo row of the window's top-left corner / pixel offset inside the tile's frame block
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
        if (PW == 16) {
            const int py = wpx * 4 + tn, px = r;
            hpb[tn] = py * IW + px; opix[tn] = py * P.W + px;
        } else {
            // lanes r = 4..11 take one image row, r = 0..3 and 12..15 the next: each ds_read_b128 lane group then reads 8 consecutive
            // halo rows per chunk position (distinct mod 8 -> conflict-free), which two rows of a 10-wide halo are not
            const bool lo = (r >= 4) && (r < 12);
            const int py = 2 * tn + (lo ? 0 : 1), px = lo ? r - 4 : (r < 4 ? r : r - 8);
            hpb[tn] = wpx * (IH * IW) + py * IW + px; opix[tn] = wpx * (P.H * P.W) + py * P.W + px;
        }
    }
    float4 bias4[4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
        bias4[tm] = P.bias ? *reinterpret_cast<const float4*>(P.bias + j * 128 + wc * 64 + tm * 16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- helpers ------------------------------------------------------------------------------------------------------
    int pcc = 0, ptap = 0, pslot = 0;                     // weight prefetch cursor (wraps at the end of a tile: the stream repeats)
    auto issue_w = [&]() {
        const char* src = wbase + (size_t)ptap * tap_stride + (size_t)(pcc << 7);
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_a + pslot * WS_SLAB + wave * 1024);
        glds16(src + wsrc[0], dst);
        glds16(src + wsrc[1], dst + 8 * 1024);
        if (++ptap == 9) { ptap = 0; if (++pcc == nchunks) pcc = 0; }
        if (++pslot == NS) pslot = 0;
    };
    unsigned okmask = 0;                                  // PRO: pieces of the buffer in flight that hold image pixels
    auto issue_halo = [&](int t, int cc, int buf) {
        const int fg = t / tiles_pf, rem = t - fg * tiles_pf, ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int f0 = fg * NP;
        const bool second = (cc << 6) >= P.C0;           // wave-uniform: which tensor of the concat this K chunk comes from
        const char* xb = reinterpret_cast<const char*>(second ? P.x1 : P.x0);
        const int Cs = second ? P.C1 : P.C0, cb = (cc << 6) - (second ? P.C0 : 0);
        const unsigned dst = __builtin_amdgcn_readfirstlane(halo_a + buf * HBUF + wave * 1024);
        okmask = 0;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int hi = hinfo[u];
            if (hi >= 0) {
                const int gy = ty * PH - 1 + (hi & 255), gx = tx * PW - 1 + ((hi >> 8) & 255), f = f0 + ((hi >> 16) & 255);
                const bool ok = gy >= 0 && gy < P.H && gx >= 0 && gx < P.W && f < P.NF;
                const size_t off = ((size_t)((f * P.H + gy) * P.W + gx) * Cs + cb + ((hi >> 24) << 3)) * 2;
                const void* src = ok ? static_cast<const void*>(xb + off) : static_cast<const void*>(g_zero_page);
                glds16(src, dst + u * 8 * 1024);
                okmask |= ok ? (1u << u) : 0u;
            }
        }
    };
    // PRO: x_hat = SiLU(x * a[c] + d[c]) in place on this thread's piece u of halo buffer `buf` (zero padding stays zero)
    auto transform = [&](int u, int cc, int buf) {
        const int hi = hinfo[u];
        if (hi < 0 || !((okmask >> u) & 1u)) return;
        const int i = (u * 8 + wave) * 64 + lane;
        char* p = halo + buf * HBUF + i * 16;
        const int c = (cc << 6) + ((hi >> 24) << 3);
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        const float4 a0 = *reinterpret_cast<const float4*>(coefA + c), a1 = *reinterpret_cast<const float4*>(coefA + c + 4);
        const float4 d0 = *reinterpret_cast<const float4*>(coefD + c), d1 = *reinterpret_cast<const float4*>(coefD + c + 4);
        uint4 o;
        o.x = pack_bf16x2(silu_f(fmaf(__uint_as_float(v.x << 16), a0.x, d0.x)), silu_f(fmaf(__uint_as_float(v.x & 0xFFFF0000u), a0.y, d0.y)));
        o.y = pack_bf16x2(silu_f(fmaf(__uint_as_float(v.y << 16), a0.z, d0.z)), silu_f(fmaf(__uint_as_float(v.y & 0xFFFF0000u), a0.w, d0.w)));
        o.z = pack_bf16x2(silu_f(fmaf(__uint_as_float(v.z << 16), a1.x, d1.x)), silu_f(fmaf(__uint_as_float(v.z & 0xFFFF0000u), a1.y, d1.y)));
        o.w = pack_bf16x2(silu_f(fmaf(__uint_as_float(v.w << 16), a1.z, d1.z)), silu_f(fmaf(__uint_as_float(v.w & 0xFFFF0000u), a1.w, d1.w)));
        *reinterpret_cast<uint4*>(p) = o;
    };
    // PRO: per-channel GroupNorm-apply (+ time scale/shift) coefficients of sample b (every thread calls; ends with a barrier)
    auto make_coef = [&](int b) {
        gn_mean_rstd_wg(P.in_stats, b, P.groups, (double)P.F * P.H * P.W * (Cin / P.groups), gmean, tid, 512);
        __syncthreads();
        const int cpg = Cin / P.groups;
        for (int c = tid; c < Cin; c += 512) {
            const int g = c / cpg;
            const float m = gmean[2 * g], rs = gmean[2 * g + 1];
            const float ga = P.gamma[c], be = P.beta[c];
            float sc = 1.f, sh = 0.f;
            if (P.ss) { sc = P.ss[(size_t)b * P.ss_stride + c] + 1.f; sh = P.ss[(size_t)b * P.ss_stride + Cin + c]; }
            coefA[c] = rs * ga * sc;
            coefD[c] = (be - m * rs * ga) * sc + sh;
        }
        __syncthreads();
    };
    // GroupNorm partial sums of the output: a wave's (tm) tile = 16 channels of ONE group; kept in registers across tiles of a sample
    float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};
    auto flush_stats = [&](int b) {
        if (!P.out_stats) return;
        const int cpg = P.Cout / P.out_groups;
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            const float s1 = reduce_q(reduce16(st_s[tm])), s2 = reduce_q(reduce16(st_q[tm]));
            st_s[tm] = 0.f; st_q[tm] = 0.f;
            if (lane == 0) {
                const int g = (j * 128 + wc * 64 + tm * 16) / cpg;
                double* dst = P.out_stats + (((size_t)b * GN_SLOTS + ((blockIdx.x + wave) % GN_SLOTS)) * P.out_groups + g) * 2;
                unsafeAtomicAdd(dst, (double)s1); unsafeAtomicAdd(dst + 1, (double)s2);
            }
        }
    };

    // ---- pipeline prologue: NS - 1 weight slabs and the first halo buffer ----------------------------------------------
    int bcoef = -1;
    const int b0 = (t0 / tiles_pf) * NP / P.F;
    if (PRO) { make_coef(b0); bcoef = b0; }
#pragma unroll
    for (int k = 0; k < NS - 1; ++k) issue_w();
    issue_halo(t0, 0, 0);
    wait_vm_lgkm0<0>();
    __builtin_amdgcn_s_barrier();
    if (PRO) {
#pragma unroll
        for (int u = 0; u < NU; ++u) transform(u, 0, 0);
        wait_vm_lgkm0<0>();
        __builtin_amdgcn_s_barrier();
    }

    int cslot = 0, hbuf = 0, bcur = b0;
    f32x4 acc[4][4];
    // one K chunk = 9 taps on halo buffer hbuf; (tn_, ccn_) = the halo to fetch meanwhile.  AFTER_EPI: the 16 stores of the previous
    // tile's epilogue sit between the weight slabs in flight, so the first three syncs leave that many more operations outstanding.
    auto run_chunk = [&](auto after_epi, int cc, int tnext, int ccnext, bool xform) {
        constexpr int EPI = decltype(after_epi)::value ? WS_STORES : 0;
        int hp[4];                                       // opaque copies: the 36 per-tap fragment offsets are recomputed (3 VALU each)
#pragma unroll                                           // instead of being hoisted out of the tile loop into 36 live registers
        for (int tn = 0; tn < 4; ++tn) { hp[tn] = hpb[tn]; asm volatile("" : "+v"(hp[tn])); }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // sync: slab (cc, tap) has landed in every wave's view; every wave is done with slab (cc, tap - 1) and, at tap 0, with the other halo buffer
            if (tap == 0) wait_vm_lgkm0<WIN + EPI>();
            else if (tap <= 2) wait_vm_lgkm0<WIN + NUMIN + EPI>();
            else if (tap == 3) wait_vm_lgkm0<WIN + NUMIN>();
            else wait_vm_lgkm0<WIN>();
            __builtin_amdgcn_s_barrier();
            issue_w();
            if (tap == 0) issue_halo(tnext, ccnext, hbuf ^ 1);
            if (PRO && xform) {                          // own pieces, landed since the sync of tap 4; spread over the remaining taps
                if (tap == 5) { transform(0, ccnext, hbuf ^ 1); transform(1, ccnext, hbuf ^ 1); }
                if (tap == 6) { transform(2, ccnext, hbuf ^ 1); transform(3, ccnext, hbuf ^ 1); }
                if (tap == 7) {
#pragma unroll
                    for (int u = 4; u < NU; ++u) transform(u, ccnext, hbuf ^ 1);
                }
            }
            const int dy = tap / 3, dx = tap - 3 * dy;
            const char* ws = ring + cslot * WS_SLAB;
            const char* hb = halo + hbuf * HBUF;
            int boff[4];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) boff[tn] = frag_off(hp[tn] + dy * IW + dx, q);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 af[4], bf[4];
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) af[tm] = *reinterpret_cast<const uint4*>(ws + ((aoff + tm * 2048) ^ (ks * 64)));
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) bf[tn] = *reinterpret_cast<const uint4*>(hb + (boff[tn] ^ (ks * 64)));
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) M::mma(acc[tm][tn], af[tm], bf[tn]);
            }
            if (++cslot == NS) cslot = 0;
        }
        hbuf ^= 1;
    };

    bool after_epilogue = false;
    for (int t = t0; t < t1; ++t) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int cc = 0; cc < nchunks; ++cc) {
            // the halo to fetch during this chunk (the very last fetch re-reads a valid tile into the idle buffer: the counted
            // waits assume the same instruction sequence in every chunk)
            const bool last = (cc + 1 == nchunks);
            const bool more = !last || (t + 1 < t1);
            const int tnext = last ? (t + 1 < t1 ? t + 1 : t) : t, ccnext = last ? 0 : cc + 1;
            if (PRO && more) {
                const int bn = (tnext / tiles_pf) * NP / P.F;
                if (bn != bcoef) { make_coef(bn); bcoef = bn; }      // uniform; the tables are only read by transform() below
            }
            if (after_epilogue) run_chunk(std::true_type{}, cc, tnext, ccnext, more);
            else run_chunk(std::false_type{}, cc, tnext, ccnext, more);
            after_epilogue = false;
        }
        // ---- epilogue of tile t: +bias, store, statistics --------------------------------------------------------------
        {
            const int fg = t / tiles_pf, rem = t - fg * tiles_pf, ty = rem / tiles_x, tx = rem - ty * tiles_x;
            const int b = fg * NP / P.F;
            if (b != bcur) { flush_stats(bcur); bcur = b; }
            const size_t tile_pix = ((size_t)fg * NP * P.H + (size_t)ty * PH) * P.W + (size_t)tx * PW;
            const int cobase = j * 128 + wc * 64 + 4 * q;
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                float s = 0.f, ss = 0.f;
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    const float4 v = make_float4(acc[tm][tn][0] + bias4[tm].x, acc[tm][tn][1] + bias4[tm].y,
                                                 acc[tm][tn][2] + bias4[tm].z, acc[tm][tn][3] + bias4[tm].w);
                    const size_t e = (tile_pix + opix[tn]) * P.Cout + cobase + tm * 16;
                    if (P.y_bf16) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(P.y) + e * 2) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                    else *reinterpret_cast<float4*>(P.y + e) = v;
                    s += (v.x + v.y) + (v.z + v.w);
                    ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                }
                st_s[tm] += s; st_q[tm] += ss;
            }
            after_epilogue = true;
        }
    }
    flush_stats(bcur);
    wait_vm_lgkm0<0>();                                   // the prefetches issued past the end must land before the LDS is released
}

// ---- host side ----------------------------------------------------------------------------------------------------------

bool conv3x3_ws_eligible(int mode, const ConvArgs& a) {
    if (mode != MODE_BF16 || a.kind != 0 || a.kh != 3 || a.kw != 3 || a.stride != 1 || a.res) return false;
    if (!a.x0_bf16 || (a.C1 && !a.x1_bf16)) return false;
    if (a.C0 % 64 || a.C1 % 64 || a.Cout % 128) return false;
    const int nct = a.Cout / 128;
    if (nct != 1 && nct != 2 && nct != 4 && nct != 8) return false;
    if (a.wrows != a.Cout || a.wrow0 != 0) return false;
    const bool sq16 = (a.H % 16 == 0) && (a.W % 16 == 0);
    const bool sq8 = (a.H == 8 && a.W == 8 && a.NF % 4 == 0 && a.F % 4 == 0);
    if (!sq16 && !sq8) return false;
    if (a.pro && (a.C1 || a.groups < 1 || a.groups > 32 || a.C0 % a.groups || a.C0 > 1024)) return false;
    if (a.out_stats && (a.out_groups < 1 || a.Cout % a.out_groups || (a.Cout / a.out_groups) % 16)) return false;
    const long tiles = sq16 ? (long)a.NF * (a.H / 16) * (a.W / 16) : a.NF / 4;
    if (tiles * nct < 128) return false;                 // too few tiles to feed the chip from persistent workgroups: generic kernel
    if ((size_t)a.NF * a.H * a.W * (size_t)std::max(a.C0, a.C1) * 2 >= (1ull << 40)) return false;
    return true;
}

hipError_t launch_conv3x3_ws(const ConvArgs& a, hipStream_t st) {
    const bool sq16 = (a.H % 16 == 0) && (a.W % 16 == 0);
    const int nct = a.Cout / 128;
    const int total = sq16 ? a.NF * (a.H / 16) * (a.W / 16) : a.NF / 4;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    const int unit = 8 * nct;                             // the decode deals ranges to the 8 XCDs
    int grid = std::max(unit, cus / unit * unit);
    grid = std::min(grid, (total * nct + unit - 1) / unit * unit);
    const int nranges = grid / nct;
    const int tpr = (total + nranges - 1) / nranges;
    const int NS = sq16 ? 4 : 3;
    const int HPX = sq16 ? 18 * 18 : 4 * 10 * 10;
    const size_t lds = (size_t)NS * WS_SLAB + 2 * (size_t)HPX * 128 + (a.pro ? (size_t)a.CinPad * 8 + 256 : 0);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto go = [&](auto kfn) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, st, a, tpr, total, nct);
        return hipGetLastError();
    };
    if (sq16) return a.pro ? go(conv3x3_ws_kernel<16, 4, true>) : go(conv3x3_ws_kernel<16, 4, false>);
    return a.pro ? go(conv3x3_ws_kernel<8, 3, true>) : go(conv3x3_ws_kernel<8, 3, false>);
}

}  // namespace vdx
