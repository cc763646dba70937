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
//   * INPUT DOUBLE BUFFER: the 64-channel input tile of the NEXT K chunk / pixel tile is fetched by LDS-DMA while the 9 taps of the
//     current one run.  Two geometries: 32 x 32 and larger frames are cut in 16 x 16 tiles with an 18 x 18 halo (out-of-image pieces
//     read a zero page, so the window needs no bounds logic); 8 x 8 and 16 x 16 frames are taken WHOLE (4 / 1 frames = 256 pixels =
//     256 consecutive rows of the tensor) plus ONE zero row that every out-of-frame tap of every lane points at -- no border rows
//     to fetch, to transform or to store.  With a prologue the issuing thread rewrites its own pieces in place (LDS -> registers
//     -> LDS), one piece per tap, between the MFMAs.
//   * FRAGMENTS ONE STEP AHEAD: a tap's sync guarantees the NEXT tap's slab, so the fragments of the next K step (the next tap's
//     first one included) are loaded while the current 16 MFMAs run; the barrier is followed by MFMAs, not by LDS latency.
//   * WAVE TILE 64 couts x 64 pixels (4 x 4 MFMA tiles of 16x16x32): 8 fragment reads per 16 MFMAs instead of 6 per 8.
//   * XCD-aware decode: the workgroups that share a pixel range (one per 128-channel output tile) are 8 ids apart -> same XCD,
//     same L2, so an input tile leaves HBM once.
//
// LDS rows are 128 bytes (64 bf16 of K); 16-byte chunk k of row r sits at position k ^ (r & 7) (conflict-free ds_read_b128 for the
// lane groups of 16 consecutive rows: MI355X_MICROARCH.md, LDS).  LDS-DMA writes lane-linear (base + 16 * lane), so the swizzle is
// applied to the SOURCE address of each lane and again when reading (cdna_hip_programming.md rule 21).
#include "vdx_common.h"
#include "vdx_internal.h"
#include "vdx_glds.h"
#include <type_traits>

namespace vdx {

namespace {

__device__ __attribute__((aligned(16))) unsigned g_zero_page[4];      // source of every out-of-image halo piece

// byte offset of 16-byte chunk (4 ks + q) of LDS row `row` whose swizzle key is `key` (chunk k sits at position k ^ (key & 7)), for
// ks = 0; ks = 1 is this ^ 64
__device__ __forceinline__ int frag_off(int row, int key, int q) { return row * 128 + (((q ^ (key & 3)) | (key & 4)) << 4); }

constexpr int WS_SLAB = 128 * 128;            // one (tap, K chunk) weight slab: 128 couts x 64 cin bf16
constexpr int WS_STORES = 16;                 // global stores per wave in the tile epilogue (4 x 4 MFMA tiles, unconditional)

}  // namespace

// GEO: 0 = 16 x 16 tiles of larger frames with an 18 x 18 halo; 8 / 16 = whole frames of that size (see the header).
constexpr int WS_NS = 4;                      // ring depth: slab t (MFMAs), t + 1 (landed, fragments being prefetched), t + 2 (in flight), + the one being refilled

template <int GEO> struct WsGeo {
    static constexpr bool WF = GEO != 0;
    static constexpr int S = WF ? GEO : 16;
    static constexpr int NP = WF ? 256 / (S * S) : 1;                 // frames per tile
    static constexpr int IW = 18;                                      // halo geometry (GEO 0 only)
    static constexpr int HPX = WF ? 257 : 18 * 18;                     // rows of one input buffer
    static constexpr int NPIECE = HPX * 8;
    static constexpr int NU = (NPIECE + 511) / 512;
    static constexpr int NUMIN = NPIECE / 512;                         // input LDS-DMA instructions EVERY wave issues per buffer
    static constexpr int HBUF = HPX * 128;
};

template <int GEO, bool PRO>
__global__ __launch_bounds__(512, 2) void conv3x3_ws_kernel(const ConvArgs P, const int tiles_per_range, const int total_tiles, const int nct, const int pin) {
    using M = Mma<MODE_BF16>;
    using G = WsGeo<GEO>;
    constexpr bool WF = G::WF;
    constexpr int S = G::S, NP = G::NP, IW = G::IW, HPX = G::HPX, NPIECE = G::NPIECE, NU = G::NU, NUMIN = G::NUMIN, HBUF = G::HBUF;
    constexpr int NS = WS_NS;
    constexpr int WIN = 2;                                // weight LDS-DMA instructions younger than the slab a sync waits for

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;                                    // [NS][128 rows][128 B]
    char* halo = ring + NS * WS_SLAB;                     // [2][HPX rows][128 B]
    float* biasl = reinterpret_cast<float*>(halo + 2 * HBUF);   // [128] bias of this output-channel tile
    float* coefA = biasl + 128;                           // PRO: [Cin] x_hat = x * a + d
    float* coefD = coefA + P.CinPad;
    float* gmean = coefD + P.CinPad;                      // [32][mean, rstd]

    const unsigned ring_a = lds_addr(ring), halo_a = lds_addr(halo);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int wc = wave & 1, wpx = wave >> 1;
    // ids i, i + 8, ... share an XCD: the nct output-channel tiles of one pixel range are consecutive slots of one XCD
    // pin (an experiment, off: VDX_WS_PIN_MB): output-channel tile j lives on the 8 / nct XCDs with xcd % nct == j, so that weight tiles which
    // together exceed an XCD's 4 MB L2 (512 -> 512: 4 x 1.18 MB) stay L2-resident, at the price of fetching every input tile once per XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int j = pin ? xcd % nct : slot % nct;
    const int range = pin ? slot * (8 / nct) + xcd / nct : (slot / nct) * 8 + xcd;
    const int t0 = range * tiles_per_range, t1 = min(t0 + tiles_per_range, total_tiles);
    if (t0 >= t1) return;
    const int tiles_x = WF ? 1 : P.W / 16, tiles_pf = WF ? 1 : tiles_x * (P.H / 16);     // GEO 0: tiles per frame
    const int nchunks = P.CinPad >> 6;
    const int Cin = P.C0 + P.C1;
    const int fpt = NP;                                   // frames per tile
    // whole-frame geometries tile PER SAMPLE: tps = ceil(F / NP) tiles, the last one of a sample only partly filled when NP does not divide
    // F (F = 10, the YAML-literal config_v2_2, at 8 x 8: tiles of 4, 4, 2 frames) -- its missing pixel rows read the zero page, are
    // not stored and stay out of the statistics; a tile never straddles two samples (per-sample prologue coefficients and statistics)
    const int tps = WF ? (P.F + NP - 1) / NP : 1;
    auto sample_of = [&](int t) { return WF ? t / tps : (t / tiles_pf) / P.F; };
    auto wf_pix0 = [&](int t) { const int b = t / tps, k = t - b * tps; return (unsigned)(b * P.F + k * NP) * (unsigned)(S * S); };   // first pixel row of tile t
    auto wf_rows = [&](int t) { const int k = t % tps; return min(NP, P.F - k * NP) * (S * S); };                                  // pixel rows the tile holds

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
    // input pieces: piece i = (u * 8 + wave) * 64 + lane -> LDS row i >> 3, position i & 7, which holds source chunk pos ^ key(row).
    // The key is the row's COLUMN in the tile (halo: ix; whole frames: row, a multiple of 8 per image line), so that the three dx of a
    // tap row share... each dx has one key for every dy: a tap's fragment address = a per-(pixel tile, dx) register + an immediate.
    // Recomputed from the lane id where needed instead of held in registers (`l` = an OPAQUE copy of the lane id: without it the
    // compiler hoists all of this out of the tile loop and spills it).
    auto piece = [&](int u, int l, int& row, int& chunk) -> bool {
        const int i = (u * 8 + wave) * 64 + l;
        row = i >> 3;
        const int key = WF ? row : row - ((row * 3641) >> 16) * IW;           // halo: ix = row % 18 (row < 324)
        chunk = (i & 7) ^ (key & 7);
        return i < NPIECE;
    };
    auto opaque_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    const char* const zero_page = reinterpret_cast<const char*>(g_zero_page);
    // fragment addresses (byte offsets into smem).  A: weight row wc * 64 + tm * 16 + r of the slab (+ tm * 2048, + slab base).
    // B: b3[tn][dx] = chunk q of the input row under tap column dx of pixel (tn, r) for tap row 0 in buffer 0; tap row dy adds the
    // IMMEDIATE dy * DYB, K step 1 is ^ 64, the other buffer +- HBUF (applied to the registers once per chunk).  Whole frames: a tap
    // that leaves the frame reads the zero row instead (zs[dy] = its address minus the immediate; one v_cndmask per fragment).
    constexpr int DYB = (WF ? S : IW) * 128;
    const int halo_o = NS * WS_SLAB;                      // byte offset of the input buffers in smem
    const int aoff = frag_off(wc * 64 + r, r, q);
    int b3[4][3], zs[3];
    int opix[4];                                          // pixel offset inside the tile
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
        if (!WF) {
            const int py = wpx * 4 + tn, px = r;
            opix[tn] = py * P.W + px;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) b3[tn][dx] = halo_o + frag_off(py * IW + px + dx, px + dx, q);
        } else {
            const int p = wpx * 64 + tn * 16 + r;
            opix[tn] = p;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) b3[tn][dx] = halo_o + frag_off(p - S - 1 + dx, p - 1 + dx, q);      // (S % 8 == 0: the key does not depend on dy)
        }
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) zs[dy] = halo_o + 256 * 128 + (q << 4) - dy * DYB;
    // whole frames: is the source pixel of tap (dy, dx) of pixel (tn, r) inside the frame?  (lane masks / uniform conditions the compiler keeps in SGPRs)
    auto tap_valid = [&](int tn, int dy, int dx) -> bool {
        const int p = wpx * 64 + tn * 16 + r, y = (p / S) % S + dy - 1, x = p % S + dx - 1;
        return y >= 0 && y < S && x >= 0 && x < S;
    };
    if (tid < 128) biasl[tid] = P.bias ? P.bias[j * 128 + tid] : 0.f;      // visible after the prologue's barriers

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
        const bool second = (cc << 6) >= P.C0;           // wave-uniform: which tensor of the concat this K chunk comes from
        const char* xb = reinterpret_cast<const char*>(second ? P.x1 : P.x0);
        const int Cs = second ? P.C1 : P.C0, cb = (cc << 6) - (second ? P.C0 : 0);
        const unsigned dst = __builtin_amdgcn_readfirstlane(halo_a + buf * HBUF + wave * 1024);
        okmask = 0;
        const int l = opaque_lane();
        if (WF) {
            const unsigned pix0 = wf_pix0(t);             // a tile = up to 256 consecutive pixel rows of the tensor (tensors < 4 GB: launcher)
            const int nrows = wf_rows(t);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                int row, ch;
                if (piece(u, l, row, ch)) {
                    const bool ok = row < nrows;          // row 256 = the zero row; rows past the sample's last frame read zeros too
                    const unsigned off = ((pix0 + row) * Cs + cb + (ch << 3)) * 2u;
                    const void* src = ok ? static_cast<const void*>(xb + off) : static_cast<const void*>(zero_page);
                    glds16(src, dst + u * 8 * 1024);
                    okmask |= ok ? (1u << u) : 0u;
                }
            }
        } else {
            const int f = t / tiles_pf, rem = t - f * tiles_pf, ty = rem / tiles_x, tx = rem - ty * tiles_x;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                int row, ch;
                if (piece(u, l, row, ch)) {
                    const int iy = (row * 3641) >> 16, ix = row - iy * IW;        // row / 18 for row < 324
                    const int gy = ty * 16 - 1 + iy, gx = tx * 16 - 1 + ix;
                    const bool ok = gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
                    const unsigned off = ((unsigned)((f * P.H + gy) * P.W + gx) * Cs + cb + (ch << 3)) * 2u;
                    const void* src = ok ? static_cast<const void*>(xb + off) : static_cast<const void*>(zero_page);
                    glds16(src, dst + u * 8 * 1024);
                    okmask |= ok ? (1u << u) : 0u;
                }
            }
        }
    };
    // PRO: x_hat = SiLU(x * a[c] + d[c]) in place on this thread's piece u of input buffer `buf` (zero padding stays zero)
    auto transform = [&](int u, int cc, int buf) {
        int row, ch;
        const int l = opaque_lane();
        if (!piece(u, l, row, ch) || !((okmask >> u) & 1u)) return;
        char* p = halo + buf * HBUF + ((u * 8 + wave) * 64 + l) * 16;
        const int c = (cc << 6) + (ch << 3);
        // two halves of 4 channels, fenced, so that at most one half's operands (2 + 8 registers) are live beside the accumulators
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint2 v = *reinterpret_cast<const uint2*>(p + 8 * h);
            const float4 a = *reinterpret_cast<const float4*>(coefA + c + 4 * h);
            const float4 d = *reinterpret_cast<const float4*>(coefD + c + 4 * h);
            uint2 o;
            o.x = pack_bf16x2(silu_f(fmaf(__uint_as_float(v.x << 16), a.x, d.x)), silu_f(fmaf(__uint_as_float(v.x & 0xFFFF0000u), a.y, d.y)));
            o.y = pack_bf16x2(silu_f(fmaf(__uint_as_float(v.y << 16), a.z, d.z)), silu_f(fmaf(__uint_as_float(v.y & 0xFFFF0000u), a.w, d.w)));
            *reinterpret_cast<uint2*>(p + 8 * h) = o;
            asm volatile("" ::: "memory");
        }
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
    // fragment loads
    auto ldA1 = [&](int sl, int tm, int ks) -> uint4 {
        return *reinterpret_cast<const uint4*>(ring + sl * WS_SLAB + ((aoff + tm * 2048) ^ (ks * 64)));
    };
    auto ldB = [&](uint4 (&b)[4], int tap, int ks) {       // from the buffer b3 / zs currently point at
        const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            int a = b3[tn][dx];
            if (WF) a = tap_valid(tn, dy, dx) ? a : zs[dy];
            b[tn] = *reinterpret_cast<const uint4*>(smem + ((a ^ (ks * 64)) + dy * DYB));
        }
    };
    auto flip_buffers = [&](int to_buf1) {                // point b3 / zs at the other input buffer
        const int d = to_buf1 ? HBUF : -HBUF;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) b3[tn][dx] += d;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) zs[dy] += d;
    };

    // ---- pipeline prologue: NS - 1 weight slabs and the first input buffer ---------------------------------------------
    int bcoef = -1;
    const int b0 = sample_of(t0);
    if (PRO) { make_coef(b0); bcoef = b0; }
    issue_halo(t0, 0, 0);
#pragma unroll
    for (int k = 0; k < NS - 1; ++k) issue_w();
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
    // fragments loaded one tap ahead (the sync of a tap guarantees the NEXT tap's slab): K step 0 of the 4 pixel tiles + the first
    // weight tile, so the barrier is followed by MFMAs, not by LDS latency.  (Holding both K steps of both operands in registers,
    // 64 of them, was measured slower: r02, spills.)
    uint4 pb[4], pa;
    ldB(pb, 0, 0); pa = ldA1(0, 0, 0);
    // one K chunk = 9 taps on input buffer hbuf; (tnext, ccnext) = the buffer to fetch meanwhile.  AFTER_EPI: the 16 stores of the
    // previous tile's epilogue sit between the weight slabs in flight, so the first two syncs leave that many more operations outstanding.
    auto run_chunk = [&](auto after_epi, int tnext, int ccnext, bool xform) {
        constexpr int EPI = decltype(after_epi)::value ? WS_STORES : 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // sync of tap: every wave's part of slab (tap + 1) has landed (issued two taps ago; at most the younger slab, and at taps
            // 0 / 1 the input pieces and the epilogue stores, may still be in flight), every wave is done with slab (tap - 1) and, at
            // tap 0, with the other input buffer; at tap 2 the input pieces issued at tap 0 (older than slab 3) have landed too
            if (tap == 0) wait_vm<WIN + EPI>();
            else if (tap == 1) wait_vm<WIN + NUMIN + EPI>();
            else wait_vm<WIN>();
            __builtin_amdgcn_s_barrier();
            if (tap == 0) issue_halo(tnext, ccnext, hbuf ^ 1);
            issue_w();
            // K step 0 on the prefetched fragments (MFMAs right behind the barrier), then K step 1, then the next tap's prefetch.
            // Prologue work (own pieces, landed since the sync of tap 2: one per tap, done by tap 7) is STAGGERED between the two waves
            // of a SIMD (w and w + 4): the first half transforms before its K step 0, the second half between the K steps, so one
            // wave's VALU block runs under the other's MFMAs instead of both issuing VALU, then both MFMA, in lockstep behind the barrier
            const bool do_x = PRO && xform && tap >= 2 && tap - 2 < NU;
            if (do_x && wave < 4) {
                transform(tap - 2, ccnext, hbuf ^ 1);
                if (NU > 6 && tap == 7) transform(6, ccnext, hbuf ^ 1);
            }
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                const uint4 a = tm == 0 ? pa : ldA1(cslot, tm, 0);
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) M::mma(acc[tm][tn], a, pb[tn]);
            }
            if (do_x && wave >= 4) {
                transform(tap - 2, ccnext, hbuf ^ 1);
                if (NU > 6 && tap == 7) transform(6, ccnext, hbuf ^ 1);
            }
            {
                uint4 b1[4];
                ldB(b1, tap, 1);
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) {
                    const uint4 a = ldA1(cslot, tm, 1);
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) M::mma(acc[tm][tn], a, b1[tn]);
                }
            }
            const int nslot = (cslot + 1 == NS) ? 0 : cslot + 1;
            if (tap == 8) flip_buffers(hbuf ^ 1);             // the next chunk's buffer (landed, transformed, made visible by this tap's sync)
            ldB(pb, tap == 8 ? 0 : tap + 1, 0);
            pa = ldA1(nslot, 0, 0);
            cslot = nslot;
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
            // the buffer to fetch during this chunk (the very last fetch re-reads a valid tile into the idle buffer: the counted
            // waits assume the same instruction sequence in every chunk)
            const bool last = (cc + 1 == nchunks);
            const bool more = !last || (t + 1 < t1);
            const int tnext = last ? (t + 1 < t1 ? t + 1 : t) : t, ccnext = last ? 0 : cc + 1;
            if (PRO && more) {
                const int bn = sample_of(tnext);
                if (bn != bcoef) { make_coef(bn); bcoef = bn; }      // uniform; the tables are only read by transform() below
            }
            if (after_epilogue) run_chunk(std::true_type{}, tnext, ccnext, more);
            else run_chunk(std::false_type{}, tnext, ccnext, more);
            after_epilogue = false;
        }
        // ---- epilogue of tile t: +bias, store, statistics --------------------------------------------------------------
        {
            const int b = sample_of(t);
            if (b != bcur) { flush_stats(bcur); bcur = b; }
            size_t tile_pix;
            int nrows = 256;
            if (WF) { tile_pix = wf_pix0(t); nrows = wf_rows(t); }
            else { const int f = t / tiles_pf, rem = t - f * tiles_pf, ty = rem / tiles_x, tx = rem - ty * tiles_x;
                   tile_pix = ((size_t)f * P.H + (size_t)ty * 16) * P.W + (size_t)tx * 16; }
            const int cobase = j * 128 + wc * 64 + 4 * q;
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                float s = 0.f, ss = 0.f;
                const float4 bs = *reinterpret_cast<const float4*>(biasl + wc * 64 + tm * 16 + 4 * q);
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    if (WF && opix[tn] >= nrows) continue;                 // (a partly filled last tile of a sample)
                    const float4 v = make_float4(acc[tm][tn][0] + bs.x, acc[tm][tn][1] + bs.y, acc[tm][tn][2] + bs.z, acc[tm][tn][3] + bs.w);
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

// ---- the 4x4 / stride-2 resampling convs on the same machinery (round 2) ------------------------------------------------------
// Downsample (reference utils.py:115-125: 4x4, stride 2, SAME) and Upsample (utils.py:103-113: ConvTranspose 4x4, stride 2) of the wide
// levels ran on the generic kernel at 0.14-0.30 of the MFMA peak (16 taps of 16 MFMAs per wave between two barriers).  Both are a
// 2 x 2-tap convolution per (phase, 64-channel chunk) on a 256-pixel whole-frame tile, so they reuse the weight ring, the double-buffered
// LDS-DMA input tile with its zero row, the counted waits and the 64 x 64 wave tiles of conv3x3_ws_kernel:
//   KIND 1, Downsample: GEO = OUTPUT frame size.  Input row 2y + i - 1 = 2(y + a) + p: the input splits in four parity planes (py, px),
//     each a frame of the OUTPUT size; plane p uses the taps a in {-1, 0} (p = 1) or {0, 1} (p = 0) per axis.  K chunk = (plane, 64
//     channels): the tile is GATHERED from the plane's pixels (16-byte pieces stay contiguous), 4 taps (dy, dx) in {ey, ey+1} x {ex, ex+1}
//     of the 3 x 3 fragment addressing with ey = 1 - py, packed tap index (1 - py + 2 ky) * 4 + (1 - px + 2 kx).
//   KIND 2, Upsample: GEO = INPUT frame size.  Output phase (ry, rx) of pixel (y, x) = sum over 2 x 2 taps of input (y + ry - 1 + ky, ...):
//     the four phases of a tile are four consecutive "virtual tiles" on the same input, taps {ry, ry+1} x {rx, rx+1}, packed tap index
//     (2 ky + ry) * 4 + (2 kx + rx), outputs scattered to (2y + ry, 2x + rx).
// No prologue, no statistics (the reference's resampling convs have neither); bias in the epilogue.
// GEO 8 / 16: whole frames of that size (the coarse side of the resampling); GEO 32: BANDS of 8 rows x 32 columns of a 32 x 32 frame
// with one halo row above and below (out-of-frame rows read the zero page, out-of-frame COLUMNS select the zero row) -- the two
// resampling convs of the widest level.  BCO = output channels per workgroup: 128 (wave tile 64 couts x 64 pixels) or 64 (every wave
// takes all 64 couts of 32 pixels; 8 KB slabs, one LDS-DMA per wave per slab).
template <int GEO, int KIND, int BCO>
__global__ __launch_bounds__(512, 2) void conv4x4_ws_kernel(const ConvArgs P, const int tiles_per_range, const int total_vtiles, const int nct) {
    using M = Mma<MODE_BF16>;
    static_assert((GEO == 8 || GEO == 16 || GEO == 32) && (KIND == 1 || KIND == 2) && (BCO == 64 || BCO == 128), "geometry");
    constexpr bool BAND = GEO == 32;
    constexpr int S = GEO;                                 // frame size of the tile's pixel grid
    constexpr int NP = BAND ? 1 : 256 / (S * S);           // frames per tile (whole-frame geometries)
    constexpr int ZROW = BAND ? 320 : 256;                 // the zero row; BAND: rows 0..319 = band rows -1..8 x 32 columns
    constexpr int HPX = ZROW + 1, NPIECE = HPX * 8, NU = (NPIECE + 511) / 512, NUMIN = NPIECE / 512, HBUF = HPX * 128;
    constexpr int NS = WS_NS;
    constexpr int WIN = BCO / 64;                          // LDS-DMA instructions per wave per slab
    constexpr int SLAB = BCO * 128;
    constexpr int TN = BCO == 128 ? 4 : 2;                 // 16-pixel tiles per wave
    constexpr int STORES = 4 * TN;                         // global stores per wave in the tile epilogue

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;                                    // [NS][128 rows][128 B]
    char* halo = ring + NS * SLAB;                        // [2][HPX rows][128 B]
    float* biasl = reinterpret_cast<float*>(halo + 2 * HBUF);
    const unsigned ring_a = lds_addr(ring), halo_a = lds_addr(halo);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int wc = BCO == 128 ? (wave & 1) : 0, wpx = BCO == 128 ? (wave >> 1) : wave;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int j = slot % nct;
    const int range = (slot / nct) * 8 + xcd;
    const int t0 = range * tiles_per_range, t1 = min(t0 + tiles_per_range, total_vtiles);
    if (t0 >= t1) return;
    const int C = P.C0;
    const int ncc = C >> 6;                               // 64-channel chunks of the input
    const int nchunks = KIND == 1 ? 4 * ncc : ncc;        // K chunks per (virtual) tile

    unsigned wsrc[WIN];
#pragma unroll
    for (int v = 0; v < WIN; ++v) {
        const int i2 = (v * 8 + wave) * 64 + lane, row = i2 >> 3, pos = i2 & 7;
        wsrc[v] = (unsigned)((j * BCO + row) * P.CinPad) * 2u + (unsigned)((pos ^ (row & 7)) << 4);
    }
    const size_t tap_stride = (size_t)P.Cout * P.CinPad * 2;
    const char* wbase = reinterpret_cast<const char*>(P.wp);
    auto piece = [&](int u, int l, int& row, int& chunk) -> bool {
        const int i = (u * 8 + wave) * 64 + l;
        row = i >> 3;
        chunk = (i & 7) ^ (row & 7);
        return i < NPIECE;
    };
    auto opaque_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    const char* const zero_page = reinterpret_cast<const char*>(g_zero_page);
    constexpr int DYB = S * 128;
    const int halo_o = NS * SLAB;
    const int aoff = frag_off(wc * 64 + r, r, q);
    int b3[TN][3], zs[3], opix[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int p = wpx * (TN * 16) + tn * 16 + r;       // pixel of the tile: (frame, y, x) row-major, or (band row, x)
        if (KIND == 1) opix[tn] = p;
        else { const int fl = p / (S * S), y = (p / S) % S, x = p % S; opix[tn] = (fl * 2 * S + 2 * y) * 2 * S + 2 * x; }      // phase (0, 0) of the pixel in the tile's output rows
        // LDS row of tap (dy, dx): whole frames p + (dy - 1) S + dx - 1; bands (halo row first) p + dy S + dx - 1.  S % 8 == 0: one key per dx
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) b3[tn][dx] = halo_o + frag_off(p - (BAND ? 0 : S) - 1 + dx, p - 1 + dx, q);
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) zs[dy] = halo_o + ZROW * 128 + (q << 4) - dy * DYB;
    auto tap_valid = [&](int tn, int dy, int dx) -> bool {
        const int p = wpx * (TN * 16) + tn * 16 + r, y = (p / S) % S + dy - 1, x = p % S + dx - 1;
        return (BAND || (y >= 0 && y < S)) && x >= 0 && x < S;          // (bands: the halo rows hold the vertical neighbours or zeros)
    };
    if (tid < BCO) biasl[tid] = P.bias ? P.bias[j * BCO + tid] : 0.f;

    // weight prefetch cursor: (phase of the virtual tile (KIND 2), K chunk, tap) in consumption order; the stream repeats per tile (KIND 1) / per 4 virtual tiles
    int pph = t0 & 3, pcc = 0, ptap = 0, pslot = 0;
    auto issue_w = [&]() {
        const int ky = ptap >> 1, kx = ptap & 1;
        int widx, kofs;
        if (KIND == 1) { const int plane = pcc / ncc, cch = pcc - plane * ncc; widx = (1 - (plane >> 1) + 2 * ky) * 4 + (1 - (plane & 1) + 2 * kx); kofs = cch << 7; }
        else { widx = (2 * ky + (pph >> 1)) * 4 + (2 * kx + (pph & 1)); kofs = pcc << 7; }
        const char* src = wbase + (size_t)widx * tap_stride + (size_t)kofs;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_a + pslot * SLAB + wave * 1024);
#pragma unroll
        for (int v = 0; v < WIN; ++v) glds16(src + wsrc[v], dst + v * 8 * 1024);
        if (++ptap == 4) { ptap = 0; if (++pcc == nchunks) { pcc = 0; pph = (pph + 1) & 3; } }
        if (++pslot == NS) pslot = 0;
    };
    auto issue_halo = [&](int vt, int cc, int buf) {
        const char* xb = reinterpret_cast<const char*>(P.x0);
        const unsigned dst = __builtin_amdgcn_readfirstlane(halo_a + buf * HBUF + wave * 1024);
        const int l = opaque_lane();
        const int plane = KIND == 1 ? cc / ncc : 0, cch = KIND == 1 ? cc - plane * ncc : cc;
        const int py = plane >> 1, px = plane & 1;
        const int tile = KIND == 1 ? vt : (vt >> 2);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            int row, ch;
            if (piece(u, l, row, ch)) {
                bool ok = row < ZROW;
                unsigned pix;
                if (BAND) {
                    const int f = tile >> 2, y = (tile & 3) * 8 + row / S - 1, x = row % S;      // 4 bands per frame; LDS row 0 = the halo row above
                    ok = ok && y >= 0 && y < S;
                    pix = KIND == 1 ? (unsigned)((f * 2 * S + 2 * y + py) * 2 * S + 2 * x + px) : (unsigned)((f * S + y) * S + x);
                } else if (KIND == 1) {
                    const int fl = row / (S * S), y = (row / S) % S, x = row % S;
                    pix = (unsigned)(((tile * NP + fl) * 2 * S + 2 * y + py) * 2 * S + 2 * x + px);
                } else pix = (unsigned)tile * 256u + row;
                const unsigned off = (pix * C + (cch << 6) + (ch << 3)) * 2u;
                const void* src = ok ? static_cast<const void*>(xb + off) : static_cast<const void*>(zero_page);
                glds16(src, dst + u * 8 * 1024);
            }
        }
    };
    auto ldA1 = [&](int sl, int tm, int ks) -> uint4 {
        return *reinterpret_cast<const uint4*>(ring + sl * SLAB + ((aoff + tm * 2048) ^ (ks * 64)));
    };
    auto flip_buffers = [&](int to_buf1) {
        const int d = to_buf1 ? HBUF : -HBUF;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) b3[tn][dx] += d;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) zs[dy] += d;
    };

    issue_halo(t0, 0, 0);
#pragma unroll
    for (int k = 0; k < NS - 1; ++k) issue_w();
    wait_vm_lgkm0<0>();
    __builtin_amdgcn_s_barrier();

    int cslot = 0, hbuf = 0;
    f32x4 acc[4][TN];
    // one K chunk = 4 taps (EY + ky, EX + kx) on input buffer hbuf; (tnext, ccnext) = the buffer to fetch meanwhile (conv3x3_ws_kernel's
    // schedule: the pieces issued at tap 0 are older than the slab tap 2's sync waits for)
    auto run_chunk = [&](auto after_epi, auto ey_, auto ex_, int tnext, int ccnext) {
        constexpr int EPI = decltype(after_epi)::value ? STORES : 0;
        constexpr int EY = decltype(ey_)::value, EX = decltype(ex_)::value;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            constexpr int dummy = 0; (void)dummy;
            const int dy = EY + (k >> 1), dx = EX + (k & 1);                 // (constants after unrolling)
            if (k == 0) wait_vm<WIN + EPI>();
            else if (k == 1) wait_vm<WIN + NUMIN + EPI>();
            else wait_vm<WIN>();
            __builtin_amdgcn_s_barrier();
            if (k == 0) issue_halo(tnext, ccnext, hbuf ^ 1);
            issue_w();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 b[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    int a = b3[tn][dx];
                    a = tap_valid(tn, dy, dx) ? a : zs[dy];
                    b[tn] = *reinterpret_cast<const uint4*>(smem + ((a ^ (ks * 64)) + dy * DYB));
                }
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) {
                    const uint4 a = ldA1(cslot, tm, ks);
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) M::mma(acc[tm][tn], a, b[tn]);
                }
            }
            if (k == 3) flip_buffers(hbuf ^ 1);
            cslot = (cslot + 1 == NS) ? 0 : cslot + 1;
        }
        hbuf ^= 1;
    };
    auto run_parity = [&](auto after_epi, int ey, int ex, int tnext, int ccnext) {
        using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>;
        if (ey == 0) { if (ex == 0) run_chunk(after_epi, T0{}, T0{}, tnext, ccnext); else run_chunk(after_epi, T0{}, T1{}, tnext, ccnext); }
        else { if (ex == 0) run_chunk(after_epi, T1{}, T0{}, tnext, ccnext); else run_chunk(after_epi, T1{}, T1{}, tnext, ccnext); }
    };

    bool after_epilogue = false;
    for (int t = t0; t < t1; ++t) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int cc = 0; cc < nchunks; ++cc) {
            const bool last = (cc + 1 == nchunks);
            const int tnext = last ? (t + 1 < t1 ? t + 1 : t) : t, ccnext = last ? 0 : cc + 1;
            int ey, ex;
            if (KIND == 1) { const int plane = cc / ncc; ey = 1 - (plane >> 1); ex = 1 - (plane & 1); }
            else { ey = (t & 3) >> 1; ex = t & 1; }
            if (after_epilogue) run_parity(std::true_type{}, ey, ex, tnext, ccnext);
            else run_parity(std::false_type{}, ey, ex, tnext, ccnext);
            after_epilogue = false;
        }
        {   // epilogue of (virtual) tile t: + bias, store
            size_t tile_pix;
            if (KIND == 1) tile_pix = (size_t)t * 256;
            else tile_pix = (size_t)(t >> 2) * 1024 + (size_t)(((t & 3) >> 1) * 2 * S + (t & 1));     // 4 output pixels per input pixel; phase offset
            const int cobase = j * BCO + wc * 64 + 4 * q;
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                const float4 bs = *reinterpret_cast<const float4*>(biasl + wc * 64 + tm * 16 + 4 * q);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const float4 v = make_float4(acc[tm][tn][0] + bs.x, acc[tm][tn][1] + bs.y, acc[tm][tn][2] + bs.z, acc[tm][tn][3] + bs.w);
                    const size_t e = (tile_pix + opix[tn]) * P.Cout + cobase + tm * 16;
                    if (P.y_bf16) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(P.y) + e * 2) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                    else *reinterpret_cast<float4*>(P.y + e) = v;
                }
            }
            after_epilogue = true;
        }
    }
    wait_vm_lgkm0<0>();
}

// ---- host side ----------------------------------------------------------------------------------------------------------

static int ws_geo(const ConvArgs& a) {                    // 0: 16 x 16 tiles, 8 / 16: whole frames, -1: not served
    if (a.H == 8 && a.W == 8) return 8;                   // (4 frames per tile; F % 4 != 0: the last tile of a sample is partly filled)
    if (a.H == 16 && a.W == 16) return 16;
    if (a.H % 16 == 0 && a.W % 16 == 0) return 0;
    return -1;
}

int conv3x3_ws_geo(const ConvArgs& a) { return ws_geo(a); }

// frame size of the 256-pixel whole-frame tiles of the 4x4 resampling kernel (output frames of Downsample, input frames of Upsample), or 0
static int ws4_geo(const ConvArgs& a) {
    const bool down = a.kind == 0 && a.kh == 4 && a.kw == 4 && a.stride == 2, up = a.kind == 1;
    if (!down && !up) return 0;
    const int s = down ? a.H / 2 : a.H;
    if (a.H != a.W || (down && (a.H % 2))) return 0;
    if (s == 32) return 32;                               // bands of 8 rows
    if (s == 16) return 16;
    if (s == 8 && a.NF % 4 == 0) return 8;
    return 0;
}

bool conv4x4_ws_eligible(int mode, const ConvArgs& a) {
    if (mode != MODE_BF16 || a.res || a.pro || a.out_stats || a.C1 || !a.x0_bf16) return false;
    const int geo = ws4_geo(a);
    if (!geo) return false;
    if (a.C0 % 64 || (a.Cout % 128 && a.Cout != 64) || a.CinPad != a.C0) return false;
    const int nct = a.Cout == 64 ? 1 : a.Cout / 128;
    if (nct != 1 && nct != 2 && nct != 4 && nct != 8) return false;
    if (a.wrows != a.Cout || a.wrow0 != 0) return false;
    const long vtiles = (long)a.NF * geo * geo / 256 * (a.kind == 1 ? 4 : 1);
    if (vtiles * nct < 128) return false;
    if ((size_t)a.NF * a.H * a.W * (size_t)a.C0 * 2 >= 0xFFFF0000ull || (size_t)a.NF * 4 * geo * geo >= 0x7FFFFFFFull) return false;
    return true;
}

hipError_t launch_conv4x4_ws(const ConvArgs& a, hipStream_t st) {
    const int geo = ws4_geo(a), bco = a.Cout == 64 ? 64 : 128, nct = a.Cout / bco, up = a.kind == 1;
    const int total = (int)((long)a.NF * geo * geo / 256) * (up ? 4 : 1);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    const int unit = 8 * nct;
    int grid = std::max(unit, cus / unit * unit);
    grid = std::min(grid, (total * nct + unit - 1) / unit * unit);
    const int nranges = grid / nct;
    int tpr = (total + nranges - 1) / nranges;
    if (up) tpr = (tpr + 3) / 4 * 4;                      // a range = whole groups of 4 phases (the weight stream's phase follows t & 3)
    const size_t lds = (size_t)WS_NS * bco * 128 + 2 * (size_t)(geo == 32 ? 321 : 257) * 128 + 512;
    auto go = [&](auto kfn) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, st, a, tpr, total, nct);
        return hipGetLastError();
    };
#define VDX_WS4(G_) do { if (bco == 64) return up ? go(conv4x4_ws_kernel<G_, 2, 64>) : go(conv4x4_ws_kernel<G_, 1, 64>);       \
                         return up ? go(conv4x4_ws_kernel<G_, 2, 128>) : go(conv4x4_ws_kernel<G_, 1, 128>); } while (0)
    if (geo == 8) VDX_WS4(8);
    if (geo == 16) VDX_WS4(16);
    VDX_WS4(32);
#undef VDX_WS4
}

bool conv3x3_ws_eligible(int mode, const ConvArgs& a) {
    if (mode != MODE_BF16 || a.kind != 0 || a.kh != 3 || a.kw != 3 || a.stride != 1 || a.res) return false;
    if (!a.x0_bf16 || (a.C1 && !a.x1_bf16)) return false;
    if (a.C0 % 64 || a.C1 % 64 || a.Cout % 128) return false;
    const int nct = a.Cout / 128;
    if (nct != 1 && nct != 2 && nct != 4 && nct != 8) return false;
    if (a.wrows != a.Cout || a.wrow0 != 0) return false;
    const int geo = ws_geo(a);
    if (geo < 0) return false;
    if (geo != 0 && (a.F <= 0 || a.NF % a.F)) return false;        // whole-frame tiles are laid out per sample
    if (a.pro && (a.C1 || a.groups < 1 || a.groups > 32 || a.C0 % a.groups || a.C0 > 1024)) return false;
    if (a.out_stats && (a.out_groups < 1 || a.Cout % a.out_groups || (a.Cout / a.out_groups) % 16)) return false;
    const int npf = geo == 0 ? 1 : 256 / (a.H * a.W);         // frames per whole-frame tile; tiles per sample = ceil(F / npf)
    const long tiles = geo == 0 ? (long)a.NF * (a.H / 16) * (a.W / 16) : (long)(a.NF / a.F) * ((a.F + npf - 1) / npf);
    if (tiles * nct < 128) return false;                 // too few tiles to feed the chip from persistent workgroups: generic kernel
    if ((size_t)a.NF * a.H * a.W * (size_t)std::max(a.C0, a.C1) * 2 >= 0xFFFF0000ull) return false;     // 32-bit byte offsets
    return true;
}

hipError_t launch_conv3x3_ws(const ConvArgs& a, hipStream_t st) {
    const int geo = ws_geo(a);
    const int nct = a.Cout / 128;
    const int npf = geo == 0 ? 1 : 256 / (a.H * a.W);
    const int total = geo == 0 ? a.NF * (a.H / 16) * (a.W / 16) : (a.NF / a.F) * ((a.F + npf - 1) / npf);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    const int unit = 8 * nct;                             // the decode deals ranges to the 8 XCDs
    int grid = std::max(unit, cus / unit * unit);
    grid = std::min(grid, (total * nct + unit - 1) / unit * unit);
    const int nranges = grid / nct;
    const int tpr = (total + nranges - 1) / nranges;
    const int HPX = geo == 0 ? 18 * 18 : 257;
    const size_t lds = (size_t)WS_NS * WS_SLAB + 2 * (size_t)HPX * 128 + 512 + (a.pro ? (size_t)a.CinPad * 8 + 256 : 0);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto go = [&](auto kfn) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
#ifndef VDX_WS_PIN_MB
#define VDX_WS_PIN_MB 0        // pin output-channel tiles to XCDs when their weights together exceed this many MB (0: never).  Measured with 3 (r03, 512 -> 512 at 8 x 8):
                               // same time (251 vs 250 us plain, 307 vs 317 with the prologue) for MORE HBM traffic (401-454 vs 290-326 MB per launch: the input tiles
                               // are then fetched by four XCDs) -- the Infinity Cache serves the weight stream as fast as the L2 does; off
#endif
        const size_t wbytes = (size_t)9 * a.Cout * a.CinPad * 2;
        const int pin = (VDX_WS_PIN_MB > 0 && nct > 1 && wbytes > (size_t)VDX_WS_PIN_MB * 1024 * 1024) ? 1 : 0;
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, st, a, tpr, total, nct, pin);
        return hipGetLastError();
    };
    if (geo == 0) return a.pro ? go(conv3x3_ws_kernel<0, true>) : go(conv3x3_ws_kernel<0, false>);
    if (geo == 8) return a.pro ? go(conv3x3_ws_kernel<8, true>) : go(conv3x3_ws_kernel<8, false>);
    return a.pro ? go(conv3x3_ws_kernel<16, true>) : go(conv3x3_ws_kernel<16, false>);
}

}  // namespace vdx
