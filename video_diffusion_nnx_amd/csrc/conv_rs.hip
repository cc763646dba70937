// Downsample / Upsample of 32-channel tensors (level 0 of dim-32 networks: configs/config_v2_2.yaml as written) on bf16 tensors.
//
// Reference: utils.py:115-125 (Downsample = Conv (1,4,4), stride 2, padding 1) and utils.py:103-113 (Upsample = ConvTranspose (1,4,4),
// stride 2, as Flax applies it: the oracle's conv_transpose_144).  These two launches ran on the generic implicit-GEMM kernel at 0.03-0.05
// of the MFMA peak (250 + 160 us at B = 64 against a memory floor of ~45 us each: 210 MB) -- conv4x4_ws_kernel serves 64 and multiples of
// 128 channels only.  With 32 channels everything a wave needs fits its registers: the whole weight set (16 taps x 2 output-channel tiles
// = 32 A fragments, 128 registers; the tile rows permuted so that a lane ends with 8 consecutive channels = one 16-byte store per pixel)
// and the input rows of 16 pixels straight from global memory / L1 as B fragments (lane (pixel, q) = 16 bytes = channels 8q..8q+7: the
// 32 input channels are ONE K chunk).  No LDS, no barrier; a wave walks a column strip of 16 pixels down the frame, so the rows it shares
// with its previous tile are L1 hits.
//   Downsample: out[oy, ox] = sum_{ky, kx} W[ky][kx] . in[2 oy - 1 + ky, 2 ox - 1 + kx]      (16 fragments -> 32 MFMAs per 16 output pixels)
//   Upsample:   out[2y + ry, 2x + rx] = sum_{a, b in {0,1}} W[2a + ry][2b + rx] . in[y + ry - 1 + a, x + rx - 1 + b]
//               (9 fragments = the 3 x 3 neighbourhood -> 4 phases x 4 taps x 2 tiles = 32 MFMAs per 16 input pixels)
#include "vdx_common.h"
#include "vdx_internal.h"
#include <algorithm>

namespace vdx {

typedef unsigned rs_u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void resample32_kernel(const ConvArgs P, const int tiles_per_wave, const long total_tiles) {
    using M = Mma<MODE_BF16>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lp = lane & 15, q = lane >> 4;
    const long w_id = (long)blockIdx.x * 4 + wave;
    const long t0 = w_id * tiles_per_wave, t1 = std::min<long>(t0 + tiles_per_wave, total_tiles);
    if (t0 >= t1) return;
    // the tile grid: KIND 0 walks OUTPUT pixels (H / 2 x W / 2), KIND 1 INPUT pixels (H x W); strips of 16 pixels in x, y fastest
    const int TH = KIND == 0 ? P.H / 2 : P.H, TW = KIND == 0 ? P.W / 2 : P.W;
    const int strips = TW >> 4;
    // weights: packed [16 taps][32 rows][64 ci (32 used)] bf16 = 128-byte rows; A-tile row i of tile tm = output channel 8 (i >> 2) + 4 tm + (i & 3)
    uint4 wf[16][2];
#pragma unroll
    for (int tap = 0; tap < 16; ++tap)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
            wf[tap][tm] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(P.wp) + ((size_t)tap * 32 + 8 * (lp >> 2) + 4 * tm + (lp & 3)) * 128 + q * 16);
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
    if (P.bias) { b0 = *reinterpret_cast<const float4*>(P.bias + 8 * q); b1 = *reinterpret_cast<const float4*>(P.bias + 8 * q + 4); }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x0), 0, (unsigned)((size_t)P.NF * P.H * P.W * 64), 0x00020000);
    const int OH = KIND == 0 ? P.H / 2 : 2 * P.H, OW = KIND == 0 ? P.W / 2 : 2 * P.W;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(P.y, 0, (unsigned)((size_t)P.NF * OH * OW * 64), 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    auto in_off = [&](int f, int iy, int ix) __attribute__((always_inline)) -> unsigned {       // this lane's 16 bytes of input pixel (iy, ix), or out of range
        const bool ok = iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
        return ok ? (unsigned)(((f * P.H + iy) * P.W + ix) * 64 + q * 16) : OOB;
    };
    auto store8 = [&](unsigned off, const f32x4& a0, const f32x4& a1) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_buffer_store_b128(rs_u32x4{pack_bf16x2(a0[0] + b0.x, a0[1] + b0.y), pack_bf16x2(a0[2] + b0.z, a0[3] + b0.w),
                                                        pack_bf16x2(a1[0] + b1.x, a1[1] + b1.y), pack_bf16x2(a1[2] + b1.z, a1[3] + b1.w)}, ry, off, 0, 0);
    };
    for (long t = t0; t < t1; ++t) {
        const int y = (int)(t % TH);
        const long r1 = t / TH;
        const int sx = (int)(r1 % strips), f = (int)(r1 / strips);
        const int x = sx * 16 + lp;                       // this lane's pixel of the tile row
        if constexpr (KIND == 0) {
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            rs_u32x4 fr[2][8];                                // two halves of 8 taps: the second is in flight during the MFMAs of the first
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int tap = hf * 8 + k, ky = tap >> 2, kx = tap & 3;
                    fr[hf][k] = __builtin_amdgcn_raw_buffer_load_b128(rx, in_off(f, 2 * y - 1 + ky, 2 * x - 1 + kx), 0, 0);
                }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint4 bfr = make_uint4(fr[hf][k].x, fr[hf][k].y, fr[hf][k].z, fr[hf][k].w);
                    M::mma(acc[0], wf[hf * 8 + k][0], bfr);
                    M::mma(acc[1], wf[hf * 8 + k][1], bfr);
                }
            store8((unsigned)(((f * OH + y) * OW + x) * 64 + q * 16), acc[0], acc[1]);
        } else {
            rs_u32x4 fr[3][3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) fr[dy][dx] = __builtin_amdgcn_raw_buffer_load_b128(rx, in_off(f, y - 1 + dy, x - 1 + dx), 0, 0);
#pragma unroll
            for (int ry_ = 0; ry_ < 2; ++ry_)
#pragma unroll
                for (int rx_ = 0; rx_ < 2; ++rx_) {
                    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const rs_u32x4 v = fr[ry_ + a][rx_ + b];              // input (y + ry - 1 + a, x + rx - 1 + b)
                            const uint4 bfr = make_uint4(v.x, v.y, v.z, v.w);
                            const int tap = (2 * a + ry_) * 4 + (2 * b + rx_);
                            M::mma(acc[0], wf[tap][0], bfr);
                            M::mma(acc[1], wf[tap][1], bfr);
                        }
                    store8((unsigned)(((f * OH + 2 * y + ry_) * OW + 2 * x + rx_) * 64 + q * 16), acc[0], acc[1]);
                }
        }
    }
}

bool resample32_eligible(int mode, const ConvArgs& a) {
    if (mode != MODE_BF16 || a.C0 != 32 || a.C1 != 0 || a.Cout != 32 || !a.x0_bf16 || !a.y_bf16 || a.res || a.pro || a.out_stats) return false;
    if (a.wrows != 32 || a.wrow0 != 0 || a.CinPad != 64) return false;
    const bool down = a.kind == 0 && a.kh == 4 && a.kw == 4 && a.stride == 2 && a.pad == 1 && a.H % 2 == 0 && a.W % 32 == 0;
    const bool up = a.kind == 1 && a.W % 16 == 0;
    if (!down && !up) return false;
    const size_t in_b = (size_t)a.NF * a.H * a.W * 64, out_b = down ? in_b / 4 : in_b * 4;
    return in_b < 0xFFFFFFF0ull && out_b < 0xFFFFFFF0ull;
}

hipError_t launch_resample32(const ConvArgs& a, hipStream_t st) {
    const bool down = a.kind == 0;
    const long tiles = (long)a.NF * (down ? a.H / 2 : a.H) * ((down ? a.W / 2 : a.W) / 16);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    const long waves = std::min<long>(tiles, (long)cus * 8);              // 8 waves per CU (two workgroups of 4 at <= 256 registers)
    const int tpw = (int)((tiles + waves - 1) / waves);
    const long blocks = (tiles + (long)tpw * 4 - 1) / ((long)tpw * 4);
    if (down) hipLaunchKernelGGL(resample32_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, st, a, tpw, tiles);
    else hipLaunchKernelGGL(resample32_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, a, tpw, tiles);
    return hipGetLastError();
}

}  // namespace vdx
