// Shared device-side helpers for the gfx950 (CDNA4) kernels.  Wave = 64 lanes everywhere.
//
// MFMA operand convention used by every GEMM-shaped kernel in this library
// ------------------------------------------------------------------------
// All operands are staged in LDS "K-contiguous": a row (an output channel of a weight matrix, or a
// pixel/token of an activation matrix) holds its reduction dimension in consecutive bytes, cut in
// 64-byte CHUNKS.  Lane l = (r = l & 15, q = l >> 4) reads the 16 bytes [16q, 16q+16) of row r of a
// chunk with ONE ds_read_b128 for either operand:
//   MODE_BF16: 16 B = 8 bf16 = k {8q..8q+7} of a 32-deep chunk  -> one v_mfma_f32_16x16x32_bf16
//   MODE_F16 : as MODE_BF16 with IEEE half operands (v_mfma_f32_16x16x32_f16): 3 more mantissa bits, 5-bit exponent
//   MODE_F32 : 16 B = 4 f32  = k {4q..4q+3} of a 16-deep chunk  -> four v_mfma_f32_16x16x4_f32,
//              step s pairing element s of both fragments (k = 4q+s: a permutation of the
//              reduction order shared by A and B, so the dot product is unchanged).
// The A operand is always the WEIGHT side (rows = output channels), the B operand the ACTIVATION
// side (columns = pixels/tokens), so the 16x16 accumulator of lane (r, q) holds output channels
// 4q..4q+3 (registers 0..3) of pixel r: four consecutive channels of one pixel = one float4 of a
// channel-last tensor.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vdx {

enum { MODE_F32 = 0, MODE_BF16 = 1, MODE_F16 = 2 };

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int CHUNK_BYTES = 64;    // one MFMA group of K per row
constexpr int ROW_BYTES = 128;     // K tile staged per row = 2 chunks
constexpr int ROW_STRIDE = 160;    // LDS row stride: 16 consecutive rows x 4 lane-quads hit 16 distinct 4-bank groups
constexpr float NORM_EPS = 1e-6f;  // Flax LayerNorm/GroupNorm epsilon
constexpr int GN_SLOTS = 32;       // GroupNorm partial-sum slots per (sample, group): spreads f64 atomics

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);   // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
    return __builtin_bit_cast(unsigned, r);
}

__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   // v_rcp_f32: 1 ulp

template <int MODE> struct Mma;

template <> struct Mma<MODE_F32> {
    static constexpr int ES = 4;                 // element bytes in LDS / packed weights
    static constexpr int KC = 16;                // elements per 64-byte chunk
    static constexpr int KT = 32;                // elements per staged row (2 chunks)
    static __device__ __forceinline__ void mma(f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
    // K = 16 step whose operands are fp32 register quadruples (lane (r,q) supplies k = 4q..4q+3): an accumulator tile of
    // a previous MFMA (rows 4q+reg) is directly such an operand
    static __device__ __forceinline__ void mma16(f32x4& acc, const f32x4& a, const f32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
    }
    // 4 consecutive-K weights of one row as an fp32 quadruple (for mma16's A operand)
    static __device__ __forceinline__ f32x4 load_w4(const char* p) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        return f32x4{v.x, v.y, v.z, v.w};
    }
    // store 4 consecutive-k values v at element offset k (multiple of 4) of an LDS row
    static __device__ __forceinline__ void store4(char* row, int k, float4 v) {
        *reinterpret_cast<float4*>(row + k * 4) = v;
    }
    static __device__ __forceinline__ void store1(char* row, int k, float v) {
        *reinterpret_cast<float*>(row + k * 4) = v;
    }
};

template <> struct Mma<MODE_BF16> {
    static constexpr int ES = 2;
    static constexpr int KC = 32;
    static constexpr int KT = 64;
    static __device__ __forceinline__ void mma(f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma16(f32x4& acc, const f32x4& a, const f32x4& b) {
        const uint2 ua = make_uint2(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]));
        const uint2 ub = make_uint2(pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3]));
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, ua), __builtin_bit_cast(s16x4, ub), acc, 0, 0, 0);
    }
    // attention-core variant (BASELINE.json configs[4], "fp8 attention QK^T / PV"): both operands rounded to OCP e4m3 (gfx950's
    // v_cvt_pk_fp8_f32), fp32 accumulate.  The 4 values of an accumulator-layout operand fill K slots 0..3 of the lane's 8 of
    // v_mfma_f32_16x16x32_fp8_fp8, slots 4..7 are zero on both sides: the same K = 16 product as mma16.
    static __device__ __forceinline__ void mma16_fp8(f32x4& acc, const f32x4& a, const f32x4& b) {
        int ua = __builtin_amdgcn_cvt_pk_fp8_f32(a[0], a[1], 0, false); ua = __builtin_amdgcn_cvt_pk_fp8_f32(a[2], a[3], ua, true);
        int ub = __builtin_amdgcn_cvt_pk_fp8_f32(b[0], b[1], 0, false); ub = __builtin_amdgcn_cvt_pk_fp8_f32(b[2], b[3], ub, true);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)(unsigned)ua, (long)(unsigned)ub, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 load_w4(const char* p) {
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u)};
    }
    static __device__ __forceinline__ void store4(char* row, int k, float4 v) {
        uint2 u;
        u.x = pack_bf16x2(v.x, v.y);
        u.y = pack_bf16x2(v.z, v.w);
        *reinterpret_cast<uint2*>(row + k * 2) = u;
    }
    static __device__ __forceinline__ void store1(char* row, int k, float v) {
        __bf16 h = (__bf16)v;
        *reinterpret_cast<__bf16*>(row + k * 2) = h;
    }
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_f16x2(float a, float b) {
    f32x2 v = {a, b};
    f16x2 r = __builtin_convertvector(v, f16x2);     // v_cvt_f16_f32 (RNE)
    return __builtin_bit_cast(unsigned, r);
}

// fp16 operands (the "fp16" of BASELINE.json configs[3]): same fragment geometry as bf16.  Activations and weights of this network
// stay far inside the fp16 range (GroupNorm / LayerNorm keep them O(1..10)); values beyond 65504 would saturate to inf.
template <> struct Mma<MODE_F16> {
    static constexpr int ES = 2;
    static constexpr int KC = 32;
    static constexpr int KT = 64;
    static __device__ __forceinline__ void mma(f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma16(f32x4& acc, const f32x4& a, const f32x4& b) {
        const uint2 ua = make_uint2(pack_f16x2(a[0], a[1]), pack_f16x2(a[2], a[3]));
        const uint2 ub = make_uint2(pack_f16x2(b[0], b[1]), pack_f16x2(b[2], b[3]));
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, ua), __builtin_bit_cast(f16x4, ub), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 load_w4(const char* p) {
        const f16x4 h = *reinterpret_cast<const f16x4*>(p);
        return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    }
    static __device__ __forceinline__ void store4(char* row, int k, float4 v) {
        *reinterpret_cast<uint2*>(row + k * 2) = make_uint2(pack_f16x2(v.x, v.y), pack_f16x2(v.z, v.w));
    }
    static __device__ __forceinline__ void store1(char* row, int k, float v) {
        *reinterpret_cast<_Float16*>(row + k * 2) = (_Float16)v;
    }
};

// QK^T / PV product of the <= 16-token attention cores: fp8 operands when F8 (bf16 mode only), the mode's K = 16 product otherwise
template <class M, bool F8> __device__ __forceinline__ void core_mma16(f32x4& acc, const f32x4& a, const f32x4& b) {
    if constexpr (F8) M::mma16_fp8(acc, a, b); else M::mma16(acc, a, b);
}

// one packed-weight element of the mode's operand type
template <int MODE> __device__ __forceinline__ void store_operand(void* dst, size_t i, float v) {
    if (MODE == MODE_F32) reinterpret_cast<float*>(dst)[i] = v;
    else if (MODE == MODE_BF16) reinterpret_cast<__bf16*>(dst)[i] = (__bf16)v;
    else reinterpret_cast<_Float16*>(dst)[i] = (_Float16)v;
}

// 4 consecutive channels of a tensor stored as fp32 or bf16 (element index i, a multiple of 4)
__device__ __forceinline__ float4 load4_f32_or_bf16(const float* base, size_t i, int is_bf16) {
    if (!is_bf16) return *reinterpret_cast<const float4*>(base + i);
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(base) + i * 2);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
}

// store 4 consecutive channels to a tensor stored as fp32 or bf16 (element index i, a multiple of 4)
__device__ __forceinline__ void store4_f32_or_bf16(float* base, size_t i, float4 v, int is_bf16) {
    if (!is_bf16) *reinterpret_cast<float4*>(base + i) = v;
    else *reinterpret_cast<uint2*>(reinterpret_cast<char*>(base) + i * 2) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
}

// Reductions over the 16 lanes that share (lane >> 4) (one DPP row), result in all of them: quad_perm [1,0,3,2], quad_perm
// [2,3,0,1], row_half_mirror, row_mirror -- four VALU ops with a DPP operand, no ds_bpermute round trip through the LDS.
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float reduce16(float v) {
    v += dpp_row<0xB1>(v);
    v += dpp_row<0x4E>(v);
    v += dpp_row<0x141>(v);
    v += dpp_row<0x140>(v);
    return v;
}
// reductions over the 4 lanes that share (lane & 15), result in all of them.  v_permlane32_swap exchanges the upper half
// of its first operand with the lower half of its second, v_permlane16_swap the odd 16-lane rows of the first with the
// even rows of the second: with both operands = v the two results are v[l] and v[l ^ 32] (resp. ^ 16) arranged so that
// one VALU op combines them -- no ds_bpermute round trip through the LDS crossbar.
__device__ __forceinline__ float reduce_q(float v) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float max_q(float v) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float max16(float v) {
    v = fmaxf(v, dpp_row<0xB1>(v));
    v = fmaxf(v, dpp_row<0x4E>(v));
    v = fmaxf(v, dpp_row<0x141>(v));
    v = fmaxf(v, dpp_row<0x140>(v));
    return v;
}
// n / d for n * d < 2^32 and d >= 2, with m = floor(2^32 / d) + 1 computed on the host: one v_mul_hi_u32
__device__ __forceinline__ int div_magic(int n, unsigned m) { return (int)__umulhi((unsigned)n, m); }

// GroupNorm statistics slab: stats[(b * GN_SLOTS + slot) * groups * 2 + g * 2 + {0: sum, 1: sumsq}]
__device__ __forceinline__ void gn_mean_rstd(const double* stats, int b, int g, int groups, double count,
                                             float& mean, float& rstd) {
    double s = 0.0, ss = 0.0;
    const double* p = stats + (size_t)b * GN_SLOTS * groups * 2 + g * 2;
    // loads in batches of 8 slots, all issued before the first add (one wait per batch instead of one L2 round trip per slot);
    // the summation order is the plain slot order
    static_assert(GN_SLOTS % 8 == 0, "GN_SLOTS");
#pragma unroll 1
    for (int k0 = 0; k0 < GN_SLOTS; k0 += 8) {
        double2 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const double2*>(p + (size_t)(k0 + k) * groups * 2);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s += t[k].x; ss += t[k].y; }
    }
    double m = s / count;
    double var = ss / count - m * m;     // fast variance, as Flax (use_fast_variance=True)
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)NORM_EPS));
}

// Workgroup form: thread t loads slot (t & 31) of group (t >> 5) -- every slot of every group in ONE batch of loads -- and the 32
// lanes of a group reduce with shuffles (a fixed tree, so the result does not depend on timing).  Writes gm[2g] = mean,
// gm[2g+1] = rstd (LDS); the caller synchronises.  nthreads is a multiple of 64.
__device__ __forceinline__ void gn_mean_rstd_wg(const double* stats, int b, int groups, double count, float* gm, int tid, int nthreads) {
    static_assert(GN_SLOTS == 32, "GN_SLOTS");
    for (int t = tid; t < groups * GN_SLOTS; t += nthreads) {
        const int g = t >> 5, slot = t & 31;
        const double2 v = *reinterpret_cast<const double2*>(stats + (((size_t)b * GN_SLOTS + slot) * groups + g) * 2);
        double s = v.x, ss = v.y;
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
        if (slot == 0) {
            const double m = s / count;
            double var = ss / count - m * m;     // fast variance, as Flax (use_fast_variance=True)
            if (var < 0.0) var = 0.0;
            gm[2 * g] = (float)m;
            gm[2 * g + 1] = (float)(1.0 / sqrt(var + (double)NORM_EPS));
        }
    }
}

}  // namespace vdx

#define VDX_CHECK_HIP(expr)                                                      \
    do {                                                                         \
        hipError_t _e = (expr);                                                  \
        if (_e != hipSuccess) return vdx_set_error(VDX_ERR_HIP, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
