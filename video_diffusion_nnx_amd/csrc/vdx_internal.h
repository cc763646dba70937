// Internal (non-ABI) declarations shared by the kernel translation units and the host runtime.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <algorithm>
#include "../../include/vdx.h"

int vdx_set_error(int code, const char* msg, const char* file, int line);

namespace vdx {

// Arguments of conv_igemm_kernel (POD, passed by value).  Caller fills the first block; launch_conv
// completes the geometry.
struct ConvArgs {
    // tensors (channel-last fp32): input = concat(x0[.., C0], x1[.., C1]) on the channel axis
    const float* x0; const float* x1; int C0, C1;
    const void* wp;                 // packed weights [taps][Cout][CinPad] (fp32 or bf16 per mode)
    const float* bias;              // [Cout] or null
    float* y; int Cout;
    int NF, F;                      // frames total (B*F), frames per sample
    int H, W;                       // input spatial size
    int kind;                       // 0: conv (kh, kw, stride, pad)   1: ConvTranspose 4x4 / stride 2 (4 phases)
    int kh, kw, stride, pad;
    // prologue on the input: 0 none, 1 GroupNorm-apply (+scale/shift) + SiLU
    int pro;
    const double* in_stats; const float* gamma; const float* beta; int groups;
    const float* ss; int ss_stride; // per-sample [scale(Cin) | shift(Cin)] rows, or null
    // epilogue: GroupNorm partial statistics of the output (or null)
    double* out_stats; int out_groups;
    // completed by launch_conv
    int Ho, Wo, Hy, Wy, CinPad, PH, PW, NP, tiles_y, tiles_x;
};

size_t conv_packed_bytes(int mode, int taps, int Cin, int Cout);
int conv_cin_pad(int mode, int Cin);
hipError_t launch_pack_weights(int mode, const float* src, void* dst, int taps, int Cin, int Cout, hipStream_t st);
hipError_t launch_conv(int mode, ConvArgs a, hipStream_t st);

}  // namespace vdx
