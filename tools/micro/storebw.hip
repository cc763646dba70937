// Store-pattern micro-benchmark (gfx950): HBM write rate of a [rows][C] bf16 tensor for the accumulator-layout epilogue
// (lane (r, q) writes 8 bytes: 4 channels of row r; four instructions cover 16 rows x 128 bytes) against row-contiguous
// 16-byte stores (8 lanes per 128-byte row).  hipcc --offload-arch=gfx950 -O3 storebw.hip -o storebw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// pattern 0: per wave 16 rows x 64 channels (128 B): instr tm writes bytes [tm*32 + q*8, +8) of row r
// pattern 1: per wave 8 rows x 128 B per instr: lane l -> row l >> 3, 16-byte piece l & 7 ; 2 instr for 16 rows
// pattern 2: like 0 but 16 B per lane (fp32 output of 4 channels): 16 rows x 256 B over 4 instr
template <int PAT>
__global__ __launch_bounds__(256) void store_kernel(char* y, long rows, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const long wave_id = (long)blockIdx.x * 4 + wave, nwaves = (long)gridDim.x * 4;
    for (long t = wave_id; t < rows / 16; t += nwaves) {
        const long row0 = t * 16;
        if (PAT == 0) {
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
                *reinterpret_cast<uint2*>(y + (row0 + r) * 128 + tm * 32 + q * 8) = make_uint2((unsigned)t, (unsigned)tm);
        } else if (PAT == 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                *reinterpret_cast<uint4*>(y + (row0 + h * 8 + (lane >> 3)) * 128 + (lane & 7) * 16) = make_uint4((unsigned)t, (unsigned)h, 0u, 1u);
        } else {
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
                *reinterpret_cast<uint4*>(y + (row0 + r) * 256 + tm * 64 + q * 16) = make_uint4((unsigned)t, (unsigned)tm, 0u, 1u);
        }
    }
}

template <int PAT>
static void run(const char* name, char* y, long rows, int row_bytes) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {1024, 4096, 16384}) {
        hipLaunchKernelGGL(store_kernel<PAT>, dim3(grid), dim3(256), 0, 0, y, rows, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(store_kernel<PAT>, dim3(grid), dim3(256), 0, 0, y, rows, 1);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s grid %6d: %7.1f us  %5.2f TB/s\n", name, grid, ms * 100.f, (double)rows * row_bytes / (ms * 1e-4) / 1e12);
    }
}

int main() {
    const long rows = 4L << 20;                       // 4 M rows (the level-0 activation of the bench batch)
    char* y = nullptr;
    if (hipMalloc(&y, (size_t)rows * 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    run<0>("8 B per lane, 16 rows x 32 B per instr (bf16)", y, rows, 128);
    run<1>("16 B per lane, 8 rows x 128 B per instr (bf16)", y, rows, 128);
    run<2>("16 B per lane, 16 rows x 64 B per instr (fp32)", y, rows, 256);
    hipFree(y);
    return 0;
}
