// LDS-DMA helpers shared by the weight-streaming kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace vdx {

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
// the per-tap sync of a weight ring: LDS-DMA count only (see conv_ws.hip)
template <int N> __device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm_lgkm0() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(N) : "memory");
}

}  // namespace vdx
