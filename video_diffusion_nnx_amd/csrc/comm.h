// RCCL communicator of a handle (non-ABI; see comm.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace vdx {
struct Comm { void* comm = nullptr; int rank = 0, world = 1; };
int comm_unique_id(void* out);                                                  // VDX_UNIQUE_ID_BYTES bytes
int comm_init(Comm* c, int rank, int world, const void* unique_id);
int comm_allreduce(Comm* c, float* ptr, size_t count, hipStream_t st);         // in-place sum over the ranks
void comm_destroy(Comm* c);
}  // namespace vdx
