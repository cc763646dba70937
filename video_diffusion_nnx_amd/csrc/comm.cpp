// Data-parallel communicator behind the C ABI (vdx.h: vdx_comm_unique_id / vdx_comm_init / vdx_allreduce_bucket / vdx_comm_destroy):
// RCCL over xGMI, one communicator per handle = per rank / GPU.  Replaces the all-reduce XLA inserts for the reference's data-sharded
// value_and_grad (trainer.py:161-177, 307-320, 361; SURVEY section 2 collective C1).
//
// RCCL is resolved at run time (dlopen of librccl.so.1 -- the copy the process has already loaded, e.g. PyTorch's, when there is one)
// so that libvdx.so has no link-time dependency on it: single-GPU users never load a communication library.
#include <dlfcn.h>
#include <string.h>
#include "vdx_internal.h"
#include "comm.h"

namespace vdx {

namespace {
// the slice of rccl.h this file uses (ABI-stable since NCCL 2.x: rccl.h:40-43, 187, 220, 260, 339, 448, 466, 611)
typedef struct { char internal[VDX_UNIQUE_ID_BYTES]; } UniqueId;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*CommDestroyFn)(void*);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*ErrStrFn)(int);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;

struct Rccl {
    void* lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr; CommInitRankFn comm_init_rank = nullptr; CommDestroyFn comm_destroy = nullptr;
    AllReduceFn all_reduce = nullptr; ErrStrFn err_str = nullptr;
    char why[256] = "";
};

Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return &r;
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    r.lib = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);            // already in the process (PyTorch's)?
    for (int i = 0; !r.lib && i < 3; ++i) r.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) { snprintf(r.why, sizeof(r.why), "RCCL not found: %s", dlerror()); return &r; }
    r.get_unique_id = (GetUniqueIdFn)dlsym(r.lib, "ncclGetUniqueId");
    r.comm_init_rank = (CommInitRankFn)dlsym(r.lib, "ncclCommInitRank");
    r.comm_destroy = (CommDestroyFn)dlsym(r.lib, "ncclCommDestroy");
    r.all_reduce = (AllReduceFn)dlsym(r.lib, "ncclAllReduce");
    r.err_str = (ErrStrFn)dlsym(r.lib, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce) {
        snprintf(r.why, sizeof(r.why), "RCCL symbols missing in the loaded librccl");
        r.lib = nullptr;
    }
    return &r;
}

int fail(const Rccl* r, int rc, const char* what, const char* file, int line) {
    char msg[320];
    snprintf(msg, sizeof(msg), "%s: %s", what, r->err_str ? r->err_str(rc) : "RCCL error");
    return vdx_set_error(VDX_ERR_HIP, msg, file, line);
}
}  // namespace

int comm_unique_id(void* out) {
    Rccl* r = rccl();
    if (!r->lib) return vdx_set_error(VDX_ERR_STATE, r->why, __FILE__, __LINE__);
    UniqueId id;
    const int rc = r->get_unique_id(&id);
    if (rc) return fail(r, rc, "ncclGetUniqueId", __FILE__, __LINE__);
    memcpy(out, id.internal, VDX_UNIQUE_ID_BYTES);
    return VDX_OK;
}

int comm_init(Comm* c, int rank, int world, const void* unique_id) {
    Rccl* r = rccl();
    if (!r->lib) return vdx_set_error(VDX_ERR_STATE, r->why, __FILE__, __LINE__);
    if (c->comm) return vdx_set_error(VDX_ERR_STATE, "comm_init: this handle already holds a communicator", __FILE__, __LINE__);
    UniqueId id;
    memcpy(id.internal, unique_id, VDX_UNIQUE_ID_BYTES);
    const int rc = r->comm_init_rank(&c->comm, world, id, rank);
    if (rc) { c->comm = nullptr; return fail(r, rc, "ncclCommInitRank", __FILE__, __LINE__); }
    c->rank = rank; c->world = world;
    return VDX_OK;
}

int comm_allreduce(Comm* c, float* ptr, size_t count, hipStream_t st) {
    if (!c->comm) return vdx_set_error(VDX_ERR_STATE, "allreduce_bucket: vdx_comm_init has not been called on this handle", __FILE__, __LINE__);
    Rccl* r = rccl();
    const int rc = r->all_reduce(ptr, ptr, count, kNcclFloat32, kNcclSum, c->comm, st);      // in place, sum; the 1/world lives in the Adam read
    if (rc) return fail(r, rc, "ncclAllReduce", __FILE__, __LINE__);
    return VDX_OK;
}

void comm_destroy(Comm* c) {
    if (!c->comm) return;
    Rccl* r = rccl();
    if (r->lib) (void)r->comm_destroy(c->comm);
    c->comm = nullptr; c->world = 1; c->rank = 0;
}

}  // namespace vdx
