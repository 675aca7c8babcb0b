// ccp_comm.hpp — RCCL behind the C ABI (ccp_comm_*): one communicator per process / GPU, used by the
// row-blocked grid path for its neighbour halo exchange and its all-reduced norms.
//
// RCCL is bound at run time (dlopen of librccl.so.1 at the first ccp_comm_* call), not at link time:
//   * libccp_gs.so keeps loading on hosts that never go multi-GPU;
//   * inside a process that already carries an RCCL (PyTorch-ROCm bundles its own build next to its own
//     HIP runtime) the soname lookup returns THAT copy, so the communicator runs on the collective
//     library that matches the process's HIP runtime instead of mixing two builds.
// Only the types of <rccl/rccl.h> are used at compile time.
#pragma once

#include "ccp_common.hpp"

#include <rccl/rccl.h>

namespace ccp {

struct RcclApi {
    ncclResult_t (*GetVersion)(int *);
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*CommAbort)(ncclComm_t);
    const char *(*GetErrorString)(ncclResult_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
};

// The bound entry points, or nullptr when no librccl.so.1 can be loaded.
const RcclApi *rccl_api();

inline int rccl_fail(ncclResult_t r, const char *what, const char *file, int line)
{
    if (getenv("CCP_GS_DEBUG")) {
        const RcclApi *api = rccl_api();
        fprintf(stderr, "[ccp_gs] %s failed: %s (%s:%d)\n", what, api ? api->GetErrorString(r) : "?", file, line);
    }
    return CCP_ERR_RCCL;
}

#define CCP_RCCL(call)                                                        \
    do {                                                                      \
        ncclResult_t r_ = (call);                                             \
        if (r_ != ncclSuccess) return ::ccp::rccl_fail(r_, #call, __FILE__, __LINE__); \
    } while (0)

}  // namespace ccp

// One rank of a communicator, bound to one HIP device.
struct ccp_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    int version = 0;                    // ncclGetVersion of the library in use
    ccp::DevBuf<double> scratch;        // small device buffer for reductions / gathers (>= 8*world ints)
};
