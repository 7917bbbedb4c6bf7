// RCCL, resolved at run time: libblaze_hip.so does not link librccl, so single-GPU hosts load it without RCCL
// on the machine, and a process that already carries an RCCL (PyTorch ships one under the same soname) gets
// THAT instance instead of a second copy with its own state.
#pragma once
#include <rccl/rccl.h>   // types and prototypes only

namespace blz {

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
// nullptr (and the last error set) if RCCL cannot be found
const RcclApi* rccl_api();

}  // namespace blz
