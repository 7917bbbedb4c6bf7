// RCCL, resolved at run time: libblaze_hip.so does not link librccl, so single-GPU hosts load it without RCCL on the
// machine.  Which copy: the one that sits NEXT TO THE HIP RUNTIME THIS LIBRARY IS BOUND TO (rccl_dyn.hip).  A process can
// carry two ROCm stacks - PyTorch bundles libamdhip64 / libhsa-runtime64 / librccl under the system sonames - and an
// RCCL from the other stack talks to an HSA runtime that was never initialised.  When PyTorch was imported first, the
// dynamic linker binds this library to PyTorch's HIP runtime, the RCCL next to it is PyTorch's, and dlopen of that path
// returns the instance PyTorch already uses (communicators of both live side by side in it); when this library came
// first, it is the system stack's.  Either way: one HIP runtime, the RCCL that belongs to it.
#pragma once
#include <rccl/rccl.h>   // types and prototypes only

namespace blz {

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;   // optional (nullptr if the library lacks it)
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char path[256] = "";                                // the file it was loaded from (diagnostics)
};
// nullptr (and the last error set) if RCCL cannot be found
const RcclApi* rccl_api();

}  // namespace blz
