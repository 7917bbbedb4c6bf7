// libblaze_hip_aux: test hooks for the bounded waits (include/blaze_hip_aux.h).  A one-lane kernel on a handle's main stream
// (blz_msm_stream / blz_ntt_stream) that spins until the host clears *token (a slot of one pinned page the process keeps) or
// max_ms have passed on the device's wall clock - the cap keeps a failing test from wedging the GPU for good.  Test
// infrastructure: not part of libblaze_hip.so.
#include <mutex>

#include "common.hpp"

namespace blz {

__global__ void k_stall(uint32_t* flag, uint64_t max_ticks) {
    const uint64_t t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u && wall_clock64() - t0 < max_ticks)
        __builtin_amdgcn_s_sleep(127);
}

static int launch_stall(hipStream_t st, uint32_t max_ms, void** token) {
    if (!token) return fail(BLZ_ERR_INVALID_PARAM, "null token");
    if (max_ms == 0 || max_ms > 30000) return fail(BLZ_ERR_INVALID_PARAM, "stall cap must be 1..30000 ms");
    int dev = 0, khz = 0;
    BLZ_HIP(hipGetDevice(&dev), BLZ_ERR_UNKNOWN);
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;   // 100 MHz
    // one pinned page per process, 64 flags handed out in turn (a flag cannot be freed - nobody knows when its kernel has read
    // it for the last time - so none is allocated per call)
    static std::mutex mu;
    static uint32_t* pool = nullptr;
    static unsigned next = 0;
    uint32_t* flag = nullptr;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!pool) BLZ_HIP(hipHostMalloc((void**)&pool, 64 * 64), BLZ_ERR_UNKNOWN);
        flag = pool + 16 * (next++ % 64u);
    }
    *flag = 1u;
    hipLaunchKernelGGL(k_stall, dim3(1), dim3(1), 0, st, flag, (uint64_t)max_ms * (uint64_t)khz);
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    *token = flag;
    return BLZ_OK;
}

}  // namespace blz

using namespace blz;

extern "C" {

int blz_test_msm_stall(blz_msm* h, uint32_t max_ms, void** token) {
    void* st = nullptr;
    int dev = 0;
    BLZ_TRY(blz_msm_stream(h, &st, &dev));
    BLZ_TRY(use_device(dev));
    return launch_stall((hipStream_t)st, max_ms, token);
}

int blz_test_ntt_stall(blz_ntt* h, uint32_t max_ms, void** token) {
    void* st = nullptr;
    int dev = 0;
    BLZ_TRY(blz_ntt_stream(h, &st, &dev));
    BLZ_TRY(use_device(dev));
    return launch_stall((hipStream_t)st, max_ms, token);
}

int blz_test_stall_release(void* token) {
    if (!token) return blz::fail(BLZ_ERR_INVALID_PARAM, "null token");
    __atomic_store_n((uint32_t*)token, 0u, __ATOMIC_RELEASE);
    return BLZ_OK;
}

}  // extern "C"
