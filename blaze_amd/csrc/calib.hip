// Calibration of the resource the MSM / NTT kernels saturate: the issue rate of v_mad_u64_u32 on THIS device at
// its clocks of THIS moment (include/blaze_hip.h blz_calib_mad_rate).  bench.py prices `roofline.integer_issue`
// against it instead of a constant measured on another box; not on the product path.
#include "common.hpp"

namespace blz {

// 8 independent 64-bit accumulators per lane, 8 multiply-adds per loop trip: no dependency stalls, no memory
__global__ __launch_bounds__(256) void k_calib_mad(uint64_t* out, uint32_t reps, uint32_t seed) {
    const uint32_t x = (threadIdx.x + blockIdx.x * 256u) * 2654435761u + seed, y = x ^ 0x9e3779b9u;
    uint64_t a0 = x, a1 = y, a2 = x + 1, a3 = y + 1, a4 = x + 2, a5 = y + 2, a6 = x + 3, a7 = y + 3;
    for (uint32_t r = 0; r < reps; ++r) {
        asm volatile(
            "v_mad_u64_u32 %[a0], vcc, %[x], %[y], %[a0]\n\t"
            "v_mad_u64_u32 %[a1], vcc, %[y], %[x], %[a1]\n\t"
            "v_mad_u64_u32 %[a2], vcc, %[x], %[x], %[a2]\n\t"
            "v_mad_u64_u32 %[a3], vcc, %[y], %[y], %[a3]\n\t"
            "v_mad_u64_u32 %[a4], vcc, %[x], %[y], %[a4]\n\t"
            "v_mad_u64_u32 %[a5], vcc, %[y], %[x], %[a5]\n\t"
            "v_mad_u64_u32 %[a6], vcc, %[x], %[x], %[a6]\n\t"
            "v_mad_u64_u32 %[a7], vcc, %[y], %[y], %[a7]\n\t"
            : [a0] "+&v"(a0), [a1] "+&v"(a1), [a2] "+&v"(a2), [a3] "+&v"(a3), [a4] "+&v"(a4), [a5] "+&v"(a5), [a6] "+&v"(a6),
              [a7] "+&v"(a7)
            : [x] "v"(x), [y] "v"(y)
            : "vcc");
    }
    const uint64_t s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (s == 0x123456789abcull) out[0] = s;   // keep the chains alive
}

}  // namespace blz

using namespace blz;

extern "C" int blz_calib_mad_rate(int device_id, uint32_t target_ms, double out[4]) {
    if (!out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    if (target_ms == 0 || target_ms > 2000) return fail(BLZ_ERR_INVALID_PARAM, "target_ms must be 1..2000");
    BLZ_TRY(use_device(device_id));
    int cus = 0, khz = 0;
    BLZ_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id), BLZ_ERR_UNKNOWN);
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, device_id) != hipSuccess) khz = 0;
    uint64_t* d = nullptr;
    BLZ_HIP(hipMalloc((void**)&d, 64), BLZ_ERR_UNKNOWN);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t he = hipEventCreate(&e0);
    if (he == hipSuccess) he = hipEventCreate(&e1);
    int rc = BLZ_OK;
    if (he != hipSuccess) rc = fail_hip(BLZ_ERR_UNKNOWN, "event creation failed");
    // 4 waves per SIMD (where the multiplier chains of the product kernels peak: profiles/r02_mul_variants.txt)
    const dim3 grid((unsigned)cus * 4), block(256);
    uint32_t reps = 1u << 14;
    float ms = 0.f;
    for (int pass = 0; pass < 2 && rc == BLZ_OK; ++pass) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_calib_mad, grid, block, 0, 0, d, reps, 7u + pass);
        (void)hipEventRecord(e1, 0);
        if (sync_event_bounded(e1, "calibration kernel") != BLZ_OK || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) {
            rc = fail_hip(BLZ_ERR_UNKNOWN, "calibration kernel failed");
            break;
        }
        if (pass == 0) {   // the first launch only sizes the second
            double want = (double)reps * (double)target_ms / (double)ms;
            if (want > 4.0e9) want = 4.0e9;
            if (want < 1024) want = 1024;
            reps = (uint32_t)want;
        }
    }
    if (rc == BLZ_OK) {
        const double ops = (double)cus * 4.0 * 256.0 * 8.0 * (double)reps;
        out[0] = ops / ((double)ms * 1e-3);
        out[1] = (double)ms;
        out[2] = (double)khz / 1000.0;
        out[3] = ops;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d);
    return rc;
}
