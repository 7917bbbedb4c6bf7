// Error convention, device selection and the per-device arena of libblaze_hip.
#include "common.hpp"

#include <cstdlib>

namespace blz {

static thread_local char g_err[1024] = "";

void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    BLZ_LOG(1, "error: %s", g_err);
}

int log_level() {
    static int lvl = -1;
    if (lvl < 0) {
        const char* s = getenv("BLAZE_LOG");
        lvl = s && *s ? atoi(s) : 0;
    }
    return lvl;
}

int device_count() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int use_device(int device_id) {
    int n = device_count();
    if (device_id < 0 || device_id >= n)
        return fail(BLZ_ERR_FILE, "no HIP device with ordinal %d (%d visible); this library has no CPU path", device_id, n);
    BLZ_HIP(hipSetDevice(device_id), BLZ_ERR_FILE);
    return BLZ_OK;
}

int ensure_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> done;   // (device, kernel) -> bytes granted
    int dev = 0;
    BLZ_HIP(hipGetDevice(&dev), BLZ_ERR_UNKNOWN);
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_pair(dev, kernel);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return BLZ_OK;
    BLZ_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), BLZ_ERR_UNKNOWN);
    done[key] = bytes;
    return BLZ_OK;
}

}  // namespace blz

extern "C" {

const char* blz_last_error_message(void) { return blz::g_err; }
int blz_device_count(void) { return blz::device_count(); }
size_t blz_point_size(int curve) { return curve == BLZ_BN254 ? 64 : (curve == BLZ_BLS377 || curve == BLZ_BLS381) ? 96 : 0; }
size_t blz_result_size(int curve) { return curve == BLZ_BN254 ? 96 : (curve == BLZ_BLS377 || curve == BLZ_BLS381) ? 144 : 0; }

int blz_device_malloc(int device_id, size_t bytes, void** out) {
    if (!out) return blz::fail(BLZ_ERR_INVALID_PARAM, "null out");
    BLZ_TRY(blz::use_device(device_id));
    BLZ_HIP(hipMalloc(out, bytes ? bytes : 16), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}
int blz_device_free(int device_id, void* p) {
    BLZ_TRY(blz::use_device(device_id));
    BLZ_HIP(hipFree(p), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}
int blz_memcpy_h2d(int device_id, void* d_dst, const void* src, size_t bytes) {
    BLZ_TRY(blz::use_device(device_id));
    BLZ_HIP(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice), BLZ_ERR_WRITE);
    return BLZ_OK;
}
int blz_memcpy_d2h(int device_id, void* dst, const void* d_src, size_t bytes) {
    BLZ_TRY(blz::use_device(device_id));
    BLZ_HIP(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost), BLZ_ERR_READ);
    return BLZ_OK;
}

}  // extern "C"
