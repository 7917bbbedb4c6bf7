// Error convention, device selection and the per-device arena of libblaze_hip.
#include "common.hpp"

#include <condition_variable>
#include <memory>

#include <chrono>
#include <cstdlib>
#include <thread>

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

int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return s && *s ? atoi(s) : dflt;
}

int plan_override(const char* key, int dflt) {
    const char* s = getenv("BLAZE_MSM_PLAN");
    if (!s || !*s) return dflt;
    const size_t kl = strlen(key);
    for (const char* p = s; *p;) {
        while (*p == ',' || *p == ' ') ++p;
        if (strncmp(p, key, kl) == 0 && p[kl] == '=') return atoi(p + kl + 1);
        while (*p && *p != ',') ++p;
    }
    return dflt;
}

int wait_timeout_ms() {
    const char* s = getenv("BLAZE_WAIT_TIMEOUT_MS");
    int v = s && *s ? atoi(s) : 120000;
    return v > 0 ? v : 120000;
}

static thread_local bool g_wait_timed_out = false;
bool wait_timed_out() { return g_wait_timed_out; }
void wait_clear() { g_wait_timed_out = false; }

// poll `query` (hipSuccess = done, hipErrorNotReady = pending) against the deadline.  The first 3 ms spin (a small task -
// the reference's default 8192 elements - is done within 2 ms, and a nap oversleeps by ~60 us), then 50 us naps.
template <class Q>
static int bounded_wait(Q&& query, const char* what) {
    g_wait_timed_out = false;
    const auto t0 = std::chrono::steady_clock::now();
    const int limit_ms = wait_timeout_ms();
    for (;;) {
        hipError_t e = query();
        if (e == hipSuccess) return BLZ_OK;
        if (e != hipErrorNotReady) {
            (void)hipGetLastError();
            return fail_hip(BLZ_ERR_UNKNOWN, "%s failed: %s", what, hipGetErrorString(e));
        }
        const auto dt = std::chrono::steady_clock::now() - t0;
        if (dt > std::chrono::milliseconds(limit_ms)) {
            g_wait_timed_out = true;
            return fail(BLZ_ERR_UNKNOWN, "%s timed out after %d ms (BLAZE_WAIT_TIMEOUT_MS): the device task did not complete; "
                        "the handle accepts only reset / free now", what, limit_ms);
        }
        if (dt < std::chrono::microseconds(3000)) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}
int sync_event_bounded(hipEvent_t ev, const char* what) {
    return bounded_wait([ev] { return hipEventQuery(ev); }, what);
}
int sync_stream_bounded(hipStream_t st, const char* what) {
    return bounded_wait([st] { return hipStreamQuery(st); }, what);
}

// The whole device against the same deadline.  HIP has no query for "every stream of the device is idle", so the drain
// (hipDeviceSynchronize) runs on a helper thread and the caller waits for it against the deadline; on expiry the helper - parked
// inside the runtime until the wedged work ends, if ever - is abandoned (it owns its state through the shared_ptr).
namespace {
struct DrainJob {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    hipError_t err = hipSuccess;
};
}  // namespace
int sync_device_bounded(const char* what) {
    g_wait_timed_out = false;
    // nothing pending anywhere is the common case for callers that drain before freeing: the null stream's query covers the
    // blocking streams only, so it cannot replace the drain, but an idle device makes the drain itself return at once
    int dev = 0;
    BLZ_HIP(hipGetDevice(&dev), BLZ_ERR_UNKNOWN);
    auto job = std::make_shared<DrainJob>();
    std::thread([job, dev] {
        hipError_t e = hipSetDevice(dev);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        std::lock_guard<std::mutex> lk(job->mu);
        job->err = e;
        job->done = true;
        job->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(job->mu);
    const int limit_ms = wait_timeout_ms();
    if (!job->cv.wait_for(lk, std::chrono::milliseconds(limit_ms), [&] { return job->done; })) {
        g_wait_timed_out = true;
        return fail(BLZ_ERR_UNKNOWN, "%s timed out after %d ms (BLAZE_WAIT_TIMEOUT_MS): device work in flight did not complete; the "
                    "buffers it may still touch are leaked, not freed", what, limit_ms);
    }
    if (job->err != hipSuccess) {
        (void)hipGetLastError();
        return fail_hip(BLZ_ERR_UNKNOWN, "%s failed: %s", what, hipGetErrorString(job->err));
    }
    return BLZ_OK;
}

int DevBuf::reserve(size_t bytes, bool exact) {
    if (bytes <= cap) return BLZ_OK;
    if (p) {
        // hipFree waits for the whole device: bounded here instead (a wedged kernel may still use the old buffer: leak it)
        const int rc = sync_device_bounded("growing a device buffer");
        if (rc != BLZ_OK) {
            p = nullptr;
            cap = 0;
            return rc;
        }
        (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    size_t want = exact ? bytes : bytes + bytes / 8;  // slack so slightly larger tasks do not reallocate (not for buffers of a fixed size)
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        e = hipMalloc(&p, bytes);
        want = bytes;
    }
    if (e != hipSuccess) {
        p = nullptr;
        return fail_hip(BLZ_ERR_UNKNOWN, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    cap = want;
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
int blz_host_malloc(int device_id, size_t bytes, void** out) {
    if (!out) return blz::fail(BLZ_ERR_INVALID_PARAM, "null out");
    *out = nullptr;
    BLZ_TRY(blz::use_device(device_id));
    BLZ_HIP(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}
int blz_host_free(void* p) {
    if (!p) return BLZ_OK;
    BLZ_HIP(hipHostFree(p), BLZ_ERR_UNKNOWN);
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
