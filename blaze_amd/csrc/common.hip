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

static std::mutex g_arena_mu;
static std::map<int, Arena*> g_arenas;

Arena& arena_for(int device_id) {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    auto it = g_arenas.find(device_id);
    if (it == g_arenas.end()) it = g_arenas.emplace(device_id, new Arena()).first;
    return *it->second;
}

ArenaExtent* arena_find(Arena& a, uint64_t pos, size_t len) {
    for (auto& e : a.ext)
        if (pos >= e.start && pos + len <= e.start + e.len) return &e;
    return nullptr;
}

int arena_write(int device_id, uint64_t pos, const void* src, size_t len, bool src_is_device, hipStream_t st) {
    BLZ_TRY(use_device(device_id));
    if (len == 0) return BLZ_OK;
    Arena& A = arena_for(device_id);
    std::lock_guard<std::mutex> lk(A.mu);
    ArenaExtent* e = arena_find(A, pos, len);
    if (!e) {
        // drop every extent the new range overlaps, then create a fresh one
        for (size_t i = 0; i < A.ext.size();) {
            ArenaExtent& x = A.ext[i];
            bool overlap = pos < x.start + x.len && x.start < pos + len;
            if (overlap) {
                if (x.raw) (void)hipFree(x.raw);
                if (x.mont) (void)hipFree(x.mont);
                A.ext.erase(A.ext.begin() + i);
            } else {
                ++i;
            }
        }
        ArenaExtent n;
        n.start = pos;
        n.len = len;
        hipError_t he = hipMalloc(&n.raw, len);
        if (he != hipSuccess) return fail(BLZ_ERR_WRITE, "arena: hipMalloc(%zu) at offset %llu failed: %s", len,
                                          (unsigned long long)pos, hipGetErrorString(he));
        A.ext.push_back(n);
        e = &A.ext.back();
    }
    e->mont_curve = -1;  // shadow is stale
    char* dst = (char*)e->raw + (pos - e->start);
    hipError_t he = hipMemcpyAsync(dst, src, len, src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    if (he != hipSuccess)
        return fail(BLZ_ERR_WRITE, "arena write of %zu bytes at offset %llu failed: %s", len, (unsigned long long)pos,
                    hipGetErrorString(he));
    return BLZ_OK;
}

}  // namespace blz

extern "C" {

const char* blz_last_error_message(void) { return blz::g_err; }
int blz_device_count(void) { return blz::device_count(); }
size_t blz_point_size(int curve) { return curve == BLZ_BN254 ? 64 : (curve == BLZ_BLS377 || curve == BLZ_BLS381) ? 96 : 0; }
size_t blz_result_size(int curve) { return curve == BLZ_BN254 ? 96 : (curve == BLZ_BLS377 || curve == BLZ_BLS381) ? 144 : 0; }

int blz_arena_release(int device_id) {
    BLZ_TRY(blz::use_device(device_id));
    blz::Arena& A = blz::arena_for(device_id);
    std::lock_guard<std::mutex> lk(A.mu);
    for (auto& x : A.ext) {
        if (x.raw) (void)hipFree(x.raw);
        if (x.mont) (void)hipFree(x.mont);
    }
    A.ext.clear();
    return BLZ_OK;
}

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
