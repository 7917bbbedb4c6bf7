// The device arena: flat, byte-addressed "HBM" per device (common.hpp), plus its cross-process form.
//
// Cross-process persistence (the reference's HBM tests rely on bases that an earlier
// process loaded: tests/integration_msm_hbm.rs:51-56 keeps the load commented out).  GPU memory belongs to a
// process, so "persistent" means: a holder process loads the bases and EXPORTS its arena (one
// hipIpcMemHandle per extent, written to a registry file); any other process ATTACHES the registry and
// addresses the same bytes through load_data_to_hbm / hbm_point_addr.  The bytes live as long as the holder.
#include "msm_engine.hpp"

#include <atomic>
#include <cstdlib>
#include <string>

namespace blz {

static std::mutex g_arena_mu;
static std::map<int, Arena*> g_arenas;

Arena& arena_for(int device_id) {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    auto it = g_arenas.find(device_id);
    if (it == g_arenas.end()) it = g_arenas.emplace(device_id, new Arena()).first;
    return *it->second;
}

ArenaExtent* arena_find(Arena& a, uint64_t pos, size_t len) {
    if (pos + len < pos) return nullptr;   // (a range that wraps around 2^64 lies in no extent)
    for (auto& e : a.ext)
        if (pos >= e.start && pos + len <= e.start + e.len) return &e;
    return nullptr;
}

uint32_t* arena_flag_acquire(Arena& a) {
    if (!a.build_flags) {
        if (hipMalloc((void**)&a.build_flags, 256 * sizeof(uint32_t)) != hipSuccess) {
            (void)hipGetLastError();
            a.build_flags = nullptr;
            return nullptr;
        }
        a.flag_free.clear();
        for (int i = 255; i >= 0; --i) a.flag_free.push_back((uint16_t)i);
    }
    if (a.flag_free.empty()) return nullptr;
    uint32_t* f = a.build_flags + a.flag_free.back();
    a.flag_free.pop_back();
    return f;
}
void arena_flag_release(Arena& a, uint32_t*& f) {
    if (f && a.build_flags && f >= a.build_flags && f < a.build_flags + 256) a.flag_free.push_back((uint16_t)(f - a.build_flags));
    f = nullptr;
}

// a build in flight (the caller has drained the device)
void arena_drop_build(Arena& a, ArenaExtent& x) {
    if (x.build.tab) (void)hipFree(x.build.tab);
    if (x.build.done) (void)hipEventDestroy(x.build.done);
    if (x.build.t0) (void)hipEventDestroy(x.build.t0);
    arena_flag_release(a, x.build.flag);
    x.build = ArenaExtent::TableBuild();
}
// the tables and a build in flight (the caller has drained the device)
void arena_drop_table(Arena& a, ArenaExtent& x) {
    for (auto& t : x.tables)
        if (t.p) (void)hipFree(t.p);
    x.tables.clear();
    x.tab_dirty_lo = x.tab_dirty_hi = 0;
    arena_drop_build(a, x);
}

void arena_free_extent(Arena& a, ArenaExtent& x) {
    if (x.raw) {
        if (x.imported) (void)hipIpcCloseMemHandle(x.raw);
        else (void)hipFree(x.raw);
    }
    if (x.mont) (void)hipFree(x.mont);
    arena_drop_table(a, x);
    if (x.shadow_ready) (void)hipEventDestroy(x.shadow_ready);
    if (x.diet_ev) (void)hipEventDestroy(x.diet_ev);
    arena_flag_release(a, x.diet_flag);
    x = ArenaExtent();
}

uint64_t arena_next_epoch() {
    static std::atomic<uint64_t> next{1};
    return next.fetch_add(1);
}

static size_t format_point_bytes(int format_id) { return (format_id & 0xff) == BLZ_BN254 ? 64 : 96; }

int arena_restore_raw(Arena& a, ArenaExtent& e, hipStream_t st) {
    (void)a;
    if (e.diet == 1) e.diet = 0;        // (a check in flight is simply forgotten: its event and the extent's flag word are reused)
    if (e.diet != 2) return BLZ_OK;
    const size_t ps = format_point_bytes(e.mont_curve), mp = mont_point_bytes(e.mont_curve & 0xff);
    void* raw = nullptr;
    BLZ_HIP(hipMalloc(&raw, e.cap), BLZ_ERR_UNKNOWN);
    int rc = msm_points_from_mont(e.mont_curve, e.mont, raw, e.len / ps, st);
    if (rc == BLZ_OK) rc = sync_stream_bounded(st, "arena: raw bytes back from the Montgomery copy");
    if (rc != BLZ_OK) {
        if (!wait_timed_out()) (void)hipFree(raw);
        return rc;
    }
    (void)mp;
    e.raw = raw;
    e.diet = 0;
    BLZ_LOG(1, "arena diet: extent at %llu: %zu raw bytes restored from the Montgomery copy", (unsigned long long)e.start, e.len);
    return BLZ_OK;
}

int arena_diet_step(Arena& a, ArenaExtent& e, size_t ps, hipStream_t st) {
    if (!(a.policy & 1) || e.imported || e.exported || e.diet >= 2 || !e.raw || !e.mont) return BLZ_OK;
    // the copy must BE the bytes: every point converted, the point grid flush with the extent, no window table on the extent (its
    // handle would restore the bytes at every launch) or being tabulated from them, and a format that holds every base (not the even-base copy of a checked precompute table)
    if (e.dirty_lo < e.dirty_hi || e.mont_phase != 0 || e.len % ps != 0 || (e.mont_curve >> 16) != 0 || e.build.tab || !e.tables.empty()) return BLZ_OK;
    if (e.diet == 0) {
        // (the extent keeps its flag word for as long as it lives: the check may sit in state 1 until a task next touches the extent)
        if (!e.diet_flag && !(e.diet_flag = arena_flag_acquire(a))) return BLZ_OK;
        if (!e.diet_ev) BLZ_HIP(hipEventCreateWithFlags(&e.diet_ev, hipEventDisableTiming), BLZ_ERR_UNKNOWN);
        BLZ_HIP(hipMemsetAsync(e.diet_flag, 0, 4, st), BLZ_ERR_UNKNOWN);
        BLZ_TRY(msm_points_all_canonical(e.mont_curve, e.raw, e.len / ps, e.diet_flag, st));
        BLZ_HIP(hipEventRecord(e.diet_ev, st), BLZ_ERR_UNKNOWN);
        e.diet = 1;
        return BLZ_OK;
    }
    // diet == 1: the check (and, in front of it on the same stream, the last conversion) may be through
    const hipError_t q = hipEventQuery(e.diet_ev);
    if (q == hipErrorNotReady) return BLZ_OK;
    if (q != hipSuccess) { (void)hipGetLastError(); e.diet = 0; return BLZ_OK; }
    uint32_t flag_h = 1;
    BLZ_HIP(hipMemcpy(&flag_h, e.diet_flag, 4, hipMemcpyDeviceToHost), BLZ_ERR_READ);
    if (flag_h) {
        BLZ_LOG(1, "arena diet: extent at %llu holds a coordinate >= q: its raw bytes stay", (unsigned long long)e.start);
        e.diet = 3;
        return BLZ_OK;
    }
    // conversions of OTHER handles read the raw bytes too (they are chained behind shadow_ready, which is complete here), and
    // hipFree waits for the whole device: a bounded drain first
    BLZ_TRY(sync_device_bounded("arena diet: dropping the raw bytes"));
    (void)hipFree(e.raw);
    e.raw = nullptr;
    e.diet = 2;
    BLZ_LOG(1, "arena diet: extent at %llu: %zu raw bytes dropped, the Montgomery copy (%zu bytes) is the only copy", (unsigned long long)e.start, e.cap,
            e.mont_bytes);
    return BLZ_OK;
}

int arena_read_bytes(Arena& a, ArenaExtent& e, uint64_t off, size_t len, void* out, hipStream_t st) {
    (void)a;
    if (len == 0) return BLZ_OK;
    if (e.diet != 2) {
        BLZ_HIP(hipMemcpy(out, (const char*)e.raw + off, len, hipMemcpyDeviceToHost), BLZ_ERR_READ);
        return BLZ_OK;
    }
    // the points that cover [off, off + len), converted back into a bounce buffer
    const size_t ps = format_point_bytes(e.mont_curve), mp = mont_point_bytes(e.mont_curve & 0xff);
    const uint64_t p0 = off / ps, p1 = (off + len + ps - 1) / ps;
    void* tmp = nullptr;
    BLZ_HIP(hipMalloc(&tmp, (size_t)(p1 - p0) * ps), BLZ_ERR_READ);
    int rc = msm_points_from_mont(e.mont_curve, (const char*)e.mont + p0 * mp, tmp, p1 - p0, st);
    if (rc == BLZ_OK && hipMemcpyAsync(out, (const char*)tmp + (off - p0 * ps), len, hipMemcpyDeviceToHost, st) != hipSuccess)
        rc = fail_hip(BLZ_ERR_READ, "get_data_from_hbm: copy out of the bounce buffer failed");
    if (rc == BLZ_OK) rc = sync_stream_bounded(st, "get_data_from_hbm: raw bytes from the Montgomery copy");
    if (rc == BLZ_OK || !wait_timed_out()) (void)hipFree(tmp);
    return rc;
}

static void mark_dirty(ArenaExtent& e, uint64_t lo, uint64_t hi) {
    if (e.dirty_lo >= e.dirty_hi) { e.dirty_lo = lo; e.dirty_hi = hi; }
    else { if (lo < e.dirty_lo) e.dirty_lo = lo; if (hi > e.dirty_hi) e.dirty_hi = hi; }
}

int arena_write(int device_id, uint64_t pos, const void* src, size_t len, bool src_is_device, hipStream_t st) {
    BLZ_TRY(use_device(device_id));
    if (len == 0) return BLZ_OK;
    if (pos + len < pos) return fail(BLZ_ERR_INVALID_PARAM, "arena: [%llu, +%zu) wraps around 2^64", (unsigned long long)pos, len);
    Arena& A = arena_for(device_id);
    std::lock_guard<std::mutex> lk(A.mu);
    const uint64_t end = pos + len;
    // extents the write overlaps or touches.  An extent shared with other processes (attached here, or exported from
    // here) is READ-ONLY: every process converts the bytes into a private Montgomery shadow and tracks staleness
    // locally, so a write on either side would leave the other side's shadow silently stale; a write that merely
    // touches such an extent starts a new extent of this process.
    std::vector<size_t> hit;
    for (size_t i = 0; i < A.ext.size(); ++i) {
        const ArenaExtent& x = A.ext[i];
        if (x.imported || x.exported) {
            // the same rule on both sides of an export: the attachers keep private Montgomery shadows of these bytes and
            // map this very allocation, so after blz_arena_export the bytes are frozen for the holder as well
            if (pos < x.start + x.len && x.start < end)
                return fail(BLZ_ERR_WRITE, "arena: [%llu, +%zu) overlaps an extent that is %s, which makes it read-only "
                            "(release the arena to load new bases)", (unsigned long long)pos, len,
                            x.imported ? "attached from another process" : "exported to other processes");
            continue;
        }
        if (pos <= x.start + x.len && x.start <= end) hit.push_back(i);
    }
    // (a dieted extent gets its raw bytes back first: a write is byte-granular, the Montgomery copy is not)
    for (size_t i : hit) BLZ_TRY(arena_restore_raw(A, A.ext[i], st));
    ArenaExtent* e = nullptr;
    if (hit.size() == 1) {
        ArenaExtent& x = A.ext[hit[0]];
        if (pos >= x.start && end <= x.start + x.cap) {   // inside it, or an append within its allocation
            e = &x;
            if (end > x.start + x.len) x.len = (size_t)(end - x.start);
        }
    }
    if (!e) {
        // a new extent over the union; the old bytes outside [pos, end) are carried over (flat memory)
        uint64_t nstart = pos, nend = end;
        for (size_t i : hit) {
            const ArenaExtent& x = A.ext[i];
            if (x.start < nstart) nstart = x.start;
            if (x.start + x.len > nend) nend = x.start + x.len;
        }
        const size_t nlen = (size_t)(nend - nstart);
        ArenaExtent n;
        n.start = nstart;
        n.len = nlen;
        n.cap = hit.empty() ? nlen : nlen + nlen / 2;   // growing: leave room, so piecewise loads copy O(total) bytes
        hipError_t he = hipMalloc(&n.raw, n.cap);
        if (he != hipSuccess && n.cap != nlen) { n.cap = nlen; he = hipMalloc(&n.raw, n.cap); }
        if (he != hipSuccess)
            return fail_hip(BLZ_ERR_WRITE, "arena: hipMalloc(%zu) at offset %llu failed: %s", n.cap, (unsigned long long)nstart,
                        hipGetErrorString(he));
        if (!hit.empty()) {
            // tasks in flight may still read the old extents or their shadows: drain (bounded) before they go
            if (sync_device_bounded("load_data_to_hbm: drain before the extents merge") != BLZ_OK) {
                (void)hipFree(n.raw);
                return BLZ_ERR_WRITE;
            }
            for (size_t i : hit) {
                const ArenaExtent& x = A.ext[i];
                BLZ_HIP(hipMemcpyAsync((char*)n.raw + (x.start - nstart), x.raw, x.len, hipMemcpyDeviceToDevice, st), BLZ_ERR_WRITE);
            }
            if (sync_stream_bounded(st, "load_data_to_hbm: carrying the old extents over") != BLZ_OK) return BLZ_ERR_WRITE;   // (the new allocation is leaked: the copy may still run)
            for (size_t k = hit.size(); k-- > 0;) {
                arena_free_extent(A, A.ext[hit[k]]);
                A.ext.erase(A.ext.begin() + hit[k]);
            }
        }
        A.ext.push_back(n);
        e = &A.ext.back();
        e->mont_curve = -1;   // no shadow yet
    }
    mark_dirty(*e, pos - e->start, end - e->start);
    if (e->build.tab) {
        // a build in flight tabulates the old bytes: gone (it still writes its table: drain first)
        if (sync_device_bounded("load_data_to_hbm: drain before the window-table build goes") != BLZ_OK) return BLZ_ERR_WRITE;
        arena_drop_build(A, *e);
    }
    if (!e->tables.empty()) {
        // The window tables are a function of the bytes.  A small rewrite keeps them: the span is remembered and the next task
        // that asks for a table re-tabulates the rows of the rewritten bases first (arena_points_table); a large one drops them
        // (a task in flight may still gather from them: drain first) and the tasks that follow rebuild, paced as ever.
        const uint64_t lo = pos - e->start, hi = end - e->start;
        const uint64_t nlo = e->tab_dirty_lo < e->tab_dirty_hi && e->tab_dirty_lo < lo ? e->tab_dirty_lo : lo;
        const uint64_t nhi = e->tab_dirty_lo < e->tab_dirty_hi && e->tab_dirty_hi > hi ? e->tab_dirty_hi : hi;
        if (nhi - nlo > ArenaExtent::TABLE_PATCH_MAX_POINTS * 64) {   // (64 = the smaller point size: BN254)
            if (sync_device_bounded("load_data_to_hbm: drain before the window tables go") != BLZ_OK) return BLZ_ERR_WRITE;
            arena_drop_table(A, *e);
        } else {
            e->tab_dirty_lo = nlo;
            e->tab_dirty_hi = nhi;
        }
    }
    e->table_refused = false;
    // the table check was about the old bytes: a table that was consistent needs another look at the elements this write
    // touches (an element's eight bases are checked against each other only); anything else is checked from scratch
    if (e->pcheck.state == 1 || e->pcheck.state == 3) {
        const uint64_t lo = pos - e->start, hi = end - e->start;
        if (e->pcheck.state == 3) {
            e->pcheck.redo_lo = lo < e->pcheck.redo_lo ? lo : e->pcheck.redo_lo;
            e->pcheck.redo_hi = hi > e->pcheck.redo_hi ? hi : e->pcheck.redo_hi;
        } else {
            e->pcheck.redo_lo = lo;
            e->pcheck.redo_hi = hi;
            e->pcheck.state = 3;
        }
    } else {
        e->pcheck = ArenaExtent::PrecompCheck();
    }
    e->diet = 0;                               // (so was a diet refusal / a canonical check in flight)
    e->epoch = arena_next_epoch();
    char* dst = (char*)e->raw + (pos - e->start);
    hipError_t he = hipMemcpyAsync(dst, src, len, src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st);
    if (he != hipSuccess)
        return fail_hip(BLZ_ERR_WRITE, "arena write of %zu bytes at offset %llu failed: %s", len, (unsigned long long)pos,
                    hipGetErrorString(he));
    // the copy is ordered on the caller's main stream, behind its tasks in flight: a bounded wait (common.hpp)
    return sync_stream_bounded(st, "load_data_to_hbm: copy into the arena");
}

// registry file: magic, count, then per extent {start, len, hipIpcMemHandle_t}
static const uint64_t kRegistryMagic = 0x314152415a4c42ull;   // "BLZARA1"
struct RegistryRecord {
    uint64_t start, len;
    hipIpcMemHandle_t handle;
};

}  // namespace blz

using namespace blz;

extern "C" {

int blz_arena_release(int device_id) {
    BLZ_TRY(use_device(device_id));
    Arena& A = arena_for(device_id);
    std::lock_guard<std::mutex> lk(A.mu);
    BLZ_TRY(sync_device_bounded("arena release"));   // (on expiry nothing is freed: wedged work may still read the extents)
    for (auto& x : A.ext) arena_free_extent(A, x);
    A.ext.clear();
    if (A.build_scratch) (void)hipFree(A.build_scratch);
    A.build_scratch = nullptr;
    A.build_scratch_bytes = 0;
    if (A.build_flags) (void)hipFree(A.build_flags);
    A.build_flags = nullptr;
    A.flag_free.clear();
    return BLZ_OK;
}

int blz_arena_set_policy(int device_id, uint32_t policy) {
    if (policy & ~(uint32_t)BLZ_ARENA_DROP_RAW) return fail(BLZ_ERR_INVALID_PARAM, "unknown arena policy bits 0x%x", policy);
    BLZ_TRY(use_device(device_id));
    Arena& A = arena_for(device_id);
    std::lock_guard<std::mutex> lk(A.mu);
    A.policy = (int)policy;
    return BLZ_OK;
}

int blz_arena_export(int device_id, const char* path) {
    if (!path) return fail(BLZ_ERR_INVALID_PARAM, "null path");
    BLZ_TRY(use_device(device_id));
    Arena& A = arena_for(device_id);
    std::lock_guard<std::mutex> lk(A.mu);
    std::vector<RegistryRecord> recs;
    for (auto& x : A.ext) {
        if (x.imported) continue;   // only what this process owns
        BLZ_TRY(arena_restore_raw(A, x, nullptr));   // (other processes map the RAW allocation)
        RegistryRecord r;
        memset(&r, 0, sizeof(r));
        r.start = x.start;
        r.len = x.len;
        BLZ_HIP(hipIpcGetMemHandle(&r.handle, x.raw), BLZ_ERR_UNKNOWN);
        x.exported = true;
        recs.push_back(r);
    }
    std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(BLZ_ERR_FILE, "arena export: cannot write %s", tmp.c_str());
    uint64_t hdr[2] = {kRegistryMagic, (uint64_t)recs.size()};
    bool ok = fwrite(hdr, sizeof(hdr), 1, f) == 1;
    if (ok && !recs.empty()) ok = fwrite(recs.data(), sizeof(RegistryRecord), recs.size(), f) == recs.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) return fail(BLZ_ERR_FILE, "arena export: writing %s failed", path);
    return BLZ_OK;
}

int blz_arena_attach(int device_id, const char* path) {
    if (!path) return fail(BLZ_ERR_INVALID_PARAM, "null path");
    BLZ_TRY(use_device(device_id));
    FILE* f = fopen(path, "rb");
    if (!f) return fail(BLZ_ERR_FILE, "arena attach: cannot read %s", path);
    uint64_t hdr[2] = {0, 0};
    std::vector<RegistryRecord> recs;
    bool ok = fread(hdr, sizeof(hdr), 1, f) == 1 && hdr[0] == kRegistryMagic && hdr[1] < (1u << 20);
    if (ok) {
        recs.resize((size_t)hdr[1]);
        ok = recs.empty() || fread(recs.data(), sizeof(RegistryRecord), recs.size(), f) == recs.size();
    }
    fclose(f);
    if (!ok) return fail(BLZ_ERR_FILE, "arena attach: %s is not an arena registry", path);
    Arena& A = arena_for(device_id);
    std::lock_guard<std::mutex> lk(A.mu);
    // all or nothing: overlaps are checked before anything is mapped, and a failed open unmaps what this call mapped
    for (const RegistryRecord& r : recs)
        for (const ArenaExtent& x : A.ext)
            if (r.start < x.start + x.len && x.start < r.start + r.len)
                return fail(BLZ_ERR_INVALID_PARAM, "arena attach: [%llu, +%llu) overlaps an extent of this process",
                            (unsigned long long)r.start, (unsigned long long)r.len);
    const size_t before = A.ext.size();
    for (const RegistryRecord& r : recs) {
        ArenaExtent n;
        n.start = r.start;
        n.len = n.cap = (size_t)r.len;
        n.imported = true;
        hipError_t he = hipIpcOpenMemHandle(&n.raw, r.handle, hipIpcMemLazyEnablePeerAccess);
        if (he != hipSuccess) {
            while (A.ext.size() > before) {
                arena_free_extent(A, A.ext.back());
                A.ext.pop_back();
            }
            return fail_hip(BLZ_ERR_UNKNOWN, "arena attach: hipIpcOpenMemHandle of [%llu, +%llu) failed: %s (nothing was attached)",
                        (unsigned long long)r.start, (unsigned long long)r.len, hipGetErrorString(he));
        }
        n.dirty_lo = 0;
        n.dirty_hi = n.len;   // this process has no shadow of it yet
        n.epoch = arena_next_epoch();
        A.ext.push_back(n);
    }
    return BLZ_OK;
}

}  // extern "C"
