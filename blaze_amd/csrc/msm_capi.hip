// C ABI of the MSM primitive: the DriverPrimitive call sequence of src/ingo_msm/msm_api.rs
// (initialize -> start_process -> set_data -> wait_result -> result) over the device pipeline.
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <thread>

#include "msm_engine.hpp"
#include "rccl_dyn.hpp"

using namespace blz;

struct blz_msm {
    int device = 0;
    int mem_type = BLZ_DMA;  // PointMemoryType, msm_cfg.rs:11-14
    uint32_t pf = 1;         // precompute_factor, msm_api.rs:46-50
    int curve = BLZ_BLS381;
    // "registers" written by initialize (msm_api.rs:72-111)
    uint32_t nof_elements = 0;
    bool bases_from_hbm = false;
    uint64_t hbm_addr = 0;
    bool initialized = false;
    // task / result queues (msm_hw_code.rs:19-25)
    bool armed = false;        // a task was pushed and waits for its data
    bool data_ready = false;   // set_data delivered a complete input
    struct Pending { int slot; uint32_t label; };
    std::deque<Pending> in_flight;   // pipelines enqueued, results not collected yet (<= MSM_QUEUE_DEPTH)
    uint32_t task_label = 0;
    struct Res { std::vector<uint8_t> bytes; uint32_t label; };
    std::deque<Res> results;
    // staged input.  Host buffers land in one of TWO staging sets, used alternately: the copy of task k+1 must not
    // wait for task k's digit sort (which itself waits for task k-1's accumulation), or the PCIe link idles for
    // a sort per task; set_free[i] is recorded on the main stream when the task staged in set i has read it.
    DevBuf scalars_buf[2], points_raw[2], points_mont;
    hipEvent_t set_free[2] = {nullptr, nullptr};
    bool set_used[2] = {false, false};
    int stage_idx = 0, staged_set = -1;
    hipStream_t copy_stream = nullptr;  // host -> device staging: runs under the previous task's accumulation
    const void* d_scalars = nullptr;
    const void* d_points_mont = nullptr;
    uint32_t staged_n = 0;
    bool staged_from_arena = false;
    bool staged_loaded_now = false;   // this set_data also loaded the bases (mode iii: points + hbm address)
    uint64_t staged_arena_pos = 0;
    MsmEngine eng;
    // a wait ran into its deadline (BLAZE_WAIT_TIMEOUT_MS): device work of this handle may never complete, so nothing
    // new is queued behind it; reset (which waits, bounded, for the streams to drain) or free are the ways out
    bool wedged = false;
    // multi-GPU exchange (blz_msm_comm_*): one communicator rank per handle
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_size = 0;
    DevBuf comm_buf;   // [send: one partial | recv: comm_size partials]
    // resident-base window table (blz_msm_set_window_table; off for new handles)
    int window_table = 0;   // 0 off, 1 where it pays (the BLS curves), 2 always
    // scalar range of this handle's tasks (blz_msm_set_scalar_range): bits [range_lo, range_hi) of every scalar; 0, 0 = all
    int range_lo = 0, range_hi = 0;
    uint64_t table_info[4] = {0, 0, 0, 0};   // of the last HBM task: table bytes, window bits, windows, build time (us)
    // checked-table plan of a precompute handle (blz_msm_set_precompute_plan; off for new handles)
    int precompute_plan = 0;
    uint64_t pc_info[4] = {0, 0, 0, 0};      // of the last HBM task: took the plan, check state of its bases, check time (us), bytes of the even-base copy
};

namespace {

#define BLZ_LIVE(h)                                                                                              \
    do {                                                                                                         \
        if ((h)->wedged)                                                                                         \
            return fail(BLZ_ERR_UNKNOWN, "handle is wedged: an earlier wait timed out (BLAZE_WAIT_TIMEOUT_MS); only " \
                                         "reset / free are accepted");                                           \
    } while (0)
// a bounded wait of this handle: remember a deadline expiry
#define BLZ_WAIT(h, expr)                          \
    do {                                           \
        wait_clear();                              \
        int rc__ = (expr);                         \
        if (rc__ != BLZ_OK) {                      \
            if (wait_timed_out()) (h)->wedged = true; \
            return rc__;                           \
        }                                          \
    } while (0)

// hbm_point_addr = (addr, offset): the byte address addr + offset of the flat arena; a sum that wraps is nobody's address
#define BLZ_ARENA_ADDR(addr, off)                                                                                          \
    do {                                                                                                                   \
        if ((uint64_t)(addr) + (uint64_t)(off) < (uint64_t)(addr))                                                         \
            return fail(BLZ_ERR_INVALID_PARAM, "HBM address %llu + offset %llu wraps around 2^64", (unsigned long long)(addr), \
                        (unsigned long long)(off));                                                                        \
    } while (0)

size_t point_size(const blz_msm* h) { return blz_point_size(h->curve); }
size_t result_size(const blz_msm* h) { return blz_result_size(h->curve); }

static bool wants_table_mode(const blz_msm* h) { return h->pf == 1 && h->window_table != 0; }

// Resolve the Montgomery-form view of `npts` points stored at arena offset `pos`: (re)builds the part of the
// extent's shadow that is stale, on this handle's main stream, and orders this stream behind conversions other
// handles may have enqueued.
// even (checked-table plan of a precompute handle): the copy holds the even bases of every element only - B_0, B_2, B_4, B_6,
// contiguous, 4 per element - and *out addresses the copy of the element at `pos`; npts counts the RAW points (8 per element).
// Granted only while the extent's table check still stands for these points (looked up under the same lock that resolves the
// copy: a write by another thread between the check and this call leaves *out null, and the caller takes the exact path).
int arena_points_mont(blz_msm* h, uint64_t pos, uint32_t npts, const void** out, bool even = false) {
    *out = nullptr;
    const size_t ps = point_size(h), mp = mont_point_bytes(h->curve);
    const size_t len = (size_t)npts * ps;
    Arena& A = arena_for(h->device);
    std::lock_guard<std::mutex> lk(A.mu);
    ArenaExtent* e = arena_find(A, pos, len);
    if (!e)
        return fail(BLZ_ERR_INVALID_PARAM, "HBM bases: no loaded extent covers [%llu, +%zu) on device %d",
                    (unsigned long long)pos, len, h->device);
    const uint32_t phase = (uint32_t)((pos - e->start) % ps);   // where the point grid sits inside the extent
    const size_t cap_pts = (e->cap - phase) / ps, ext_pts = (e->len - phase) / ps;
    const uint64_t first = (pos - e->start - phase) / ps;
    if (even) {
        const ArenaExtent::PrecompCheck& C = e->pcheck;
        if ((first & 7u) || C.state != 1 || C.curve != h->curve || C.phase != phase || first < C.first || first + npts > C.first + C.npts) return BLZ_OK;
    }
    const int fmt = h->eng.format_id() | (even ? 1 << 16 : 0);
    const size_t want_bytes = (even ? (cap_pts / 8) * 4 : cap_pts) * mp;
    if (e->mont_curve != fmt || e->mont_phase != phase || e->mont_bytes < want_bytes) {
        // another curve / grid / layout (or the first use): a fresh shadow, everything stale
        BLZ_TRY(arena_restore_raw(A, *e, h->eng.stream));   // (a dieted extent: the new copy is made from the bytes)
        if (e->mont) {
            BLZ_TRY(sync_device_bounded("replacing a Montgomery shadow"));   // a task of another handle may still read the old one
            (void)hipFree(e->mont);
            e->mont = nullptr;
        }
        e->mont_bytes = want_bytes + 16;
        BLZ_HIP(hipMalloc(&e->mont, e->mont_bytes), BLZ_ERR_UNKNOWN);
        e->mont_curve = fmt;   // curve and layout of the copy (BN254 has two: msm_engine.hpp `repr`; bit 16: even bases only)
        e->mont_phase = phase;
        e->dirty_lo = 0;
        e->dirty_hi = e->len;
        e->shadow_recorded = false;   // (the device was drained above: nothing recorded earlier is still running)
    }
    if (!e->shadow_ready) BLZ_HIP(hipEventCreateWithFlags(&e->shadow_ready, hipEventDisableTiming), BLZ_ERR_UNKNOWN);
    // Conversions are chained through ONE event: whoever touches the shadow next - to read it or to convert another
    // span - first orders its stream behind the last conversion recorded, whichever handle enqueued it.  (Without the
    // wait in the dirty branch, handle B converting a small appended span re-recorded the event while handle A's
    // full-extent conversion was still running on A's stream, and B's task read points A had not written yet.)
    if (e->shadow_recorded) BLZ_HIP(hipStreamWaitEvent(h->eng.stream, e->shadow_ready, 0), BLZ_ERR_UNKNOWN);
    if (e->dirty_lo < e->dirty_hi) {
        // only the points the written span touches
        uint64_t lo = e->dirty_lo > phase ? (e->dirty_lo - phase) / ps : 0;
        uint64_t hi = e->dirty_hi > phase ? (e->dirty_hi - phase + ps - 1) / ps : 0;
        if (hi > ext_pts) hi = ext_pts;
        if (even) {
            // whole elements (an element whose tail has not been loaded yet is converted when the load that completes it dirties it)
            const uint64_t elo = lo / 8, ehi = hi / 8 < ext_pts / 8 ? (hi + 7) / 8 : ext_pts / 8;
            if (elo < ehi)
                BLZ_TRY(h->eng.points_to_mont_even((const char*)e->raw + phase + elo * 8 * ps, (char*)e->mont + elo * 4 * mp, (uint32_t)((ehi - elo) * 4)));
        } else if (lo < hi) {
            BLZ_TRY(h->eng.points_to_mont((const char*)e->raw + phase + lo * ps, (char*)e->mont + lo * mp, (uint32_t)(hi - lo)));
        }
        BLZ_HIP(hipEventRecord(e->shadow_ready, h->eng.stream), BLZ_ERR_UNKNOWN);
        e->shadow_recorded = true;
        e->dirty_lo = e->dirty_hi = 0;
    }
    *out = (const char*)e->mont + (even ? first / 8 * 4 : first) * mp;
    if (even) h->pc_info[3] = e->mont_bytes;
    else if (!wants_table_mode(h)) BLZ_TRY(arena_diet_step(A, *e, ps, h->eng.stream));   // (a table is tabulated from the raw bytes: no diet under such a handle)
    return BLZ_OK;
}

// Checked-table plan (msm_impl.hip.hpp k_check_precompute): is the x8 table of the `nelem` elements at arena offset `pos` what
// precompute_base_* produces?  Answered once per (extent contents, range): the check runs on this handle's main stream (969 /
// 2275 multiply-adds per Jacobian doubling, 224 doublings per element: 0.68 s for 2^26 BN254 elements, 1.33 s for BLS - 68 / 83 %
// of the bare multiply-add rate; XYZZ doublings, same box: 0.84 / 1.71 s) and the caller waits
// for it - with the arena unlocked; the answer is committed only if no write reached the extent in the meantime (epoch).
int arena_precompute_check(blz_msm* h, uint64_t pos, uint32_t nelem, bool* ok, uint64_t* checked_elems = nullptr) {
    *ok = false;
    if (checked_elems) *checked_elems = nelem;
    const size_t ps = point_size(h);
    const size_t len = (size_t)nelem * 8 * ps;
    Arena& A = arena_for(h->device);
    uint64_t epoch = 0, first = 0;
    uint32_t phase = 0;
    uint32_t* flag = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool partial = false;          // only the elements a write touched since the table was found consistent
    uint64_t chk_elems = nelem;
    {
        std::lock_guard<std::mutex> lk(A.mu);
        ArenaExtent* e = arena_find(A, pos, len);
        if (!e)
            return fail(BLZ_ERR_INVALID_PARAM, "HBM bases: no loaded extent covers [%llu, +%zu) on device %d", (unsigned long long)pos, len, h->device);
        phase = (uint32_t)((pos - e->start) % ps);
        first = (pos - e->start - phase) / ps;
        h->pc_info[1] = 0;
        if (first & 7u) {
            BLZ_LOG(1, "precompute plan: the task's bases do not start on the extent's element grid (point %llu): exact path", (unsigned long long)first);
            return BLZ_OK;
        }
        const ArenaExtent::PrecompCheck& C = e->pcheck;
        const bool covered = C.state != 0 && C.curve == h->curve && C.phase == phase && first >= C.first && first + (uint64_t)nelem * 8 <= C.first + C.npts;
        if (covered && C.state != 3) {
            *ok = C.state == 1;
            if (checked_elems) *checked_elems = C.npts / 8;
            h->pc_info[1] = (uint64_t)C.state;
            h->pc_info[2] = (uint64_t)(C.ms * 1000.0f);
            return BLZ_OK;
        }
        uint64_t chk_pos = pos;
        if (covered) {
            // state 3: the elements of the checked range that the writes since then touched
            uint64_t plo = C.redo_lo > phase ? (C.redo_lo - phase) / ps : 0, phi = C.redo_hi > phase ? (C.redo_hi - phase + ps - 1) / ps : 0;
            uint64_t elo = plo / 8, ehi = (phi + 7) / 8;
            if (elo < C.first / 8) elo = C.first / 8;
            if (ehi > (C.first + C.npts) / 8) ehi = (C.first + C.npts) / 8;
            partial = true;
            chk_elems = ehi > elo ? ehi - elo : 0;
            chk_pos = e->start + phase + elo * 8 * ps;
        }
        if (!A.build_flags && hipMalloc((void**)&A.build_flags, 256 * sizeof(uint32_t)) != hipSuccess) {
            (void)hipGetLastError();
            A.build_flags = nullptr;
            BLZ_LOG(1, "precompute plan: no memory for the check's flag: exact path");
            return BLZ_OK;
        }
        BLZ_TRY(arena_restore_raw(A, *e, h->eng.stream));   // (the check reads the raw bytes)
        flag = A.build_flags + (A.build_flag_next++ & 255u);
        epoch = e->epoch;
        hipStream_t st = h->eng.stream;
        if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess) {
            if (t0) (void)hipEventDestroy(t0);
            return fail(BLZ_ERR_UNKNOWN, "event creation failed");
        }
        int rc = BLZ_OK;
        if (hipMemsetAsync(flag, 0, 4, st) != hipSuccess || hipEventRecord(t0, st) != hipSuccess) rc = fail(BLZ_ERR_UNKNOWN, "precompute check: enqueue failed");
        if (rc == BLZ_OK) rc = h->eng.check_precompute((const char*)e->raw + (chk_pos - e->start), chk_elems, flag, st);
        if (rc == BLZ_OK && hipEventRecord(t1, st) != hipSuccess) rc = fail(BLZ_ERR_UNKNOWN, "precompute check: enqueue failed");
        if (rc != BLZ_OK) {
            (void)hipEventDestroy(t0);
            (void)hipEventDestroy(t1);
            return rc;
        }
    }
    // (the raw bytes cannot go away under the kernel: whoever frees or moves an extent drains the device first)
    uint32_t flag_h = 1;
    wait_clear();
    int rc = sync_event_bounded(t1, "precompute plan: table check");
    if (rc != BLZ_OK && wait_timed_out()) h->wedged = true;
    float ms = 0;
    if (rc == BLZ_OK) {
        (void)hipEventElapsedTime(&ms, t0, t1);
        if (hipMemcpy(&flag_h, flag, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(BLZ_ERR_READ, "precompute check: flag read failed");
    }
    if (rc == BLZ_OK || !wait_timed_out()) {
        (void)hipEventDestroy(t0);
        (void)hipEventDestroy(t1);
    }
    BLZ_TRY(rc);
    std::lock_guard<std::mutex> lk(A.mu);
    ArenaExtent* e = arena_find(A, pos, len);
    if (!e || e->epoch != epoch) {
        BLZ_LOG(1, "precompute plan: the extent was written while its table was being checked: exact path for this task");
        return BLZ_OK;
    }
    ArenaExtent::PrecompCheck& C = e->pcheck;
    C.state = flag_h ? 2 : 1;
    C.curve = h->curve;
    C.phase = phase;
    if (!partial) {   // (a partial check confirms - or refutes - the range that had been checked before)
        C.first = first;
        C.npts = (uint64_t)nelem * 8;
    }
    C.redo_lo = C.redo_hi = 0;
    C.ms = ms;
    *ok = flag_h == 0;
    if (checked_elems) *checked_elems = C.npts / 8;
    h->pc_info[1] = (uint64_t)C.state;
    h->pc_info[2] = (uint64_t)(ms * 1000.0f);
    BLZ_LOG(1, "precompute plan: %s%llu elements checked: the table %s (%.1f ms)", partial ? "rewritten span, " : "", (unsigned long long)chk_elems,
            flag_h ? "is NOT B_j = 2^32 B_(j-1) over on-curve bases: exact path (8n points, 32-bit chunks)" : "is consistent: 4n even bases, 64-bit chunks", ms);
    return BLZ_OK;
}

// Window table of the `npts` bases at arena offset `pos` (msm_impl.hip.hpp k_build_window_table, MsmPlan::table), kept with the
// extent (a small rewrite has its rows re-tabulated below, a large one drops it: arena.hip arena_write).  *out stays null - and the task takes the plain path - while the table is not to be
// had: it is still being built, there is no memory for it, a base has even order, or the task is over a sub-range whose
// best window width is not the table's.
//
// The build is never one blocking lump inside a task (round 3 built synchronously inside the first task's launch: 3.1 s for
// 2^26 bases in a call a host expects to take milliseconds), and it does not run BESIDE the tasks either - measured
// (profiles/r04_window_table_async.txt): on a lowest-priority stream its long-lived waves hold their registers and halve the
// tasks' speed for as long as it takes; confined to a quarter of the CUs it is worse (the accumulation's blocks on the shared CUs
// issue behind the build's older waves and become the kernel's tail).  So the build is PACED by the tasks: it is cut into chunks
// of TABLE_BUILD_CHUNK bases (~5.5 ms of the chip), every task launched over the bases first enqueues `chunk_budget` of them on
// its own main stream - a fixed, small surcharge per task while the table is being built - and keeps taking the plain path;
// the first task launched after the last chunk has completed adopts the table.  blz_msm_prepare_window_table enqueues ALL the
// remaining chunks at once for a host that would rather pay the build now.  Results are bit-identical either way
// (tests/test_gpu_msm_table.py).
constexpr uint32_t TABLE_BUILD_CHUNK = 3u << 16;   // bases per launch = the build kernel's lanes (msm_impl.hip.hpp TABLE_BUILD_BLOCKS x 64)
constexpr int TABLE_CHUNKS_PER_TASK = 4;           // ~22 ms on top of a 2^26 task's 117: 86 tasks until a 2^26 table is there
int arena_points_table(blz_msm* h, uint64_t pos, uint32_t npts, const void** out, int* c_out, int chunk_budget) {
    *out = nullptr;
    *c_out = 0;
    const size_t ps = point_size(h), mp = mont_point_bytes(h->curve);
    const size_t len = (size_t)npts * ps;
    Arena& A = arena_for(h->device);
    std::lock_guard<std::mutex> lk(A.mu);
    ArenaExtent* e = arena_find(A, pos, len);
    if (!e)
        return fail(BLZ_ERR_INVALID_PARAM, "HBM bases: no loaded extent covers [%llu, +%zu) on device %d",
                    (unsigned long long)pos, len, h->device);
    if (npts == 0) return BLZ_OK;
    BLZ_TRY(arena_restore_raw(A, *e, h->eng.stream));   // (tables are tabulated from the raw bytes)
    const uint32_t phase = (uint32_t)((pos - e->start) % ps);
    const uint64_t first = (pos - e->start - phase) / ps;
    const int fmt = h->eng.format_id();
    const int lo = h->range_hi ? h->range_lo : 0, hi = h->range_hi ? h->range_hi : 256;
    const int need = hi - lo < 256 ? hi - lo + 1 : 257;
    const int want_c = table_window_bits(npts, need);
    if (want_c == 0) return BLZ_OK;
    ArenaExtent::TableBuild& B = e->build;
    // the chunks this launch owes the build in flight (chained through B.done: the chunks share the scratch rows)
    auto enqueue_chunks = [&](int budget) -> int {
        hipStream_t st = h->eng.stream;
        const uint64_t nchunks = (B.npts + TABLE_BUILD_CHUNK - 1) / TABLE_BUILD_CHUNK;
        if (B.next_chunk >= nchunks || budget == 0) return BLZ_OK;
        // (the scratch rows are the arena's: chunks of every build on the device run one after the other)
        if (!A.scratch_event) BLZ_HIP(hipEventCreateWithFlags(&A.scratch_event, hipEventDisableTiming), BLZ_ERR_UNKNOWN);
        if (A.scratch_recorded) BLZ_HIP(hipStreamWaitEvent(st, A.scratch_event, 0), BLZ_ERR_UNKNOWN);
        for (; B.next_chunk < nchunks && budget != 0; ++B.next_chunk, --budget) {
            const uint64_t p0 = B.next_chunk * TABLE_BUILD_CHUNK;
            const uint32_t cnt = (uint32_t)(B.npts - p0 < TABLE_BUILD_CHUNK ? B.npts - p0 : TABLE_BUILD_CHUNK);
            if (B.next_chunk == 0) BLZ_HIP(hipEventRecord(B.t0, st), BLZ_ERR_UNKNOWN);
            BLZ_TRY(h->eng.build_table((const char*)e->raw + B.phase + (B.first + p0) * ps, (char*)B.tab + p0 * (size_t)B.W * mp, cnt, B.c, B.W, B.lo,
                                       A.build_scratch, B.flag, st));
        }
        BLZ_HIP(hipEventRecord(B.done, st), BLZ_ERR_UNKNOWN);
        B.recorded = true;
        BLZ_HIP(hipEventRecord(A.scratch_event, st), BLZ_ERR_UNKNOWN);
        A.scratch_recorded = true;
        return BLZ_OK;
    };
    hipError_t bq = hipErrorNotReady;
    if (B.tab) {
        const uint64_t nchunks = (B.npts + TABLE_BUILD_CHUNK - 1) / TABLE_BUILD_CHUNK;
        if (B.next_chunk >= nchunks && B.recorded) {
            bq = hipEventQuery(B.done);
            if (bq != hipSuccess && bq != hipErrorNotReady) { (void)hipGetLastError(); return fail(BLZ_ERR_UNKNOWN, "window table build failed: %s", hipGetErrorString(bq)); }
        }
    }
    if (B.tab && bq == hipSuccess) {
        // a build has completed: adopt its table
        uint32_t flag_h = 0;
        float ms = 0;
        BLZ_HIP(hipMemcpy(&flag_h, B.flag, 4, hipMemcpyDeviceToHost), BLZ_ERR_READ);   // (the build is complete: nothing to wait for)
        (void)hipEventElapsedTime(&ms, B.t0, B.done);   // first chunk .. last chunk, the tasks in between included
        if (flag_h) {
            BLZ_LOG(1, "window table: a base has a multiple at infinity (a point of even order): plain path for this extent");
            BLZ_TRY(sync_device_bounded("dropping a refused window table"));   // (hipFree waits for the device: bounded first)
            (void)hipFree(B.tab);
            (void)hipEventDestroy(B.done);
            (void)hipEventDestroy(B.t0);
            B = ArenaExtent::TableBuild();
            e->table_refused = true;
            return BLZ_OK;
        }
        ArenaExtent::WindowTable t;
        t.p = B.tab;
        t.bytes = B.bytes;
        t.format = B.format;
        t.phase = B.phase;
        t.first = B.first;
        t.npts = B.npts;
        t.c = B.c;
        t.W = B.W;
        t.lo = B.lo;
        t.hi = B.hi;
        t.build_ms = ms;
        e->tables.push_back(t);
        B.tab = nullptr;
        (void)hipEventDestroy(B.done);
        (void)hipEventDestroy(B.t0);
        B = ArenaExtent::TableBuild();
        BLZ_LOG(1, "window table: %llu bases x %d windows of %d bits, %.1f MiB, complete %.1f ms after its first chunk", (unsigned long long)t.npts,
                t.W, t.c, t.bytes / 1048576.0, ms);
    }
    // rows of bases that were rewritten since the tables were built (arena_write: small rewrites keep the tables): re-tabulated
    // here, on this handle's main stream, behind a drain (another handle's task may be gathering from the very rows).  A table
    // of another format than this handle's (its curve's other arithmetic), or one whose scratch rows are gone, is dropped instead.
    if (e->tab_dirty_lo < e->tab_dirty_hi && !e->tables.empty()) {
        BLZ_TRY(sync_device_bounded("window table: drain before the rewritten bases are re-tabulated"));
        uint32_t* pflag = nullptr;
        if (A.build_flags) {
            pflag = A.build_flags + (A.build_flag_next++ & 255u);
            BLZ_HIP(hipMemsetAsync(pflag, 0, 4, h->eng.stream), BLZ_ERR_UNKNOWN);
        }
        bool patched = false;
        for (size_t k = e->tables.size(); k-- > 0;) {
            ArenaExtent::WindowTable& t = e->tables[k];
            const uint64_t plo = e->tab_dirty_lo > t.phase ? (e->tab_dirty_lo - t.phase) / ps : 0;
            const uint64_t phi = e->tab_dirty_hi > t.phase ? (e->tab_dirty_hi - t.phase + ps - 1) / ps : 0;
            const uint64_t lo_p = plo > t.first ? plo : t.first, hi_p = phi < t.first + t.npts ? phi : t.first + t.npts;
            if (lo_p >= hi_p) continue;
            if (t.format != fmt || !pflag || A.build_scratch_bytes < h->eng.table_scratch_bytes(t.W) + 16) {
                (void)hipFree(t.p);
                e->tables.erase(e->tables.begin() + (long)k);
                continue;
            }
            BLZ_TRY(h->eng.build_table((const char*)e->raw + t.phase + lo_p * ps, (char*)t.p + (lo_p - t.first) * (size_t)t.W * mp, (uint32_t)(hi_p - lo_p), t.c,
                                       t.W, t.lo, A.build_scratch, pflag, h->eng.stream));
            patched = true;
        }
        if (patched) {
            uint32_t flag_h = 0;
            BLZ_WAIT(h, sync_stream_bounded(h->eng.stream, "window table: rewritten bases re-tabulated"));
            BLZ_HIP(hipMemcpy(&flag_h, pflag, 4, hipMemcpyDeviceToHost), BLZ_ERR_READ);
            if (flag_h) {
                BLZ_LOG(1, "window table: a rewritten base has a multiple at infinity (a point of even order): plain path for this extent");
                arena_drop_table(*e);   // (the stream was drained just now, the device before)
                e->table_refused = true;
                return BLZ_OK;
            }
            BLZ_LOG(1, "window table: rows of the rewritten bases re-tabulated (bytes [%llu, %llu) of the extent)", (unsigned long long)e->tab_dirty_lo,
                    (unsigned long long)e->tab_dirty_hi);
        }
        e->tab_dirty_lo = e->tab_dirty_hi = 0;
    }
    // one table per (bases, scalar range) that was asked for: the handles of a curve share it, a sub-range of its bases is
    // served from it, a handle with another scalar range gets its own (two handles evicting each other's table on every
    // launch would rebuild for ever: ADVICE r03)
    const ArenaExtent::WindowTable* T = nullptr;
    for (const auto& t : e->tables)
        if (t.format == fmt && t.phase == phase && first >= t.first && first + npts <= t.first + t.npts && t.lo == lo && t.hi == hi) T = &t;
    if (T && T->c != want_c) return BLZ_OK;   // a sub-range that wants other windows: the plain path, not a rebuild
    if (!T) {
        // (a refusal - no memory for a table, a base of even order - stops NEW builds until the next write; tables that are
        // in place keep being served, and a build in flight keeps being paced)
        if (B.tab) {
            // a build in flight: this launch pays its share if the build is for this handle's bases and range (one build at a
            // time: another's turn comes when this one is through)
            if (B.format == fmt && B.phase == phase && first >= B.first && first + npts <= B.first + B.npts && B.lo == lo && B.hi == hi)
                BLZ_TRY(enqueue_chunks(chunk_budget));
            return BLZ_OK;
        }
        if (e->table_refused) return BLZ_OK;
        if (e->tables.size() >= ArenaExtent::MAX_TABLES) {
            BLZ_LOG(1, "window table: the extent already holds %zu tables: plain path for this handle", e->tables.size());
            return BLZ_OK;
        }
        const int c = want_c, W = table_windows(c, need);
        const size_t bytes = (size_t)npts * W * mp + 16;
        size_t free_b = 0, total_b = 0;
        BLZ_HIP(hipMemGetInfo(&free_b, &total_b), BLZ_ERR_UNKNOWN);
        const size_t scratch_b = h->eng.table_scratch_bytes(W) + 16;
        // what a task of this shape still has to allocate next to the table: entries and sort intermediates (~32 B per
        // entry), bucket tables and partial sums
        const size_t workspace = (size_t)npts * W * 32 + ((size_t)1 << (c - 1)) * 256 + ((size_t)1 << 30);
        if (free_b < bytes + scratch_b + workspace) {
            BLZ_LOG(1, "window table: %zu bytes for %u bases (c = %d, %d windows) do not fit beside the workspace (%zu free): plain path",
                    bytes, npts, c, W, free_b);
            e->table_refused = true;
            return BLZ_OK;
        }
        // the build's scratch rows belong to the arena and are kept (freeing them would wait for every task in flight)
        if (A.build_scratch_bytes < scratch_b) {
            if (A.build_scratch) {
                BLZ_TRY(sync_device_bounded("growing the window-table scratch"));
                (void)hipFree(A.build_scratch);
                A.build_scratch = nullptr;
                A.build_scratch_bytes = 0;
            }
            if (hipMalloc(&A.build_scratch, scratch_b) != hipSuccess) {
                (void)hipGetLastError();
                A.build_scratch = nullptr;
                e->table_refused = true;
                return BLZ_OK;
            }
            A.build_scratch_bytes = scratch_b;
        }
        void* tab = nullptr;
        const auto t_alloc = std::chrono::steady_clock::now();
        const hipError_t he_tab = hipMalloc(&tab, bytes);
        BLZ_LOG(1, "window table: hipMalloc(%zu) took %.1f ms", bytes,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_alloc).count());
        if (he_tab != hipSuccess) {
            (void)hipGetLastError();
            e->table_refused = true;
            return BLZ_OK;
        }
        hipEvent_t done = nullptr, t0 = nullptr;
        int rc = BLZ_OK;
        if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&done) != hipSuccess) rc = fail(BLZ_ERR_UNKNOWN, "event creation failed");
        // the build's "a multiple came out as infinity" flag: its own slot (builds share the scratch rows in stream order, but a
        // flag is read by the host when its build is ADOPTED, possibly after a later build has started)
        if (!A.build_flags && hipMalloc((void**)&A.build_flags, 256 * sizeof(uint32_t)) != hipSuccess) {
            (void)hipGetLastError();
            A.build_flags = nullptr;
            (void)hipFree(tab);
            if (t0) (void)hipEventDestroy(t0);
            if (done) (void)hipEventDestroy(done);
            e->table_refused = true;
            return BLZ_OK;
        }
        uint32_t* flag = A.build_flags + (A.build_flag_next++ & 255u);
        if (rc == BLZ_OK && hipMemsetAsync(flag, 0, 4, h->eng.stream) != hipSuccess) rc = fail(BLZ_ERR_UNKNOWN, "memset failed");
        if (rc != BLZ_OK) {
            if (t0) (void)hipEventDestroy(t0);
            if (done) (void)hipEventDestroy(done);
            (void)hipFree(tab);
            return rc;
        }
        B.tab = tab;
        B.flag = flag;
        B.bytes = bytes;
        B.done = done;
        B.t0 = t0;
        B.format = fmt;
        B.c = c;
        B.W = W;
        B.lo = lo;
        B.hi = hi;
        B.phase = phase;
        B.first = first;
        B.npts = npts;
        B.next_chunk = 0;
        B.recorded = false;
        BLZ_LOG(1, "window table: build of %u bases x %d windows of %d bits started (%.1f MiB, %u chunks); tasks take the plain path until it is there",
                npts, W, c, bytes / 1048576.0, (unsigned)((npts + TABLE_BUILD_CHUNK - 1) / TABLE_BUILD_CHUNK));
        return enqueue_chunks(chunk_budget);   // (the raw bases are in place: every arena write ends with a host-side wait)
    }
    *out = (const char*)T->p + (first - T->first) * (size_t)T->W * mp;
    *c_out = T->c;
    h->table_info[0] = T->bytes;
    h->table_info[1] = (uint64_t)T->c;
    h->table_info[2] = (uint64_t)T->W;
    h->table_info[3] = (uint64_t)(T->build_ms * 1000.0f);
    return BLZ_OK;
}

// (BN254 loses with a table - 64-byte points: its accumulation is already at the gather rate, 2^26 71.8 -> 74.6 ms - so mode 1,
// "where it pays", leaves it on the plain path)
bool wants_table(const blz_msm* h) {
    return h->pf == 1 && (h->window_table == 2 || (h->window_table == 1 && h->curve != BLZ_BN254));
}

// BN254 has two arithmetics (msm_engine.hpp `repr`): the 9 x 29-bit reduced radix wins while the accumulation is bound by its
// multiplier, 32-bit limbs win once it is bound by the memory system's rate of random line gathers out of a copy far larger than
// the caches (profiles/r05_tlb_probe.txt).  The exact path of a precompute
// handle always runs on 32-bit limbs (2^29 bases, 32 GiB); the plan's even-base copy is a quarter of that per element, so it
// takes the reduced radix up to 2^25 elements (8 GiB of even bases) and 32-bit limbs above - measured, same box, ms per MSM in a
// stream of tasks, reduced radix / 32-bit limbs: 2^20 1.85 / 2.20, 2^22 6.60 / 7.49, 2^24 18.2 / 19.2, 2^26 74.8 / 70.0
// (exact path: 2.2, 7.2, 24.6, 92.5).  Decided by the size of the CHECKED table, not of the task (tasks over sub-ranges of one table
// would otherwise flip the arithmetic - and with it the format of the extent's copy - from task to task).  BLAZE_MSM_PLAN
// pc_repr=0|1 forces one (tests).  Switched only while nothing of the handle is in flight.
int plan_repr_bn254(uint64_t nelem) {
    const int forced = plan_override("pc_repr", -1);
    if (forced == 0 || forced == 1) return forced;
    return nelem > (1ull << 25) ? 1 : 0;
}

// Which task serves `n` elements whose bases sit in the arena at `pos`: a precompute handle on the checked-table plan whose
// table is consistent sums 4n even bases over 64-bit chunks; a pf = 1 handle with a window table in place gathers from it;
// everything else is the plain task over the Montgomery copy.  Resolves h->d_points_mont (shadow pointers are resolved when
// the task is launched, not when its data was staged: a load by another handle in between may have moved or re-converted
// the extent).
int resolve_arena_task(blz_msm* h, uint64_t pos, uint32_t n, bool allow_table, bool allow_plan, uint32_t* npts, int* sbits, int* table_c) {
    *npts = n * h->pf;
    *sbits = h->pf == 1 ? 256 : 32;
    *table_c = 0;
    memset(h->table_info, 0, sizeof(h->table_info));
    memset(h->pc_info, 0, sizeof(h->pc_info));
    if (h->pf == BLZ_PRECOMPUTE_FACTOR && h->precompute_plan && allow_plan && n > 0) {
        bool ok = false;
        uint64_t checked = n;
        BLZ_TRY(arena_precompute_check(h, pos, n, &ok, &checked));
        if (h->curve == BLZ_BN254 && h->in_flight.empty()) h->eng.repr = exp_knob("BLAZE_BN254_REPR", ok ? plan_repr_bn254(checked) : 1) ? 1 : 0;
        if (ok && h->eng.plan_for(n * 4, 64).c != 0) {
            const void* even = nullptr;
            BLZ_TRY(arena_points_mont(h, pos, n * 8, &even, true));
            if (even) {
                h->d_points_mont = even;
                *npts = n * 4;
                *sbits = 64;
                h->pc_info[0] = 1;
                return BLZ_OK;
            }
            // (the extent was written between the check and now: this task takes the exact path, the next one checks again)
            h->pc_info[1] = 0;
            if (h->curve == BLZ_BN254 && h->in_flight.empty()) h->eng.repr = exp_knob("BLAZE_BN254_REPR", 1) ? 1 : 0;
        }
    }
    if (allow_table && wants_table(h)) {
        const void* tab = nullptr;
        BLZ_TRY(arena_points_table(h, pos, *npts, &tab, table_c, TABLE_CHUNKS_PER_TASK));
        if (tab) h->d_points_mont = tab;
        else *table_c = 0;
    }
    if (!*table_c) BLZ_TRY(arena_points_mont(h, pos, *npts, &h->d_points_mont));
    return BLZ_OK;
}

int launch_if_ready(blz_msm* h) {
    if (!(h->armed && h->data_ready)) return BLZ_OK;
    if (!h->eng.can_accept())
        return fail(BLZ_ERR_INVALID_PARAM, "task queue full (%d in flight); call wait_result first", MSM_QUEUE_DEPTH);
    uint32_t npts = h->staged_n * h->pf;
    int sbits = h->pf == 1 ? 256 : 32;
    int slot = 0;
    int table_c = 0;
    memset(h->table_info, 0, sizeof(h->table_info));
    memset(h->pc_info, 0, sizeof(h->pc_info));
    // (a task that has just loaded its own table - set_data mode iii, msm_api.rs:203-216 - is a DMA-mode task as far as the plan is
    // concerned: a check per task would cost more than it saves)
    if (h->staged_from_arena) BLZ_TRY(resolve_arena_task(h, h->staged_arena_pos, h->staged_n, true, !h->staged_loaded_now, &npts, &sbits, &table_c));
    h->eng.inputs_event = h->staged_set >= 0 ? h->set_free[h->staged_set] : nullptr;
    BLZ_TRY(h->eng.run(h->d_points_mont, h->d_scalars, npts, sbits, &slot, table_c, h->range_lo, h->range_hi));
    if (h->staged_set >= 0) h->set_used[h->staged_set] = true;
    h->staged_set = -1;
    h->armed = false;
    h->data_ready = false;
    h->in_flight.push_back({slot, h->task_label});
    return BLZ_OK;
}

int stage_common(blz_msm* h, bool have_points, const void* points, size_t points_len, const void* scalars,
                 size_t scalars_len, uint32_t n, int has_hbm, uint64_t hbm_addr, uint64_t hbm_off, bool on_device) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    BLZ_LIVE(h);
    BLZ_TRY(use_device(h->device));
    if (!have_points && !has_hbm) return BLZ_OK;  // reference: falls through every branch (msm_api.rs:163-216)
    if (has_hbm) BLZ_ARENA_ADDR(hbm_addr, hbm_off);
    if (!scalars && n) return fail(BLZ_ERR_INVALID_PARAM, "null scalars");
    if (scalars_len != (size_t)n * BLZ_SCALAR_SIZE)
        return fail(BLZ_ERR_INVALID_PARAM, "scalars length %zu != nof_elements %u * 32", scalars_len, n);
    size_t want_pts = (size_t)n * h->pf * point_size(h);
    if (have_points && points_len != want_pts)
        return fail(BLZ_ERR_INVALID_PARAM, "points length %zu != nof_elements %u * precompute_factor %u * %zu", points_len,
                    n, h->pf, point_size(h));
    if ((uint64_t)n * h->pf >= (1ull << 31)) return fail(BLZ_ERR_INVALID_PARAM, "too many points");
    // refuse sizes the window planner cannot serve BEFORE anything is copied or converted (u32 entry indexing:
    // points x windows <= 2^32 - 2^26 (msm_engine.hpp MSM_MAX_ENTRIES) with windows of at most 23 bits - 256-bit scalars need 12,
    // so pf = 1 stops at 352 321 536 points (2^28.39; checked there: tests/test_gpu_msm.py), the 32-bit chunks of pf = 8 at 2^31 - 2^25)
    if (n && h->eng.plan_for(n * h->pf, h->pf == 1 ? 256 : 32).c == 0)
        return fail(BLZ_ERR_INVALID_PARAM, "no window plan for %llu points of %d-bit scalars (u32 entry indexing: at most 352321536 points at pf = 1, 2113929216 at pf = 8)",
                    (unsigned long long)n * h->pf, h->pf == 1 ? 256 : 32);
    if (!h->eng.can_accept())
        return fail(BLZ_ERR_INVALID_PARAM, "task queue full (%d in flight); call wait_result first", MSM_QUEUE_DEPTH);
    hipStream_t st = h->eng.stream;
    // Host buffers are staged on their own stream, so the PCIe transfer of this task overlaps the
    // accumulation of the task in flight (the reference's DMA writes overlap device compute the same
    // way).
    hipStream_t cst = h->copy_stream;
    uint32_t npts = n * h->pf;
    // this task's staging set was last used two tasks ago: its to-Montgomery pass and digit sort must have read
    // it before the new copies land (an event on the main stream, not a host wait; normally long past)
    const int set = h->stage_idx;
    if (!on_device) {
        if (h->set_used[set]) BLZ_HIP(hipStreamWaitEvent(cst, h->set_free[set], 0), BLZ_ERR_UNKNOWN);
        h->stage_idx ^= 1;
        h->staged_set = set;
    } else {
        h->staged_set = -1;
    }

    h->staged_loaded_now = have_points && has_hbm;
    if (have_points && has_hbm) {
        // msm_api.rs:203-206: load_data_to_hbm(points, addr, offset) first
        BLZ_WAIT(h, arena_write(h->device, hbm_addr + hbm_off, points, points_len, on_device, st));
        h->bases_from_hbm = true;
        h->hbm_addr = hbm_addr;
    }
    if (has_hbm) {
        // bases come from the arena.  The reference's initialize() programs only hbm_point_addr.0
        // as the start address (msm_api.rs:84-95) while load_data_to_hbm writes at addr+offset
        // (msm_api.rs:312); both tests use offset 0.  Here the task reads where the load wrote.
        {
            Arena& A = arena_for(h->device);
            std::lock_guard<std::mutex> lk(A.mu);
            if (!arena_find(A, hbm_addr + hbm_off, (size_t)npts * point_size(h)))
                return fail(BLZ_ERR_INVALID_PARAM, "HBM bases: no loaded extent covers [%llu, +%zu) on device %d",
                            (unsigned long long)(hbm_addr + hbm_off), (size_t)npts * point_size(h), h->device);
        }
        h->staged_from_arena = true;
        h->staged_arena_pos = hbm_addr + hbm_off;
    } else {
        h->staged_from_arena = false;
    }
    // Host buffers with a task already armed (DMA mode, the reference's primary flow: tests/integration_msm.rs:149-207):
    // the task is enqueued PIECE BY PIECE while its data crosses the link, the way the reference streams interleaved chunks
    // of scalars and points into the card's FIFOs while the card computes (msm_api.rs:175-202).  Per piece: its scalars,
    // then its sort stage goes to the device; its points, then their to-Montgomery pass and the piece's accumulation
    // (MsmEngine::begin / sort_slice / accumulate_slice / end: the pieces share one bucket space and the bucket sums are
    // carried from piece to piece).  Link and multiplier work at the same time; what is left on the critical path behind the
    // last byte is the last piece's accumulation, the bucket reduce and the tail.
    // The reference's HBM flow (bases resident in the arena, the scalars a host Vec<u8> with every task:
    // tests/integration_msm_hbm.rs:57-100) goes the same way when the handle is idle: a lone task's 2 GiB of scalars would
    // otherwise cross the link with the chip doing nothing (38 of 163 ms at 2^26); in a stream of tasks the whole upload
    // already hides under the previous task's accumulation, and the task keeps its one-piece form (hidden sort, no
    // carried sums).
    int sbits = h->pf == 1 ? 256 : 32;
    const bool dma_pieces = !on_device && !has_hbm && h->armed && npts > 0 && exp_knob("BLAZE_DMA_OVERLAP", 1) != 0;
    const bool hbm_pieces = !on_device && has_hbm && h->armed && npts > 0 && (npts >= (1u << 22) || env_int("BLAZE_MSM_PIECES", 0) > 1) &&
                            h->in_flight.empty() && !wants_table(h) &&
                            exp_knob("BLAZE_DMA_OVERLAP", 1) != 0;
    if (dma_pieces || hbm_pieces) {
        const size_t mp = mont_point_bytes(h->curve), ps = point_size(h);
        BLZ_TRY(h->scalars_buf[set].reserve(scalars_len));
        const void* arena_mont = nullptr;
        memset(h->table_info, 0, sizeof(h->table_info));
        memset(h->pc_info, 0, sizeof(h->pc_info));
        if (dma_pieces) {
            BLZ_TRY(h->points_raw[set].reserve(want_pts));
            BLZ_TRY(h->points_mont.reserve((size_t)npts * mp));
        } else {
            // (stale spans are converted on the main stream; a precompute handle on the checked-table plan: 4n even bases, 64-bit chunks)
            int tc = 0;
            BLZ_TRY(resolve_arena_task(h, h->staged_arena_pos, n, false, !h->staged_loaded_now, &npts, &sbits, &tc));
            arena_mont = h->d_points_mont;
        }
        const size_t sb = (size_t)sbits / 8;
        // pieces of >= 2^19 points with their scalars (64 MiB of host bytes: 1.2 ms of link), at most 16.
        // Measured (profiles/r04_dma_pieces.txt): 2^22 elements 22.6 ms in one piece, 16.2 / 15.35 / 17.1 in 4 / 8 / 16; 2^26
        // 270.8, 191.8 / 178.3 / 171.5
        int pieces = env_int("BLAZE_MSM_PIECES", 0);   // (the same switch forces the piece count of device-resident tasks, msm.hip run())
        if (pieces <= 0) {
            if (dma_pieces) {
                pieces = (int)(npts >> 19);
                if (npts >= (1u << 20) && npts <= (1u << 21)) pieces = (int)(npts >> 18);   // 2^20: 5.49 ms in 2 pieces, 5.23 in 4; 2^21: 8.47 in 4, 8.25 in 8
                if (pieces > 16) pieces = 16;
                // with another task in flight the link is the bound whatever the pieces do, and every piece costs it the
                // ~150 us of launches between two copies: fewer, larger pieces (2^22: 10.4 against 10.8 ms per MSM)
                if (!h->in_flight.empty() && pieces > 4) pieces = 4;
            } else {
                // scalars alone: the link is a quarter of the task, and every piece pays the sort stage's passes over the
                // bucket space again (not hidden here) - 2^26: 163.7 ms whole, 158.5 / 145.4 / 181.8 in 16 / 8 / 32 pieces
                // (2^22 .. 2^24 lone tasks: 14.25 / 25.6 / 46.2 ms whole, 13.3 / 23.6 / 42.5 in two pieces, 12.7 / 22.5 / 40.6 in four)
                pieces = (int)(npts >> 23);
                if (pieces > 8) pieces = 8;
                if (pieces < 4) pieces = 4;
            }
        }
        if (pieces < 1) pieces = 1;
        int slot = -1;
        h->eng.inputs_event = h->set_free[set];
        BLZ_TRY(h->eng.begin(npts, sbits, &slot, 0, h->range_lo, h->range_hi, pieces, true));
        const uint32_t per = h->eng.slots[slot].pts_per_slice;
        pieces = h->eng.slots[slot].slices;
        int rc = BLZ_OK;
        auto copy_in = [&](void* dst, const void* src, size_t len, const char* what) -> int {
            if (hipMemcpyAsync(dst, src, len, hipMemcpyHostToDevice, cst) != hipSuccess) return fail(BLZ_ERR_WRITE, "%s failed", what);
            // the caller may drop its buffers as soon as we return (set_data is synchronous: utils.rs:71), and the piece's
            // device work is enqueued when its bytes are there.  The first copy waits for the staging set's previous user
            // (set_free, two tasks back): bounded like every wait
            wait_clear();
            const int r = sync_stream_bounded(cst, what);
            if (r != BLZ_OK && wait_timed_out()) h->wedged = true;
            return r;
        };
        for (int k = 0; k < pieces && rc == BLZ_OK; ++k) {
            const uint32_t p0 = (uint32_t)k * per;
            const uint32_t np = npts - p0 < per ? npts - p0 : per;
            char* d_sc = (char*)h->scalars_buf[set].p + (size_t)p0 * sb;
            rc = copy_in(d_sc, (const char*)scalars + (size_t)p0 * sb, (size_t)np * sb, "set_data: host -> device copy of the scalars");
            if (rc == BLZ_OK) rc = h->eng.sort_slice(slot, k, d_sc, np);
            if (dma_pieces) {
                char* d_raw = (char*)h->points_raw[set].p + (size_t)p0 * ps;
                char* d_mont = (char*)h->points_mont.p + (size_t)p0 * mp;
                if (rc == BLZ_OK) rc = copy_in(d_raw, (const char*)points + (size_t)p0 * ps, (size_t)np * ps, "set_data: host -> device copy of the points");
                if (rc == BLZ_OK) rc = h->eng.points_to_mont(d_raw, d_mont, np);
                if (rc == BLZ_OK) rc = h->eng.accumulate_slice(slot, k, d_mont);
            } else if (rc == BLZ_OK) {
                rc = h->eng.accumulate_slice(slot, k, (const char*)arena_mont + (size_t)p0 * mp);
            }
        }
        if (rc == BLZ_OK) rc = h->eng.end(slot);
        if (rc != BLZ_OK) {
            h->eng.abandon(slot);
            return rc;
        }
        h->d_scalars = h->scalars_buf[set].p;
        h->d_points_mont = dma_pieces ? h->points_mont.p : arena_mont;
        h->staged_n = n;
        h->set_used[set] = true;
        h->staged_set = -1;
        h->armed = false;
        h->data_ready = false;
        h->in_flight.push_back({slot, h->task_label});
        return BLZ_OK;
    }
    // Everything else is staged whole: the scalars first ...
    if (on_device) {
        if (((uintptr_t)scalars) % 16) return fail(BLZ_ERR_INVALID_PARAM, "device scalars must be 16-byte aligned");
        h->d_scalars = scalars;
    } else {
        BLZ_TRY(h->scalars_buf[set].reserve(scalars_len ? scalars_len : 16));
        if (scalars_len) BLZ_HIP(hipMemcpyAsync(h->scalars_buf[set].p, scalars, scalars_len, hipMemcpyHostToDevice, cst), BLZ_ERR_WRITE);
        h->d_scalars = h->scalars_buf[set].p;
        // the caller may drop its buffers as soon as we return (set_data is synchronous: utils.rs:71).  The copy waits
        // for the staging set's previous user (set_free, two tasks back): bounded like every wait
        BLZ_WAIT(h, sync_stream_bounded(cst, "set_data: host -> device copy of the scalars"));
    }
    h->staged_n = n;
    if (!has_hbm) {
        // ... then the points, converted to Montgomery form on the main stream
        const size_t mp = mont_point_bytes(h->curve);
        const size_t want_mont = (size_t)npts * mp;
        BLZ_TRY(h->points_mont.reserve(want_mont ? want_mont : 16));
        if (on_device) {
            if (((uintptr_t)points) % 16) return fail(BLZ_ERR_INVALID_PARAM, "device points must be 16-byte aligned");
            BLZ_TRY(h->eng.points_to_mont(points, h->points_mont.p, npts));
        } else {
            BLZ_TRY(h->points_raw[set].reserve(want_pts ? want_pts : 16));
            if (want_pts) BLZ_HIP(hipMemcpyAsync(h->points_raw[set].p, points, want_pts, hipMemcpyHostToDevice, cst), BLZ_ERR_WRITE);
            BLZ_WAIT(h, sync_stream_bounded(cst, "set_data: host -> device copy of the points"));
            BLZ_TRY(h->eng.points_to_mont(h->points_raw[set].p, h->points_mont.p, npts));
        }
        h->d_points_mont = h->points_mont.p;
    }
    h->staged_n = n;
    h->data_ready = true;
    return launch_if_ready(h);
}

}  // namespace

extern "C" {

int blz_msm_new(int device_id, int mem_type, int is_precompute, int curve, blz_msm** out) {
    if (!out) return fail(BLZ_ERR_INVALID_PARAM, "null out");
    *out = nullptr;
    if (curve < 0 || curve > 2) return fail(BLZ_ERR_INVALID_PARAM, "unknown curve %d", curve);
    if (mem_type != BLZ_HBM && mem_type != BLZ_DMA) return fail(BLZ_ERR_INVALID_PARAM, "unknown mem_type %d", mem_type);
    BLZ_TRY(use_device(device_id));
    blz_msm* h = new blz_msm();
    h->device = device_id;
    h->mem_type = mem_type;
    h->pf = is_precompute ? BLZ_PRECOMPUTE_FACTOR : BLZ_PRECOMPUTE_FACTOR_BASE;
    h->curve = curve;
    h->window_table = 0;   // opt-in: blz_msm_set_window_table
    int rc = h->eng.init(device_id, curve, (int)h->pf);
    if (rc == BLZ_OK && hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess)
        rc = fail(BLZ_ERR_UNKNOWN, "copy stream creation failed");
    for (int i = 0; i < 2 && rc == BLZ_OK; ++i)
        if (hipEventCreateWithFlags(&h->set_free[i], hipEventDisableTiming) != hipSuccess)
            rc = fail(BLZ_ERR_UNKNOWN, "event creation failed");
    if (rc != BLZ_OK) {
        h->eng.destroy();
        delete h;
        return rc;
    }
    *out = h;
    return BLZ_OK;
}

void blz_msm_free(blz_msm* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->comm) (void)blz_msm_comm_free(h);
    if (!h->eng.destroy() || sync_stream_bounded(h->copy_stream, "free: copy stream") != BLZ_OK) {
        delete h;   // wedged device work may still touch the staging buffers: they are leaked, not freed
        return;
    }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    for (int i = 0; i < 2; ++i) {
        h->scalars_buf[i].release();
        h->points_raw[i].release();
        if (h->set_free[i]) (void)hipEventDestroy(h->set_free[i]);
    }
    h->points_mont.release();
    delete h;
}

// field value v into bits [lo, hi] of the image-parameter word, most significant bit at `lo`: the layout
// MSMImageParametrs::parse_image_params reads (msm_api.rs:333-354: reverse_bits, then packed_struct msb0 ranges)
static uint32_t put_msb_first(uint32_t v, int lo, int hi) {
    uint32_t w = 0;
    for (int k = 0, b = hi; b >= lo; ++k, --b) w |= ((v >> k) & 1u) << b;
    return w;
}

int blz_msm_loaded_binary_parameters(blz_msm* h, uint32_t out[2]) {
    if (!h || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    // [0] image id 'MI35'; [1] decodes with MSMImageParametrs::parse_image_params: is_stub 0, curve code (0 BLS12-377,
    // 1 BN254, 2 BLS12-381: the numbering debug_information implies, msm_api.rs:359-364) above two flag bits,
    // "EC adders" = compute units / 16 (saturated at the field's 15), bucket-memory address width = the widest
    // window's bucket index bits at the headline size (2^26 elements), segments = XCDs
    out[0] = 0x4D493335u;
    const uint32_t curve_code = h->curve == BLZ_BLS377 ? 0u : h->curve == BLZ_BN254 ? 1u : 2u;
    int cus = 0, dev = h->device;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    uint32_t adders = (uint32_t)cus / 16u;
    if (adders > 15u) adders = 15u;
    const MsmPlan P = h->eng.plan_for(h->pf == 1 ? (1u << 26) : (1u << 29), h->pf == 1 ? 256 : 32);
    const uint32_t width = P.c > 0 ? (uint32_t)(P.c - 1) : 0u;
    out[1] = put_msb_first(0, 28, 31) | put_msb_first((curve_code << 2) & 0xffu, 20, 27) | put_msb_first(adders, 16, 19) |
             put_msb_first(width & 0xffu, 8, 15) | put_msb_first(8u, 4, 7);
    return BLZ_OK;
}

int blz_msm_set_window_table(blz_msm* h, int enable) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    if (enable < 0 || enable > 2) return fail(BLZ_ERR_INVALID_PARAM, "window table mode %d (0 off, 1 where it pays, 2 always)", enable);
    h->window_table = enable;
    return BLZ_OK;
}

int blz_msm_prepare_window_table(blz_msm* h, uint32_t nof_elements, uint64_t hbm_addr, uint64_t hbm_off, int wait_ms, int* ready) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    BLZ_LIVE(h);
    if (ready) *ready = 0;
    BLZ_TRY(use_device(h->device));
    if (!wants_table(h) || nof_elements == 0) return BLZ_OK;
    BLZ_ARENA_ADDR(hbm_addr, hbm_off);
    const auto t0 = std::chrono::steady_clock::now();
    const int limit = wait_ms < 0 ? wait_timeout_ms() : wait_ms;
    for (;;) {
        const void* tab = nullptr;
        int c = 0;
        // starts the build / enqueues what is left of it (a waiting host has nothing better to do: all of it), or adopts a finished one
        BLZ_TRY(arena_points_table(h, hbm_addr + hbm_off, nof_elements, &tab, &c, wait_ms != 0 ? -1 : TABLE_CHUNKS_PER_TASK));
        if (tab) {
            if (ready) *ready = 1;
            return BLZ_OK;
        }
        bool building = false;
        {
            // is the build in flight THIS handle's (its bases, its scalar range)?  Another handle's build is paced by that
            // handle's tasks: waiting for it here would sit out the whole deadline (one build at a time per extent)
            Arena& A = arena_for(h->device);
            std::lock_guard<std::mutex> lk(A.mu);
            const size_t ps = point_size(h);
            const uint64_t pos = hbm_addr + hbm_off;
            ArenaExtent* e = arena_find(A, pos, (size_t)nof_elements * ps);
            if (e && e->build.tab != nullptr) {
                const ArenaExtent::TableBuild& B = e->build;
                const uint32_t phase = (uint32_t)((pos - e->start) % ps);
                const uint64_t first = (pos - e->start - phase) / ps;
                const int lo = h->range_hi ? h->range_lo : 0, hi = h->range_hi ? h->range_hi : 256;
                building = B.format == h->eng.format_id() && B.phase == phase && first >= B.first && first + nof_elements <= B.first + B.npts &&
                           B.lo == lo && B.hi == hi;
            }
        }
        if (!building) return BLZ_OK;   // refused (no memory, a base of even order), another handle's table stays, or another handle's build is in flight
        if (std::chrono::steady_clock::now() - t0 >= std::chrono::milliseconds(limit)) return BLZ_OK;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

int blz_msm_set_precompute_plan(blz_msm* h, int enable) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    if (enable != 0 && enable != 1) return fail(BLZ_ERR_INVALID_PARAM, "precompute plan mode %d (0 exact path, 1 checked-table plan)", enable);
    if (enable && h->pf != BLZ_PRECOMPUTE_FACTOR) return fail(BLZ_ERR_INVALID_PARAM, "the checked-table plan is for precompute handles (MSMInit.is_precompute)");
    h->precompute_plan = enable;
    return BLZ_OK;
}

int blz_msm_prepare_precompute_plan(blz_msm* h, uint32_t nof_elements, uint64_t hbm_addr, uint64_t hbm_off, int* consistent) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    BLZ_LIVE(h);
    if (consistent) *consistent = 0;
    BLZ_TRY(use_device(h->device));
    if (!h->precompute_plan || nof_elements == 0) return BLZ_OK;
    if ((uint64_t)nof_elements * h->pf >= (1ull << 31)) return fail(BLZ_ERR_INVALID_PARAM, "too many points");
    BLZ_ARENA_ADDR(hbm_addr, hbm_off);
    bool ok = false;
    uint64_t checked = nof_elements;
    BLZ_TRY(arena_precompute_check(h, hbm_addr + hbm_off, nof_elements, &ok, &checked));
    if (ok && h->in_flight.empty()) {
        // the even-base copy too, so that the first task finds it in place
        if (h->curve == BLZ_BN254) h->eng.repr = exp_knob("BLAZE_BN254_REPR", plan_repr_bn254(checked)) ? 1 : 0;
        const void* p = nullptr;
        BLZ_TRY(arena_points_mont(h, hbm_addr + hbm_off, nof_elements * 8, &p, true));
        if (!p) ok = false;   // (written in between)
        BLZ_WAIT(h, sync_stream_bounded(h->eng.stream, "precompute plan: even-base copy"));
    }
    if (consistent) *consistent = ok ? 1 : 0;
    return BLZ_OK;
}

int blz_msm_precompute_plan_info(blz_msm* h, uint64_t out[4]) {
    if (!h || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    for (int i = 0; i < 4; ++i) out[i] = h->pc_info[i];
    return BLZ_OK;
}

int blz_msm_set_scalar_range(blz_msm* h, uint32_t bit_lo, uint32_t bit_hi) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    if ((bit_lo == 0 && bit_hi == 0) || (bit_lo == 0 && bit_hi == 256)) {
        h->range_lo = h->range_hi = 0;
        return BLZ_OK;
    }
    if (h->pf != 1) return fail(BLZ_ERR_INVALID_PARAM, "scalar ranges need precompute_factor 1 (a precompute handle's scalars are 32-bit chunks already)");
    if (bit_lo >= bit_hi || bit_hi > 256 || (bit_lo & 31u) || (bit_hi & 31u))
        return fail(BLZ_ERR_INVALID_PARAM, "scalar range [%u, %u): want 32-bit aligned 0 <= lo < hi <= 256", bit_lo, bit_hi);
    h->range_lo = (int)bit_lo;
    h->range_hi = (int)bit_hi;
    return BLZ_OK;
}

// One candidate of the shard layout: R scalar ranges of 256 / R bits x nranks / R element chunks; rank = chunk * R + range.
// Estimates for the most expensive rank of the layout (range 0 holds the most real bits).
static int shard_candidate(int curve, uint32_t nof_elements, int nranks, int rank, uint32_t flags, int R, uint32_t out[8], double* cost_ms) {
    static const int r_bits[3] = {253, 255, 254};
    if (R < 1 || R > 8 || (R & (R - 1)) || nranks % R) return fail(BLZ_ERR_INVALID_PARAM, "%d scalar ranges do not divide %d ranks", R, nranks);
    const int PC = nranks / R;
    const uint64_t base = nof_elements / PC, rem = nof_elements % PC;
    const uint32_t per = (uint32_t)(base + (rem ? 1 : 0));   // the largest chunk
    if (per == 0 && R > 1) return fail(BLZ_ERR_INVALID_PARAM, "fewer elements than element chunks");
    const int vbits = 256 / R;
    const MsmPlan P = make_plan(per ? per : 1, vbits, vbits < r_bits[curve] ? vbits : r_bits[curve], 0);
    if (P.c == 0) return fail(BLZ_ERR_INVALID_PARAM, "no window plan for %u elements of %d bits", per, vbits);
    // the planner's cost is in ns, fitted to the round-1 kernels to RANK plans; as an absolute time it runs 13 % above what
    // the shards measure today (profiles/r03_shard_layouts.txt: element split 64.1 / 33.4 / 18.4 ms measured against 74.8 /
    // 39.3 / 20.7 estimated, scalar split 61.4 / 32.6 / 18.3 against 71.5 / 36.1 / 19.7) - and it is compared with a
    // transfer time here, so it is scaled
    const double compute_ms = P.cost * 1e-6 * 0.87;
    // measured host -> device rate of pageable buffers on this platform (56.3 GB/s)
    const double link_bytes = (flags & BLZ_SHARD_SCALARS_FROM_HOST ? (double)per * 32.0 : 0.0) +
                              (flags & BLZ_SHARD_BASES_FROM_HOST ? (double)per * (double)blz_point_size(curve) : 0.0);
    const double link_ms = link_bytes / 56.3e9 * 1e3;
    const double mem_bytes = (double)per * ((double)blz_point_size(curve) + (double)mont_point_bytes(curve) + 32.0);
    const int pc = rank / R, rg = rank % R;
    const uint64_t first = (uint64_t)pc * base + ((uint64_t)pc < rem ? pc : rem);
    out[0] = (uint32_t)first;
    out[1] = (uint32_t)(base + ((uint64_t)pc < rem ? 1 : 0));
    out[2] = (uint32_t)(rg * vbits);
    out[3] = (uint32_t)((rg + 1) * vbits);
    out[4] = (uint32_t)R;
    out[5] = (uint32_t)(compute_ms * 1e3);
    out[6] = (uint32_t)(link_ms * 1e3);
    out[7] = (uint32_t)(mem_bytes / 1048576.0);
    // A stream of tasks overlaps a task's transfer with its predecessor's compute - not for free: measured per rank of a 2^26
    // job (profiles/r04_shard_layouts.txt: every candidate with resident scalars and with scalars from host memory), a task
    // whose upload hides costs its compute + 13 - 20 % of the upload (the blocking set_data keeps the host from collecting and
    // submitting; the copy shares HBM with the accumulation), and one whose upload does not hide costs the upload + 3 - 4 ms.
    if (cost_ms) {
        const double hidden = compute_ms + 0.15 * link_ms, exposed = 1.1 * link_ms;
        *cost_ms = hidden > exposed ? hidden : exposed;
    }
    return BLZ_OK;
}

int blz_msm_shard_layout_candidate(int curve, uint32_t nof_elements, int nranks, int rank, uint32_t flags, int R, uint32_t out[8]) {
    if (!out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    if (curve < 0 || curve > 2) return fail(BLZ_ERR_INVALID_PARAM, "unknown curve %d", curve);
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(BLZ_ERR_INVALID_PARAM, "rank %d of %d", rank, nranks);
    return shard_candidate(curve, nof_elements, nranks, rank, flags, R, out, nullptr);
}

int blz_msm_shard_layout_ex(int curve, uint32_t nof_elements, int nranks, int rank, uint32_t flags, uint32_t out[8]) {
    if (!out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    if (curve < 0 || curve > 2) return fail(BLZ_ERR_INVALID_PARAM, "unknown curve %d", curve);
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(BLZ_ERR_INVALID_PARAM, "rank %d of %d", rank, nranks);
    // BLAZE_SHARD = elements | bits forces R = 1 / the largest R that divides nranks
    const char* mode = getenv("BLAZE_SHARD");
    const bool force_elements = mode && strcmp(mode, "elements") == 0, force_bits = mode && strcmp(mode, "bits") == 0;
    // device memory a rank may spend on its bases (raw + Montgomery copy) and scalars: half of the 288 GB, the rest is
    // workspace (entries, sort intermediates, partial sums) and whatever else the host keeps there
    const double mem_budget_mib = 144.0 * 1024.0;
    int bestR = 0;
    double best = 1e300, cost1 = 1e300;
    uint32_t tmp[8];
    for (int R = 1; R <= 8; R *= 2) {
        if (nranks % R) continue;
        double cost = 0;
        if (shard_candidate(curve, nof_elements, nranks, rank, flags, R, tmp, &cost) != BLZ_OK) continue;
        if ((double)tmp[7] > mem_budget_mib && R > 1) continue;
        if (R == 1) cost1 = cost;
        if (force_elements) { if (R == 1) { bestR = 1; break; } continue; }
        if (force_bits) { bestR = R; continue; }
        // the element split is the simpler layout (no shared bases, the smallest upload per rank): a scalar split has to
        // beat it by more than 2 % of the planner's estimate
        const double eff = R == 1 ? cost : cost * 1.02;
        if (eff < best) { best = eff; bestR = R; }
    }
    (void)cost1;
    if (bestR == 0) return fail(BLZ_ERR_INVALID_PARAM, "no shard layout for %u elements on %d ranks", nof_elements, nranks);
    return shard_candidate(curve, nof_elements, nranks, rank, flags, bestR, out, nullptr);
}

int blz_msm_shard_layout(int curve, uint32_t nof_elements, int nranks, int rank, uint32_t out[4]) {
    if (!out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    uint32_t o[8];
    BLZ_TRY(blz_msm_shard_layout_ex(curve, nof_elements, nranks, rank, 0u, o));
    for (int i = 0; i < 4; ++i) out[i] = o[i];
    return BLZ_OK;
}

int blz_msm_window_table_info(blz_msm* h, uint64_t out[4]) {
    if (!h || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    for (int i = 0; i < 4; ++i) out[i] = h->table_info[i];
    return BLZ_OK;
}

int blz_msm_initialize(blz_msm* h, uint32_t nof_elements, int has_hbm, uint64_t hbm_addr, uint64_t hbm_off) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    BLZ_LIVE(h);
    (void)hbm_off;  // msm_api.rs:84: only hbm_point_addr.0 is programmed
    if (h->mem_type == BLZ_DMA && !has_hbm) {
        h->bases_from_hbm = false;  // BASES_SOURCE = 0 (msm_api.rs:75-81)
    } else {
        if (!has_hbm)  // reference: params.hbm_point_addr.unwrap() panics (msm_api.rs:84)
            return fail(BLZ_ERR_INVALID_PARAM, "mem_type HBM requires hbm_point_addr");
        h->bases_from_hbm = true;  // BASES_SOURCE = 1 + start address (msm_api.rs:85-95)
        h->hbm_addr = hbm_addr;
    }
    h->nof_elements = nof_elements;
    h->initialized = true;
    return BLZ_OK;
}

int blz_msm_start_process(blz_msm* h) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    BLZ_LIVE(h);
    if (!h->initialized) return fail(BLZ_ERR_INVALID_PARAM, "start_process before initialize");
    if (h->armed) return fail(BLZ_ERR_INVALID_PARAM, "a task is already queued and waits for data");
    h->armed = true;
    h->task_label += 1;
    if (h->data_ready && h->staged_n != h->nof_elements)
        return fail(BLZ_ERR_INVALID_PARAM, "staged data has %u elements, task expects %u", h->staged_n, h->nof_elements);
    return launch_if_ready(h);
}

int blz_msm_set_data(blz_msm* h, const uint8_t* points, size_t points_len, const uint8_t* scalars, size_t scalars_len,
                     uint32_t nof_elements, int has_hbm, uint64_t hbm_addr, uint64_t hbm_off) {
    if (h && h->armed && nof_elements != h->nof_elements)
        return fail(BLZ_ERR_INVALID_PARAM, "set_data carries %u elements, queued task expects %u", nof_elements, h->nof_elements);
    return stage_common(h, points != nullptr, points, points_len, scalars, scalars_len, nof_elements, has_hbm, hbm_addr,
                        hbm_off, false);
}

int blz_msm_set_data_device(blz_msm* h, const void* d_points, size_t points_len, const void* d_scalars,
                            size_t scalars_len, uint32_t nof_elements, int has_hbm, uint64_t hbm_addr, uint64_t hbm_off) {
    if (h && h->armed && nof_elements != h->nof_elements)
        return fail(BLZ_ERR_INVALID_PARAM, "set_data carries %u elements, queued task expects %u", nof_elements, h->nof_elements);
    return stage_common(h, d_points != nullptr, d_points, points_len, d_scalars, scalars_len, nof_elements, has_hbm,
                        hbm_addr, hbm_off, true);
}

int blz_msm_wait_result(blz_msm* h) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    BLZ_LIVE(h);
    if (h->in_flight.empty()) {
        if (!h->results.empty()) return BLZ_OK;  // RESULT_VALID already set
        return fail(BLZ_ERR_INVALID_PARAM, "wait_result with no task in flight (the reference would spin forever)");
    }
    // tasks complete in submission order: wait for the oldest, move its bytes to the result queue.  The wait is
    // bounded (BLAZE_WAIT_TIMEOUT_MS; the reference polls RESULT_VALID without a deadline, msm_api.rs:222-238): on
    // expiry the task stays queued, the handle turns reset-only and the error is Unknown.
    blz_msm::Pending p = h->in_flight.front();
    blz_msm::Res r;
    r.bytes.resize(result_size(h));
    r.label = p.label;
    wait_clear();
    const int rc = h->eng.finish(p.slot, r.bytes.data());
    if (rc != BLZ_OK && wait_timed_out()) {
        h->wedged = true;
        return rc;
    }
    h->in_flight.pop_front();
    if (rc != BLZ_OK) return rc;
    h->results.push_back(std::move(r));
    return BLZ_OK;
}

int blz_msm_result(blz_msm* h, uint8_t* out, size_t out_cap, size_t* out_len, uint32_t* label) {
    if (!h || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    if (h->results.empty()) return fail(BLZ_ERR_READ, "ADDR_HIF2CPU_C_RESULT: result queue is empty");
    blz_msm::Res& r = h->results.front();
    if (out_cap < r.bytes.size()) return fail(BLZ_ERR_INVALID_PARAM, "result buffer too small: %zu < %zu", out_cap, r.bytes.size());
    memcpy(out, r.bytes.data(), r.bytes.size());
    if (out_len) *out_len = r.bytes.size();
    if (label) *label = r.label;
    h->results.pop_front();  // POP_RESULT (msm_api.rs:264-268)
    return BLZ_OK;
}

int blz_msm_load_data_to_hbm(blz_msm* h, const uint8_t* points, size_t len, uint64_t addr, uint64_t off) {
    if (!h || (!points && len)) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    BLZ_LIVE(h);
    BLZ_ARENA_ADDR(addr, off);
    BLZ_WAIT(h, arena_write(h->device, addr + off, points, len, false, h->eng.stream));
    h->bases_from_hbm = true;  // msm_api.rs:301-311 flips BASES_SOURCE and programs the address
    h->hbm_addr = addr;
    return BLZ_OK;
}

int blz_msm_load_data_to_hbm_device(blz_msm* h, const void* d_points, size_t len, uint64_t addr, uint64_t off) {
    if (!h || (!d_points && len)) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    BLZ_LIVE(h);
    BLZ_ARENA_ADDR(addr, off);
    BLZ_WAIT(h, arena_write(h->device, addr + off, d_points, len, true, h->eng.stream));
    h->bases_from_hbm = true;
    h->hbm_addr = addr;
    return BLZ_OK;
}

int blz_msm_get_data_from_hbm(blz_msm* h, uint8_t* out, size_t len, uint64_t addr, uint64_t off) {
    if (!h || (!out && len)) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    BLZ_ARENA_ADDR(addr, off);
    BLZ_TRY(use_device(h->device));
    Arena& A = arena_for(h->device);
    std::lock_guard<std::mutex> lk(A.mu);
    ArenaExtent* e = arena_find(A, addr + off, len);
    if (!e) return fail(BLZ_ERR_READ, "no loaded extent covers [%llu, +%zu)", (unsigned long long)(addr + off), len);
    return arena_read_bytes(A, *e, addr + off - e->start, len, out, h->eng.aux_stream);
}

int blz_msm_task_label(blz_msm* h, uint32_t* out) {
    if (!h || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    *out = h->task_label;
    return BLZ_OK;
}
int blz_msm_nof_elements(blz_msm* h, uint32_t* out) {
    if (!h || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    *out = h->nof_elements;
    return BLZ_OK;
}
int blz_msm_is_engine_ready(blz_msm* h, uint32_t* out) {
    if (!h || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    *out = h->eng.can_accept() ? 1u : 0u;
    return BLZ_OK;
}

int blz_msm_reset(blz_msm* h) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    BLZ_TRY(use_device(h->device));
    // waits (bounded) until every stream of the handle has drained; a handle whose device work still does not
    // complete stays wedged and reset fails with Unknown again
    BLZ_TRY(sync_stream_bounded(h->copy_stream, "reset: copy stream"));
    BLZ_TRY(h->eng.sync_all());
    h->wedged = false;
    h->armed = h->data_ready = false;
    h->in_flight.clear();
    h->results.clear();
    h->staged_n = 0;
    return BLZ_OK;
}

int blz_msm_last_timings(blz_msm* h, float out[8]) {
    if (!h || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    memcpy(out, h->eng.last_ms, sizeof(float) * 8);
    return BLZ_OK;
}

// Device memory behind a handle, in bytes: [0] the engine's workspace (sort intermediates, entries, bucket tables, partial
// sums, reduce levels: grown to the largest task seen), [1] the handle's staging buffers (host scalars / points of DMA-mode tasks,
// their Montgomery copy, the exchange buffer), and - shared by every handle of the device - the arena: [2] raw bytes as loaded
// (allocated capacity), [3] Montgomery copies of the bases, [4] window tables (built, being built, and the builds' scratch rows);
// [5] the sum.
int blz_msm_memory_info(blz_msm* h, uint64_t out[6]) {
    if (!h || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    MsmEngine& E = h->eng;
    uint64_t ws = 0;
    for (const DevBuf* b : {&E.coarse, &E.inter, &E.inter2, &E.slice_map, &E.partial, &E.blocksums, &E.result, &E.sort3_tabs, &E.bucket_sums, &E.bucket_ident})
        ws += b->cap;
    for (const auto& B : E.sbuf)
        for (const DevBuf* b : {&B.count, &B.off, &B.unit_off, &B.unit_bucket, &B.unit_order, &B.lenhist, &B.entries, &B.stats, &B.range_scalars}) ws += b->cap;
    for (const auto& S : E.slots)
        for (const DevBuf* b : {&S.lvlA[0], &S.lvlA[1], &S.lvlC[0], &S.lvlC[1]}) ws += b->cap;
    uint64_t staging = h->points_mont.cap + h->comm_buf.cap;
    for (int i = 0; i < 2; ++i) staging += h->scalars_buf[i].cap + h->points_raw[i].cap;
    uint64_t raw = 0, mont = 0, tables = 0;
    {
        Arena& A = arena_for(h->device);
        std::lock_guard<std::mutex> lk(A.mu);
        for (const auto& e : A.ext) {
            if (!e.imported && e.raw) raw += e.cap;
            if (e.mont) mont += e.mont_bytes;
            for (const auto& t : e.tables) tables += t.bytes;
            if (e.build.tab) tables += e.build.bytes;
        }
        tables += A.build_scratch_bytes;
    }
    out[0] = ws;
    out[1] = staging;
    out[2] = raw;
    out[3] = mont;
    out[4] = tables;
    out[5] = ws + staging + raw + mont + tables;
    return BLZ_OK;
}

int blz_msm_last_sort_hidden(blz_msm* h, int* out) {
    if (!h || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    *out = h->eng.last_sort_hidden ? 1 : 0;
    return BLZ_OK;
}

int blz_msm_plan(int curve, uint32_t nof_elements, int is_precompute, uint32_t out[4], uint8_t* widths) {
    if (!out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    if (curve < 0 || curve > 2) return fail(BLZ_ERR_INVALID_PARAM, "unknown curve %d", curve);
    static const int r_bits[3] = {253, 255, 254};
    const uint64_t npts = (uint64_t)nof_elements * (is_precompute ? BLZ_PRECOMPUTE_FACTOR : BLZ_PRECOMPUTE_FACTOR_BASE);
    if (npts == 0 || npts >= (1ull << 31)) return fail(BLZ_ERR_INVALID_PARAM, "nof_elements out of range");
    const int sbits = is_precompute ? 32 : 256;
    MsmPlan P = make_plan((uint32_t)npts, sbits, is_precompute ? 32 : r_bits[curve], 0);
    if (P.c == 0) return fail(BLZ_ERR_INVALID_PARAM, "no window plan");
    out[0] = (uint32_t)P.c;
    out[1] = (uint32_t)P.W;
    out[2] = P.L;
    out[3] = (uint32_t)P.G;
    if (widths)
        for (int w = 0; w < P.W; ++w) widths[w] = P.width[w];
    return BLZ_OK;
}

int blz_msm_combine_partials(blz_msm* h, const uint8_t* partials, size_t count, uint8_t* out, size_t out_cap) {
    if (!h || !out || (!partials && count)) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    if (out_cap < result_size(h)) return fail(BLZ_ERR_INVALID_PARAM, "result buffer too small");
    BLZ_LIVE(h);
    BLZ_WAIT(h, h->eng.combine_partials(partials, count, out, false));
    return BLZ_OK;
}

// ---- multi-GPU exchange: RCCL all-gather of the per-rank partial results + rank-ordered add
#define BLZ_NCCL(api, call)                                                                                  \
    do {                                                                                                     \
        ncclResult_t r__ = (call);                                                                           \
        if (r__ != ncclSuccess) return fail(BLZ_ERR_UNKNOWN, "%s failed: %s", #call, (api)->GetErrorString(r__)); \
    } while (0)

int blz_comm_unique_id(uint8_t out[BLZ_COMM_ID_BYTES]) {
    if (!out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    const RcclApi* api = rccl_api();
    if (!api) return BLZ_ERR_LOAD_FAILED;
    static_assert(BLZ_COMM_ID_BYTES == sizeof(ncclUniqueId), "id size");
    ncclUniqueId id;
    BLZ_NCCL(api, api->GetUniqueId(&id));
    memcpy(out, &id, sizeof(id));
    return BLZ_OK;
}

// Communicator bring-up is a rendezvous: ncclCommInitRank returns when EVERY rank has called it, and for ever never
// if one of them died on the way.  It therefore runs on a helper thread and the caller waits for it against
// BLAZE_COMM_TIMEOUT_MS (default 60 000); on expiry the call fails with Unknown and the helper - parked inside RCCL - is
// abandoned (it owns its state through the shared_ptr and never touches the handle).
struct CommJob {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    int rc = BLZ_OK;
    std::string err;
    std::vector<ncclComm_t> comms;
};
static int comm_timeout_ms() {
    const char* s = getenv("BLAZE_COMM_TIMEOUT_MS");
    int v = s && *s ? atoi(s) : 60000;
    return v > 0 ? v : 60000;
}
static int run_comm_job(std::shared_ptr<CommJob> job, std::function<int(CommJob&)> fn, const char* what) {
    std::thread([job, fn] {
        int rc = fn(*job);
        std::lock_guard<std::mutex> lk(job->mu);
        job->rc = rc;
        if (rc != BLZ_OK) job->err = blz_last_error_message();   // the message lives in the helper's thread-local buffer
        job->done = true;
        job->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(job->mu);
    const int limit = comm_timeout_ms();
    if (!job->cv.wait_for(lk, std::chrono::milliseconds(limit), [&] { return job->done; }))
        return fail(BLZ_ERR_UNKNOWN, "%s did not complete within %d ms (BLAZE_COMM_TIMEOUT_MS): a peer rank never arrived, or "
                    "RCCL cannot reach it; the bring-up thread is abandoned", what, limit);
    if (job->rc != BLZ_OK) return fail(job->rc, "%s", job->err.c_str());
    return BLZ_OK;
}

int blz_msm_comm_init(blz_msm* h, int rank, int nranks, const uint8_t id_bytes[BLZ_COMM_ID_BYTES]) {
    if (!h || !id_bytes) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(BLZ_ERR_INVALID_PARAM, "rank %d of %d", rank, nranks);
    if (h->comm) return fail(BLZ_ERR_INVALID_PARAM, "communicator already initialised on this handle");
    const RcclApi* api = rccl_api();
    if (!api) return BLZ_ERR_LOAD_FAILED;
    BLZ_TRY(use_device(h->device));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    auto job = std::make_shared<CommJob>();
    job->comms.assign(1, nullptr);
    const int dev = h->device;
    char what[96];
    snprintf(what, sizeof(what), "ncclCommInitRank (rank %d of %d)", rank, nranks);
    BLZ_TRY(run_comm_job(job, [api, dev, id, rank, nranks](CommJob& j) -> int {
        BLZ_HIP(hipSetDevice(dev), BLZ_ERR_FILE);
        BLZ_NCCL(api, api->CommInitRank(&j.comms[0], nranks, id, rank));   // collective: every rank calls it
        return BLZ_OK;
    }, what));
    h->comm = job->comms[0];
    h->comm_rank = rank;
    h->comm_size = nranks;
    return h->comm_buf.reserve((size_t)(nranks + 1) * result_size(h) + 64);
}

// One process driving several devices (the "management layer" of README.md:20-22 as a single host thread): one
// handle per device, rank i = handles[i].  The n bring-ups are one RCCL group (ncclGroupStart / End), because n
// sequential ncclCommInitRank calls from one thread would each wait for the ones that thread has not made yet.
int blz_msm_comm_init_all(blz_msm* const* handles, int n) {
    if (!handles || n < 1) return fail(BLZ_ERR_INVALID_PARAM, "no handles");
    for (int i = 0; i < n; ++i) {
        if (!handles[i]) return fail(BLZ_ERR_INVALID_PARAM, "null handle %d", i);
        if (handles[i]->comm) return fail(BLZ_ERR_INVALID_PARAM, "communicator already initialised on handle %d", i);
        if (handles[i]->curve != handles[0]->curve) return fail(BLZ_ERR_INVALID_PARAM, "handles of different curves");
        for (int k = 0; k < i; ++k)
            if (handles[k]->device == handles[i]->device)
                return fail(BLZ_ERR_INVALID_PARAM, "handles %d and %d share device %d (RCCL: one rank per device)", k, i, handles[i]->device);
    }
    const RcclApi* api = rccl_api();
    if (!api) return BLZ_ERR_LOAD_FAILED;
    auto job = std::make_shared<CommJob>();
    job->comms.assign((size_t)n, nullptr);
    std::vector<int> devs;
    for (int i = 0; i < n; ++i) devs.push_back(handles[i]->device);
    BLZ_TRY(run_comm_job(job, [api, devs, n](CommJob& j) -> int {
        ncclUniqueId id;
        BLZ_NCCL(api, api->GetUniqueId(&id));
        BLZ_NCCL(api, api->GroupStart());
        for (int i = 0; i < n; ++i) {
            if (hipSetDevice(devs[i]) != hipSuccess) { (void)api->GroupEnd(); return fail(BLZ_ERR_FILE, "hipSetDevice(%d) failed", devs[i]); }
            ncclResult_t r = api->CommInitRank(&j.comms[i], n, id, i);
            if (r != ncclSuccess) { (void)api->GroupEnd(); return fail(BLZ_ERR_UNKNOWN, "ncclCommInitRank(rank %d) failed: %s", i, api->GetErrorString(r)); }
        }
        BLZ_NCCL(api, api->GroupEnd());
        return BLZ_OK;
    }, "ncclCommInitRank group (single process)"));
    for (int i = 0; i < n; ++i) {
        handles[i]->comm = job->comms[i];
        handles[i]->comm_rank = i;
        handles[i]->comm_size = n;
        BLZ_TRY(use_device(handles[i]->device));
        BLZ_TRY(handles[i]->comm_buf.reserve((size_t)(n + 1) * result_size(handles[i]) + 64));
    }
    return BLZ_OK;
}

// enqueue this handle's half of the exchange on its exchange stream (no host wait)
static int enqueue_all_gather(blz_msm* h, const RcclApi* api, const uint8_t* partial, uint8_t** recv_out) {
    BLZ_TRY(use_device(h->device));
    // own stream: the exchange must not queue behind the next task's accumulation on the main stream
    hipStream_t st = h->eng.aux_stream;
    const size_t rs = result_size(h);
    uint8_t* send = h->comm_buf.as<uint8_t>();
    uint8_t* recv = send + ((rs + 63) / 64) * 64;
    BLZ_HIP(hipMemcpyAsync(send, partial, rs, hipMemcpyHostToDevice, st), BLZ_ERR_WRITE);
    BLZ_NCCL(api, api->AllGather(send, recv, rs, ncclUint8, h->comm, st));
    *recv_out = recv;
    return BLZ_OK;
}

int blz_msm_all_gather_combine(blz_msm* h, const uint8_t* partial, uint8_t* out, size_t out_cap) {
    if (!h || !partial || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    BLZ_LIVE(h);
    if (!h->comm) return fail(BLZ_ERR_INVALID_PARAM, "all_gather_combine before comm_init");
    if (out_cap < result_size(h)) return fail(BLZ_ERR_INVALID_PARAM, "result buffer too small");
    const RcclApi* api = rccl_api();
    if (!api) return BLZ_ERR_LOAD_FAILED;
    uint8_t* recv = nullptr;
    BLZ_TRY(enqueue_all_gather(h, api, partial, &recv));
    // rank order = buffer order; the wait inside is bounded (a peer that never joins the all-gather: Unknown, wedged)
    BLZ_WAIT(h, h->eng.combine_partials(recv, (size_t)h->comm_size, out, true));
    return BLZ_OK;
}

// The exchange for the handles of blz_msm_comm_init_all, from the one thread that drives them: partials and out hold
// n x result_size bytes in handle order; every handle's sum is written (identical bytes).  All n all-gathers are
// enqueued as one RCCL group before any of them is waited for.
int blz_msm_all_gather_combine_all(blz_msm* const* handles, int n, const uint8_t* partials, uint8_t* out, size_t out_cap) {
    if (!handles || n < 1 || !partials || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    const RcclApi* api = rccl_api();
    if (!api) return BLZ_ERR_LOAD_FAILED;
    for (int i = 0; i < n; ++i) {
        if (!handles[i] || !handles[i]->comm || handles[i]->comm_size != n || handles[i]->comm_rank != i)
            return fail(BLZ_ERR_INVALID_PARAM, "handle %d is not rank %d of a %d-rank communicator (blz_msm_comm_init_all)", i, i, n);
        BLZ_LIVE(handles[i]);
    }
    const size_t rs = result_size(handles[0]);
    if (out_cap < rs * (size_t)n) return fail(BLZ_ERR_INVALID_PARAM, "result buffer too small: %zu < %zu", out_cap, rs * (size_t)n);
    std::vector<uint8_t*> recv((size_t)n, nullptr);
    BLZ_NCCL(api, api->GroupStart());
    for (int i = 0; i < n; ++i) {
        int rc = enqueue_all_gather(handles[i], api, partials + (size_t)i * rs, &recv[i]);
        if (rc != BLZ_OK) { (void)api->GroupEnd(); return rc; }
    }
    BLZ_NCCL(api, api->GroupEnd());
    for (int i = 0; i < n; ++i) BLZ_WAIT(handles[i], handles[i]->eng.combine_partials(recv[i], (size_t)n, out + (size_t)i * rs, true));
    return BLZ_OK;
}

int blz_msm_stream(blz_msm* h, void** hip_stream, int* device_id) {
    if (!h || !hip_stream) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    if (device_id) *device_id = h->device;
    *hip_stream = (void*)h->eng.stream;
    return BLZ_OK;
}

int blz_msm_comm_free(blz_msm* h) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    if (!h->comm) return BLZ_OK;
    const RcclApi* api = rccl_api();
    if (api) {
        (void)hipSetDevice(h->device);
        // a communicator whose exchange never completed cannot be destroyed gracefully (ncclCommDestroy waits for it)
        if (sync_stream_bounded(h->eng.aux_stream, "comm_free: exchange stream") == BLZ_OK) (void)api->CommDestroy(h->comm);
        else if (api->CommAbort) (void)api->CommAbort(h->comm);
    }
    h->comm = nullptr;
    h->comm_size = 0;
    h->comm_buf.release();
    return BLZ_OK;
}

}  // extern "C"
