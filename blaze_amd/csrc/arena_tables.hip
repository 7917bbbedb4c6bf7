// Arena extents as the MSM tasks see them: the Montgomery copy of an extent's points, the checked-table plan of precompute
// handles and the resident-base window tables - built, kept and invalidated per extent (common.hpp ArenaExtent).
#include <chrono>
#include <thread>

#include "msm_handle.hpp"

namespace blz {

bool wants_table_mode(const blz_msm* h) { return h->pf == 1 && h->window_table != 0; }

// Resolve the Montgomery-form view of `npts` points stored at arena offset `pos`: (re)builds the part of the
// extent's shadow that is stale, on this handle's main stream, and orders this stream behind conversions other
// handles may have enqueued.
// even (checked-table plan of a precompute handle): the copy holds the even bases of every element only - B_0, B_2, B_4, B_6,
// contiguous, 4 per element - and *out addresses the copy of the element at `pos`; npts counts the RAW points (8 per element).
// Granted only while the extent's table check still stands for these points (looked up under the same lock that resolves the
// copy: a write by another thread between the check and this call leaves *out null, and the caller takes the exact path).
int arena_points_mont(blz_msm* h, uint64_t pos, uint32_t npts, const void** out, bool even) {
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
// for it - with the arena unlocked; the answer is committed only if no write reached the extent (epoch) and no other check of it
// committed (PrecompCheck::gen) in the meantime.
int arena_precompute_check(blz_msm* h, uint64_t pos, uint32_t nelem, bool* ok, uint64_t* checked_elems) {
    *ok = false;
    if (checked_elems) *checked_elems = nelem;
    const size_t ps = point_size(h);
    const size_t len = (size_t)nelem * 8 * ps;
    Arena& A = arena_for(h->device);
    uint64_t epoch = 0, first = 0, gen = 0;
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
        BLZ_TRY(arena_restore_raw(A, *e, h->eng.stream));   // (the check reads the raw bytes)
        if (!(flag = arena_flag_acquire(A))) {
            BLZ_LOG(1, "precompute plan: no flag word for the check: exact path");
            return BLZ_OK;
        }
        epoch = e->epoch;
        gen = C.gen;   // (the verdict below is about the record as it stands NOW: state, range and redo span)
        hipStream_t st = h->eng.stream;
        if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess) {
            if (t0) (void)hipEventDestroy(t0);
            arena_flag_release(A, flag);
            return fail_hip(BLZ_ERR_UNKNOWN, "event creation failed");
        }
        int rc = BLZ_OK;
        if (hipMemsetAsync(flag, 0, 4, st) != hipSuccess || hipEventRecord(t0, st) != hipSuccess) rc = fail_hip(BLZ_ERR_UNKNOWN, "precompute check: enqueue failed");
        if (rc == BLZ_OK) rc = h->eng.check_precompute((const char*)e->raw + (chk_pos - e->start), chk_elems, flag, st);
        if (rc == BLZ_OK && hipEventRecord(t1, st) != hipSuccess) rc = fail_hip(BLZ_ERR_UNKNOWN, "precompute check: enqueue failed");
        if (rc != BLZ_OK) {
            (void)hipEventDestroy(t0);
            (void)hipEventDestroy(t1);
            // (a check kernel that did get enqueued may still raise the word: it stays owned - 4 bytes - rather than be handed to someone else)
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
        if (hipMemcpy(&flag_h, flag, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail_hip(BLZ_ERR_READ, "precompute check: flag read failed");
    }
    const bool kernel_done = rc == BLZ_OK || !wait_timed_out();
    if (kernel_done) {
        (void)hipEventDestroy(t0);
        (void)hipEventDestroy(t1);
    }
    std::lock_guard<std::mutex> lk(A.mu);
    if (kernel_done) arena_flag_release(A, flag);   // (a wedged check keeps its word: the kernel may still write it)
    BLZ_TRY(rc);
    ArenaExtent* e = arena_find(A, pos, len);
    if (!e || e->epoch != epoch) {
        BLZ_LOG(1, "precompute plan: the extent was written while its table was being checked: exact path for this task");
        return BLZ_OK;
    }
    ArenaExtent::PrecompCheck& C = e->pcheck;
    if (C.gen != gen) {
        // another handle's check of this extent committed while this one ran unlocked: its record (state, range) is not the one
        // this verdict was formed against - a partial re-check must not turn a range the other check refuted into a consistent
        // one, a full check must not shrink or move the other's range.  This task takes the exact path; the next one looks again.
        BLZ_LOG(1, "precompute plan: another check of the extent committed while this one ran: exact path for this task");
        return BLZ_OK;
    }
    ++C.gen;
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
            arena_flag_release(A, B.flag);
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
        arena_flag_release(A, B.flag);
        B = ArenaExtent::TableBuild();
        BLZ_LOG(1, "window table: %llu bases x %d windows of %d bits, %.1f MiB, complete %.1f ms after its first chunk", (unsigned long long)t.npts,
                t.W, t.c, t.bytes / 1048576.0, ms);
    }
    // rows of bases that were rewritten since the tables were built (arena_write: small rewrites keep the tables): re-tabulated
    // here, on this handle's main stream, behind a drain (another handle's task may be gathering from the very rows).  A table
    // of another format than this handle's (its curve's other arithmetic), or one whose scratch rows are gone, is dropped instead.
    if (e->tab_dirty_lo < e->tab_dirty_hi && !e->tables.empty()) {
        BLZ_TRY(sync_device_bounded("window table: drain before the rewritten bases are re-tabulated"));
        uint32_t* pflag = arena_flag_acquire(A);   // owned for this call (every exit below is behind a drained stream, or leaks the word)
        if (pflag && hipMemsetAsync(pflag, 0, 4, h->eng.stream) != hipSuccess) {
            (void)hipGetLastError();
            arena_flag_release(A, pflag);
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
        if (!patched) arena_flag_release(A, pflag);
        if (patched) {
            uint32_t flag_h = 0;
            BLZ_WAIT(h, sync_stream_bounded(h->eng.stream, "window table: rewritten bases re-tabulated"));
            const hipError_t he_flag = hipMemcpy(&flag_h, pflag, 4, hipMemcpyDeviceToHost);
            arena_flag_release(A, pflag);
            BLZ_HIP(he_flag, BLZ_ERR_READ);
            if (flag_h) {
                BLZ_LOG(1, "window table: a rewritten base has a multiple at infinity (a point of even order): plain path for this extent");
                arena_drop_table(A, *e);   // (the stream was drained just now, the device before)
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
        if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&done) != hipSuccess) rc = fail_hip(BLZ_ERR_UNKNOWN, "event creation failed");
        // the build's "a multiple came out as infinity" flag: a word of its own until the build is adopted or dropped (builds share
        // the scratch rows in stream order, but a flag is read by the host when its build is ADOPTED, many tasks later)
        uint32_t* flag = arena_flag_acquire(A);
        if (!flag) {
            (void)hipFree(tab);
            if (t0) (void)hipEventDestroy(t0);
            if (done) (void)hipEventDestroy(done);
            e->table_refused = true;
            return BLZ_OK;
        }
        if (rc == BLZ_OK && hipMemsetAsync(flag, 0, 4, h->eng.stream) != hipSuccess) rc = fail_hip(BLZ_ERR_UNKNOWN, "memset failed");
        if (rc != BLZ_OK) {
            if (t0) (void)hipEventDestroy(t0);
            if (done) (void)hipEventDestroy(done);
            (void)hipFree(tab);
            arena_flag_release(A, flag);
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
// the caches (profiles/r05_tlb_probe.txt).  A precompute handle's tasks off the plan - the exact path (2^29 bases, 32 GiB at
// config 3), DMA-mode tasks, tasks that bring their own table - run on 32-bit limbs; the plan's even-base copy is a quarter of that
// per element, so it takes the reduced radix up to 2^25 elements (8 GiB of even bases) and 32-bit limbs above - measured, same box,
// ms per MSM in a stream of tasks, reduced radix / 32-bit limbs: 2^20 1.85 / 2.20, 2^22 6.60 / 7.49, 2^24 18.2 / 19.2, 2^26 74.8 / 70.0
// (exact path: 2.2, 7.2, 24.6, 92.5).  Decided by the size of the CHECKED table, not of the task (tasks over sub-ranges of one table
// would otherwise flip the arithmetic - and with it the format of the extent's copy - from task to task).  BLAZE_MSM_PLAN
// pc_repr=0|1 forces one (tests).
//
// The arithmetic is chosen PER TASK (task_repr_bn254pc below, from the path the task takes), and changed only while nothing of the
// handle is in flight: the engine's shared workspace and the extent's Montgomery copy are in ONE format at a time, and a change
// costs a device drain and a reconversion of the copy.  A task launched beside one in flight therefore keeps the arithmetic of
// the one in flight - every path is correct on either arithmetic (same bytes: tests/test_gpu_msm_precompute.py), only the speed
// differs - and the choice is made again by the first task that finds the handle idle.
int plan_repr_bn254(uint64_t nelem) {
    const int forced = plan_override("pc_repr", -1);
    if (forced == 0 || forced == 1) return forced;
    return nelem > (1ull << 25) ? 1 : 0;
}

void task_repr_bn254pc(blz_msm* h, bool on_plan, uint64_t checked_elems) {
    if (h->curve != BLZ_BN254 || h->pf != BLZ_PRECOMPUTE_FACTOR || !h->in_flight.empty()) return;
    h->eng.repr = exp_knob("BLAZE_BN254_REPR", on_plan ? plan_repr_bn254(checked_elems) : 1) ? 1 : 0;
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
        const bool plan = ok && h->eng.plan_for(n * 4, 64).c != 0;
        task_repr_bn254pc(h, plan, checked);
        if (plan) {
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
            task_repr_bn254pc(h, false, 0);
        }
    } else {
        task_repr_bn254pc(h, false, 0);   // (a precompute handle off the plan: mode iii, the plan switched off)
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

}  // namespace blz
