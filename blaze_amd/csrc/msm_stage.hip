// set_data of the MSM primitive (src/ingo_msm/msm_api.rs:155-220): where a task's bytes go - staged whole, handed to the device
// piece by piece while they cross the link, or taken from the arena - and the launch of a task whose data is complete.
#include "msm_handle.hpp"

namespace blz {

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

// How many pieces a task whose data arrives over the link is enqueued in (MsmEngine::begin): pieces of >= 2^19 points with
// their scalars (64 MiB of host bytes: 1.2 ms of link), at most 16.
// Measured (profiles/r04_dma_pieces.txt): 2^22 elements 22.6 ms in one piece, 16.2 / 15.35 / 17.1 in 4 / 8 / 16; 2^26
// 270.8, 191.8 / 178.3 / 171.5
static int pick_pieces(const blz_msm* h, uint32_t npts, bool with_points) {
    int pieces = env_int("BLAZE_MSM_PIECES", 0);   // (the same switch forces the piece count of device-resident tasks, msm.hip run())
    if (pieces <= 0) {
        if (with_points) {
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
    return pieces < 1 ? 1 : pieces;
}

int ring_reserve(blz_msm* h, uint32_t piece_pts) {
    blz_msm::PieceRing& R = h->ring;
    if (piece_pts <= R.slot_pts) return BLZ_OK;
    // (a growing ring is reallocated behind a bounded drain of the device - DevBuf::reserve - so no piece in flight reads the old one)
    const size_t ps = point_size(h), mp = mont_point_bytes(h->curve);
    BLZ_TRY(R.raw.reserve((size_t)piece_pts * blz_msm::PieceRing::SLOTS * ps + 16, true));
    BLZ_TRY(R.mont.reserve((size_t)piece_pts * blz_msm::PieceRing::SLOTS * mp + 16, true));
    R.slot_pts = piece_pts;
    for (bool& r : R.recorded) r = false;
    return BLZ_OK;
}

// piece `k` of the ring (a running number): where its raw points and their Montgomery copy go; the copy stream is ordered behind
// the to-Montgomery pass that last read the slot
static int ring_slot(blz_msm* h, uint64_t k, char** raw, char** mont) {
    blz_msm::PieceRing& R = h->ring;
    const int slot = (int)(k % blz_msm::PieceRing::SLOTS);
    *raw = (char*)R.raw.p + (size_t)slot * R.slot_pts * point_size(h);
    *mont = (char*)R.mont.p + (size_t)slot * R.slot_pts * mont_point_bytes(h->curve);
    return slot;
}
static int ring_wait_free(blz_msm* h, uint64_t k) {
    blz_msm::PieceRing& R = h->ring;
    const int slot = (int)(k % blz_msm::PieceRing::SLOTS);
    if (R.recorded[slot]) BLZ_HIP(hipStreamWaitEvent(h->copy_stream, R.raw_read[slot], 0), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}
// the piece's raw points -> Montgomery copy on the main stream; the slot's raw bytes are free behind it
static int ring_to_mont(blz_msm* h, uint64_t k, uint32_t np) {
    blz_msm::PieceRing& R = h->ring;
    char *raw = nullptr, *mont = nullptr;
    const int slot = ring_slot(h, k, &raw, &mont);
    BLZ_TRY(h->eng.points_to_mont(raw, mont, np));
    BLZ_HIP(hipEventRecord(R.raw_read[slot], h->eng.stream), BLZ_ERR_UNKNOWN);
    R.recorded[slot] = true;
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
        task_repr_bn254pc(h, false, 0);   // (DMA-mode task of a precompute handle: its arithmetic off the plan)
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
        if (!dma_pieces) {
            // (stale spans are converted on the main stream; a precompute handle on the checked-table plan: 4n even bases, 64-bit chunks)
            int tc = 0;
            BLZ_TRY(resolve_arena_task(h, h->staged_arena_pos, n, false, !h->staged_loaded_now, &npts, &sbits, &tc));
            arena_mont = h->d_points_mont;
        }
        const size_t sb = (size_t)sbits / 8;
        int pieces = pick_pieces(h, npts, dma_pieces);
        int slot = -1;
        h->eng.inputs_event = h->set_free[set];
        BLZ_TRY(h->eng.begin(npts, sbits, &slot, 0, h->range_lo, h->range_hi, pieces, true));
        const uint32_t per = h->eng.slots[slot].pts_per_slice;
        pieces = h->eng.slots[slot].slices;
        int rc = dma_pieces ? ring_reserve(h, per) : BLZ_OK;
        auto copy_in = [&](void* dst, const void* src, size_t len, const char* what) -> int {
            if (hipMemcpyAsync(dst, src, len, hipMemcpyHostToDevice, cst) != hipSuccess) return fail_hip(BLZ_ERR_WRITE, "%s failed", what);
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
                const uint64_t rk = h->ring.next;
                char *d_raw = nullptr, *d_mont = nullptr;
                if (rc == BLZ_OK) {
                    (void)ring_slot(h, rk, &d_raw, &d_mont);
                    rc = ring_wait_free(h, rk);
                }
                if (rc == BLZ_OK) rc = copy_in(d_raw, (const char*)points + (size_t)p0 * ps, (size_t)np * ps, "set_data: host -> device copy of the points");
                if (rc == BLZ_OK) rc = ring_to_mont(h, rk, np);
                if (rc == BLZ_OK) rc = h->eng.accumulate_slice(slot, k, d_mont);
                if (rc == BLZ_OK) h->ring.next = rk + 1;
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
        h->d_points_mont = dma_pieces ? nullptr : arena_mont;   // (a DMA task's points lived in the ring, piece by piece)
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

// ---- a task streamed over several set_data calls -----------------------------------------------------------------------------
// The reference's set_data walks its input in 2048-element chunks into FIFOs (msm_api.rs:155-202) and the card counts elements
// against the NUMBER_OF_MSM_ELEMENTS register that initialize() wrote (msm_hw_code.rs:18-19): whether a task's bytes come in one
// call or in many is invisible to it.  Here: with a task armed, a set_data that carries FEWER elements than the task still lacks
// is the next slice of it - in any of the three modes (scalars only over bases in the arena; points + scalars; points into the
// arena + scalars), any slice sizes (the reference's 2048-element cadence, ragged tails, one element).  The slices land back to
// back in the handle's staging set; the task is enqueued piece by piece as enough of them have arrived (the same pieces - and the
// same engine steps - as a one-call DMA-mode task: a piece's sort and accumulation run while the next slices cross the link)
// or, for tasks too small to cut, launched whole when the last slice is in.  The task is complete when received == armed_n;
// more than that is refused, and so are start_process / a mode change while a task is half-fed.  The reference's largest DMA-mode
// shape (tests/integration_msm.rs:386-467: 2^26 elements x 8 bases, a 48 GiB host vector) thus runs from host slices of any size.
void stream_abandon(blz_msm* h) {
    blz_msm::Stream& S = h->strm;
    if (S.open && S.slot >= 0) h->eng.abandon(S.slot);
    if (S.open && S.set >= 0) h->set_used[S.set] = true;   // (copies may have landed in it: the next user waits for set_free, recorded below or long past)
    S = blz_msm::Stream();
}

// hand the pieces that are complete to the engine (all of them once the task's last slice is in)
static int stream_pump(blz_msm* h) {
    blz_msm::Stream& S = h->strm;
    if (S.slot < 0) return BLZ_OK;
    const size_t mp = mont_point_bytes(h->curve), sb = (size_t)S.sbits / 8;
    const uint32_t avail = S.received * S.ppe;
    while (S.done_pts < avail && (avail - S.done_pts >= S.per || S.received == S.total)) {
        const uint32_t np = avail - S.done_pts < S.per ? avail - S.done_pts : S.per;
        const int k = S.next_piece;
        const char* d_sc = (const char*)h->scalars_buf[S.set].p + (size_t)S.done_pts * sb;
        BLZ_TRY(h->eng.sort_slice(S.slot, k, d_sc, np));
        if (S.mode == 2) {
            char *d_raw = nullptr, *d_mont = nullptr;
            (void)ring_slot(h, S.ring_first + (uint64_t)k, &d_raw, &d_mont);
            BLZ_TRY(ring_to_mont(h, S.ring_first + (uint64_t)k, np));
            BLZ_TRY(h->eng.accumulate_slice(S.slot, k, d_mont));
        } else {
            // the extent's copy as it stands NOW: a load between two slices may have moved the extent or rewritten bases (their
            // points are converted here, ahead of the piece); a checked table that a write re-opened cannot be served mid-task
            const void* mont = nullptr;
            BLZ_TRY(arena_points_mont(h, S.arena_pos, S.total * h->pf, &mont, S.even));
            if (!mont)
                return fail(BLZ_ERR_INVALID_PARAM, "the precompute table was rewritten while a task over it was being streamed on the checked-table "
                                                   "plan: reset the handle and send the task again");
            BLZ_TRY(h->eng.accumulate_slice(S.slot, k, (const char*)mont + (size_t)S.done_pts * mp));
        }
        S.done_pts += np;
        S.next_piece = k + 1;
    }
    return BLZ_OK;
}

int stage_stream(blz_msm* h, bool have_points, const void* points, size_t points_len, const void* scalars, size_t scalars_len, uint32_t m,
                 int has_hbm, uint64_t hbm_addr, uint64_t hbm_off, bool on_device) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    BLZ_LIVE(h);
    BLZ_TRY(use_device(h->device));
    blz_msm::Stream& S = h->strm;
    if (!h->armed) return fail(BLZ_ERR_INVALID_PARAM, "set_data with a part of a task needs the task queued first (start_process)");
    if (!have_points && !has_hbm) return BLZ_OK;  // reference: falls through every branch (msm_api.rs:163-216)
    if (has_hbm) BLZ_ARENA_ADDR(hbm_addr, hbm_off);
    const uint32_t total = h->nof_elements;
    const uint32_t got = S.open ? S.received : 0;
    if ((uint64_t)got + m > total)
        return fail(BLZ_ERR_INVALID_PARAM, "set_data carries %u elements, the queued task lacks only %u of its %u", m, total - got, total);
    if (!scalars && m) return fail(BLZ_ERR_INVALID_PARAM, "null scalars");
    if (scalars_len != (size_t)m * BLZ_SCALAR_SIZE) return fail(BLZ_ERR_INVALID_PARAM, "scalars length %zu != nof_elements %u * 32", scalars_len, m);
    const size_t ps = point_size(h), mp = mont_point_bytes(h->curve);
    if (have_points && points_len != (size_t)m * h->pf * ps)
        return fail(BLZ_ERR_INVALID_PARAM, "points length %zu != nof_elements %u * precompute_factor %u * %zu", points_len, m, h->pf, ps);
    const int mode = have_points ? (has_hbm ? 3 : 2) : 1;
    hipStream_t cst = h->copy_stream;
    if (!S.open) {
        // ---- the first slice: what stage_common checks for a whole task, for the task as armed
        if ((uint64_t)total * h->pf >= (1ull << 31)) return fail(BLZ_ERR_INVALID_PARAM, "too many points");
        if (h->eng.plan_for(total * h->pf, h->pf == 1 ? 256 : 32).c == 0)
            return fail(BLZ_ERR_INVALID_PARAM, "no window plan for %llu points of %d-bit scalars", (unsigned long long)total * h->pf, h->pf == 1 ? 256 : 32);
        if (!h->eng.can_accept()) return fail(BLZ_ERR_INVALID_PARAM, "task queue full (%d in flight); call wait_result first", MSM_QUEUE_DEPTH);
        blz_msm::Stream N;
        N.mode = mode;
        N.src_device = on_device;
        N.total = total;
        N.arena_pos = has_hbm ? hbm_addr + hbm_off : 0;
        N.npts = total * h->pf;
        N.sbits = h->pf == 1 ? 256 : 32;
        N.ppe = h->pf;
        if (mode == 1) {
            Arena& A = arena_for(h->device);
            std::lock_guard<std::mutex> lk(A.mu);
            if (!arena_find(A, N.arena_pos, (size_t)N.npts * ps))
                return fail(BLZ_ERR_INVALID_PARAM, "HBM bases: no loaded extent covers [%llu, +%zu) on device %d", (unsigned long long)N.arena_pos,
                            (size_t)N.npts * ps, h->device);
        } else if (mode == 2) {
            task_repr_bn254pc(h, false, 0);
        }
        // the staging set (stage_common): last used two tasks ago
        N.set = h->stage_idx;
        if (h->set_used[N.set]) BLZ_HIP(hipStreamWaitEvent(cst, h->set_free[N.set], 0), BLZ_ERR_UNKNOWN);
        BLZ_TRY(h->scalars_buf[N.set].reserve((size_t)total * BLZ_SCALAR_SIZE));
        // in pieces?  The rules of a one-call task (stage_common): DMA mode always, scalars over resident bases when the handle is
        // idle and the task large; a task that brings its table (mode 3) is launched whole behind its last slice
        const bool overlap = exp_knob("BLAZE_DMA_OVERLAP", 1) != 0;
        bool in_pieces = false;
        if (mode == 2) in_pieces = overlap;
        else if (mode == 1) in_pieces = overlap && (N.npts >= (1u << 22) || env_int("BLAZE_MSM_PIECES", 0) > 1) && h->in_flight.empty() && !wants_table(h);
        memset(h->table_info, 0, sizeof(h->table_info));
        memset(h->pc_info, 0, sizeof(h->pc_info));
        if (in_pieces) {
            if (mode == 1) {
                int tc = 0;
                uint32_t npts = 0;
                BLZ_TRY(resolve_arena_task(h, N.arena_pos, total, false, true, &npts, &N.sbits, &tc));
                N.npts = npts;
                N.ppe = npts / total;
                N.even = h->pc_info[0] != 0;
            }
            const int pieces = pick_pieces(h, N.npts, mode == 2);
            if (pieces > 1) {
                h->eng.inputs_event = h->set_free[N.set];
                BLZ_TRY(h->eng.begin(N.npts, N.sbits, &N.slot, 0, h->range_lo, h->range_hi, pieces, true));
                N.per = h->eng.slots[N.slot].pts_per_slice;
                N.pieces = h->eng.slots[N.slot].slices;
                if (N.pieces <= 1) {   // (the engine made one piece of it: the whole-task launch serves that)
                    h->eng.abandon(N.slot);
                    N.slot = -1;
                } else if (mode == 2) {
                    // the task's pieces take consecutive numbers of the handle's piece ring (stage_common's one-call tasks too)
                    const int rrc = ring_reserve(h, N.per);
                    if (rrc != BLZ_OK) {
                        h->eng.abandon(N.slot);
                        return rrc;
                    }
                    N.ring_first = h->ring.next;
                    h->ring.next += (uint64_t)N.pieces;
                }
            }
        }
        if (N.slot < 0) {
            N.npts = total * h->pf; N.sbits = h->pf == 1 ? 256 : 32; N.ppe = h->pf; N.even = false;
            if (mode == 2) {   // launched whole behind its last slice: buffers of the task's size (small tasks, BLAZE_MSM_PIECES=1)
                BLZ_TRY(h->points_raw[N.set].reserve((size_t)N.npts * ps));
                BLZ_TRY(h->points_mont.reserve((size_t)N.npts * mp));
            }
        }
        h->stage_idx ^= 1;
        N.open = true;
        S = N;
        BLZ_LOG(2, "streamed task: %u elements, mode %d (%s), %s", total, mode, mode == 1 ? "scalars over arena bases" : mode == 2 ? "points + scalars" : "points into the arena + scalars",
                S.slot >= 0 ? "enqueued in pieces as the slices arrive" : "launched whole behind its last slice");
        if (S.slot >= 0) BLZ_LOG(2, "streamed task: %d pieces of %u points%s", S.pieces, S.per, mode == 2 ? " through the piece ring" : "");
    } else {
        if (mode != S.mode || on_device != S.src_device)
            return fail(BLZ_ERR_INVALID_PARAM, "the queued task is being fed in another mode (points / hbm_point_addr / host or device pointers differ from its first slice)");
        if (mode == 1 && hbm_addr + hbm_off != S.arena_pos)
            return fail(BLZ_ERR_INVALID_PARAM, "hbm_point_addr differs from the first slice's (a streamed task names the address of its FIRST base in every slice)");
    }
    if (mode == 3) {
        // msm_api.rs:203-206: load_data_to_hbm(points, addr, offset) first - the slices' tables back to back
        const uint64_t want = S.arena_pos + (uint64_t)S.received * h->pf * ps;
        if (hbm_addr + hbm_off != want)
            return fail(BLZ_ERR_INVALID_PARAM, "slice of a streamed task: its points go to %llu, behind the %u elements already loaded at %llu (got %llu)",
                        (unsigned long long)want, S.received, (unsigned long long)S.arena_pos, (unsigned long long)(hbm_addr + hbm_off));
    }
    // ---- this slice's bytes.  The caller may drop its buffers as soon as we return (set_data is synchronous: utils.rs:71): bounded
    // waits; a failure from here on loses the task (bytes of it may be missing): the stream is given up, the task stays armed
    // and may be sent again from its first element
    int rc = BLZ_OK;
    auto land = [&](const char* what) {   // what was enqueued must land before the caller's buffers go - also when an enqueue failed
        wait_clear();
        const int wrc = sync_stream_bounded(cst, what);
        if (wrc != BLZ_OK && wait_timed_out()) h->wedged = true;
        if (rc == BLZ_OK) rc = wrc;
    };
    if (m && mode == 2 && S.slot >= 0) {
        // points + scalars of a task enqueued in pieces: the scalars to their place in the task's buffer, the points PART BY PART into
        // the ring slots of the pieces they belong to (a slice may end in the middle of a piece, or span several), each part's pieces
        // handed to the engine before the next part is copied - a ring slot is only free once its previous piece has been consumed
        char* d_sc = (char*)h->scalars_buf[S.set].p + (size_t)S.received * BLZ_SCALAR_SIZE;
        if (hipMemcpyAsync(d_sc, scalars, scalars_len, hipMemcpyDefault, cst) != hipSuccess) rc = fail_hip(BLZ_ERR_WRITE, "set_data: copy of the scalars failed");
        uint32_t off = 0;   // elements of this slice already copied
        while (rc == BLZ_OK && off < m) {
            const uint64_t at_pts = (uint64_t)S.received * S.ppe;          // (S.received moves with every part)
            const uint32_t k = (uint32_t)(at_pts / S.per), fill = (uint32_t)(at_pts % S.per);
            uint32_t take = (S.per - fill) / S.ppe;                        // elements that still fit into piece k (per is a multiple of 16 points: whole elements)
            if (take > m - off) take = m - off;
            char *d_raw = nullptr, *d_mont = nullptr;
            (void)ring_slot(h, S.ring_first + k, &d_raw, &d_mont);
            if (fill == 0) rc = ring_wait_free(h, S.ring_first + k);
            if (rc == BLZ_OK && hipMemcpyAsync(d_raw + (size_t)fill * ps, (const char*)points + (size_t)off * h->pf * ps, (size_t)take * h->pf * ps, hipMemcpyDefault, cst) != hipSuccess)
                rc = fail_hip(BLZ_ERR_WRITE, "set_data: copy of the points failed");
            land("set_data: copy of a slice of the task");
            if (rc != BLZ_OK) break;
            S.received += take;
            off += take;
            rc = stream_pump(h);
        }
        if (rc != BLZ_OK && off == 0) land("set_data: copy of a slice of the task");
    } else if (m) {
        char* d_sc = (char*)h->scalars_buf[S.set].p + (size_t)S.received * BLZ_SCALAR_SIZE;
        if (hipMemcpyAsync(d_sc, scalars, scalars_len, hipMemcpyDefault, cst) != hipSuccess) rc = fail_hip(BLZ_ERR_WRITE, "set_data: copy of the scalars failed");
        if (rc == BLZ_OK && mode == 2) {
            char* d_raw = (char*)h->points_raw[S.set].p + (size_t)S.received * h->pf * ps;
            if (hipMemcpyAsync(d_raw, points, points_len, hipMemcpyDefault, cst) != hipSuccess) rc = fail_hip(BLZ_ERR_WRITE, "set_data: copy of the points failed");
        }
        land("set_data: copy of a slice of the task");
        if (rc == BLZ_OK && mode == 3) {
            wait_clear();
            rc = arena_write(h->device, hbm_addr + hbm_off, points, points_len, on_device, h->eng.stream);
            if (rc != BLZ_OK && wait_timed_out()) h->wedged = true;
            if (rc == BLZ_OK) { h->bases_from_hbm = true; h->hbm_addr = hbm_addr; }
        }
    }
    if (rc == BLZ_OK && !(mode == 2 && S.slot >= 0)) {
        S.received += m;
        rc = stream_pump(h);
    }
    if (rc != BLZ_OK) {
        stream_abandon(h);
        return rc;
    }
    if (S.received < S.total) return BLZ_OK;
    // ---- the last slice is in
    h->d_scalars = h->scalars_buf[S.set].p;
    h->staged_n = S.total;
    if (S.slot >= 0) {
        rc = h->eng.end(S.slot);
        if (rc != BLZ_OK) {
            stream_abandon(h);
            return rc;
        }
        h->d_points_mont = nullptr;
        h->set_used[S.set] = true;
        h->staged_set = -1;
        h->armed = false;
        h->data_ready = false;
        h->in_flight.push_back({S.slot, h->task_label});
        S = blz_msm::Stream();
        return BLZ_OK;
    }
    // launched whole (stage_common's last steps)
    h->staged_set = S.set;
    h->staged_from_arena = S.mode != 2;
    h->staged_arena_pos = S.arena_pos;
    h->staged_loaded_now = S.mode == 3;
    if (S.mode == 2) {
        rc = h->eng.points_to_mont(h->points_raw[S.set].p, h->points_mont.p, S.npts);
        if (rc != BLZ_OK) {
            stream_abandon(h);
            return rc;
        }
        h->d_points_mont = h->points_mont.p;
    }
    S = blz_msm::Stream();
    h->data_ready = true;
    rc = launch_if_ready(h);
    if (rc != BLZ_OK) h->data_ready = false;   // (refused at launch - bases gone from the arena, say: the task stays armed, its data is not kept)
    return rc;
}

}  // namespace blz
