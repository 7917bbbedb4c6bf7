// C ABI of the MSM primitive: the DriverPrimitive call sequence of src/ingo_msm/msm_api.rs
// (initialize -> start_process -> set_data -> wait_result -> result) over the device pipeline.
#include <chrono>
#include <thread>

#include "msm_handle.hpp"

using namespace blz;

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
        rc = fail_hip(BLZ_ERR_UNKNOWN, "copy stream creation failed");
    for (int i = 0; i < 2 && rc == BLZ_OK; ++i)
        if (hipEventCreateWithFlags(&h->set_free[i], hipEventDisableTiming) != hipSuccess)
            rc = fail_hip(BLZ_ERR_UNKNOWN, "event creation failed");
    for (int i = 0; i < blz_msm::PieceRing::SLOTS && rc == BLZ_OK; ++i)
        if (hipEventCreateWithFlags(&h->ring.raw_read[i], hipEventDisableTiming) != hipSuccess)
            rc = fail_hip(BLZ_ERR_UNKNOWN, "event creation failed");
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
    h->ring.raw.release();
    h->ring.mont.release();
    for (hipEvent_t& e : h->ring.raw_read)
        if (e) (void)hipEventDestroy(e);
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
        task_repr_bn254pc(h, true, checked);
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
    if (h->armed && h->strm.open)
        return fail(BLZ_ERR_INVALID_PARAM, "a task is already queued and has received %u of its %u elements", h->strm.received, h->strm.total);
    if (h->armed) return fail(BLZ_ERR_INVALID_PARAM, "a task is already queued and waits for data");
    h->armed = true;
    h->task_label += 1;
    if (h->data_ready && h->staged_n != h->nof_elements)
        return fail(BLZ_ERR_INVALID_PARAM, "staged data has %u elements, task expects %u", h->staged_n, h->nof_elements);
    return launch_if_ready(h);
}

int blz_msm_set_data(blz_msm* h, const uint8_t* points, size_t points_len, const uint8_t* scalars, size_t scalars_len,
                     uint32_t nof_elements, int has_hbm, uint64_t hbm_addr, uint64_t hbm_off) {
    // with a task queued, fewer elements than the task still lacks = the next slice of it (msm_stage.hip stage_stream: the card
    // counts what its FIFOs receive against NUMBER_OF_MSM_ELEMENTS, msm_api.rs:155-202); more is refused there
    if (h && h->armed && (h->strm.open || nof_elements != h->nof_elements))
        return stage_stream(h, points != nullptr, points, points_len, scalars, scalars_len, nof_elements, has_hbm, hbm_addr, hbm_off, false);
    return stage_common(h, points != nullptr, points, points_len, scalars, scalars_len, nof_elements, has_hbm, hbm_addr,
                        hbm_off, false);
}

int blz_msm_set_data_device(blz_msm* h, const void* d_points, size_t points_len, const void* d_scalars,
                            size_t scalars_len, uint32_t nof_elements, int has_hbm, uint64_t hbm_addr, uint64_t hbm_off) {
    if (h && h->armed && (h->strm.open || nof_elements != h->nof_elements))
        return stage_stream(h, d_points != nullptr, d_points, points_len, d_scalars, scalars_len, nof_elements, has_hbm, hbm_addr, hbm_off, true);
    return stage_common(h, d_points != nullptr, d_points, points_len, d_scalars, scalars_len, nof_elements, has_hbm,
                        hbm_addr, hbm_off, true);
}

int blz_msm_wait_result(blz_msm* h) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    BLZ_LIVE(h);
    if (h->in_flight.empty()) {
        if (!h->results.empty()) return BLZ_OK;  // RESULT_VALID already set
        if (h->strm.open)
            return fail(BLZ_ERR_INVALID_PARAM, "wait_result: the queued task has received %u of its %u elements (the reference would spin forever)",
                        h->strm.received, h->strm.total);
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
int blz_msm_stream_progress(blz_msm* h, uint32_t out[2]) {
    if (!h || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    out[0] = h->strm.open ? h->strm.received : 0u;
    out[1] = h->armed ? h->nof_elements : 0u;
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
    stream_abandon(h);   // (a half-fed task goes with the reset)
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
    uint64_t staging = h->points_mont.cap + h->comm_buf.cap + h->ring.raw.cap + h->ring.mont.cap;
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

int blz_msm_stream(blz_msm* h, void** hip_stream, int* device_id) {
    if (!h || !hip_stream) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    if (device_id) *device_id = h->device;
    *hip_stream = (void*)h->eng.stream;
    return BLZ_OK;
}

}  // extern "C"
