// C ABI of the MSM primitive: the DriverPrimitive call sequence of src/ingo_msm/msm_api.rs
// (initialize -> start_process -> set_data -> wait_result -> result) over the device pipeline.
#include <deque>

#include "msm_engine.hpp"

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
    // staged input
    DevBuf scalars_buf, points_raw, points_mont;
    hipStream_t copy_stream = nullptr;  // host -> device staging: runs under the previous task's accumulation
    const void* d_scalars = nullptr;
    const void* d_points_mont = nullptr;
    uint32_t staged_n = 0;
    MsmEngine eng;
};

namespace {

size_t point_size(const blz_msm* h) { return blz_point_size(h->curve); }
size_t result_size(const blz_msm* h) { return blz_result_size(h->curve); }

// resolve the Montgomery-form view of `npts` points stored at arena offset `pos`
int arena_points_mont(blz_msm* h, uint64_t pos, uint32_t npts, const void** out) {
    size_t len = (size_t)npts * point_size(h);
    Arena& A = arena_for(h->device);
    std::lock_guard<std::mutex> lk(A.mu);
    ArenaExtent* e = arena_find(A, pos, len);
    if (!e)
        return fail(BLZ_ERR_INVALID_PARAM, "HBM bases: no loaded extent covers [%llu, +%zu) on device %d",
                    (unsigned long long)pos, len, h->device);
    // the shadow holds the Montgomery form of the extent's whole points, one per mont_point_bytes()
    if ((pos - e->start) % point_size(h) != 0)
        return fail(BLZ_ERR_INVALID_PARAM, "HBM point address must be a whole number of points into its loaded extent");
    const size_t mp = mont_point_bytes(h->curve);
    if (e->mont_curve != h->curve) {
        uint32_t ext_pts = (uint32_t)(e->len / point_size(h));
        if (e->mont) { (void)hipFree(e->mont); e->mont = nullptr; }   // stride differs between curves
        BLZ_HIP(hipMalloc(&e->mont, (size_t)ext_pts * mp + 16), BLZ_ERR_UNKNOWN);
        BLZ_TRY(h->eng.points_to_mont(e->raw, e->mont, ext_pts));
        e->mont_curve = h->curve;
    }
    *out = (const char*)e->mont + (pos - e->start) / point_size(h) * mp;
    return BLZ_OK;
}

int launch_if_ready(blz_msm* h) {
    if (!(h->armed && h->data_ready)) return BLZ_OK;
    if (!h->eng.can_accept())
        return fail(BLZ_ERR_INVALID_PARAM, "task queue full (%d in flight); call wait_result first", MSM_QUEUE_DEPTH);
    uint32_t npts = h->staged_n * h->pf;
    int sbits = h->pf == 1 ? 256 : 32;
    int slot = 0;
    BLZ_TRY(h->eng.run(h->d_points_mont, h->d_scalars, npts, sbits, &slot));
    h->armed = false;
    h->data_ready = false;
    h->in_flight.push_back({slot, h->task_label});
    return BLZ_OK;
}

int stage_common(blz_msm* h, bool have_points, const void* points, size_t points_len, const void* scalars,
                 size_t scalars_len, uint32_t n, int has_hbm, uint64_t hbm_addr, uint64_t hbm_off, bool on_device) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    BLZ_TRY(use_device(h->device));
    if (!have_points && !has_hbm) return BLZ_OK;  // reference: falls through every branch (msm_api.rs:163-216)
    if (!scalars && n) return fail(BLZ_ERR_INVALID_PARAM, "null scalars");
    if (scalars_len != (size_t)n * BLZ_SCALAR_SIZE)
        return fail(BLZ_ERR_INVALID_PARAM, "scalars length %zu != nof_elements %u * 32", scalars_len, n);
    size_t want_pts = (size_t)n * h->pf * point_size(h);
    if (have_points && points_len != want_pts)
        return fail(BLZ_ERR_INVALID_PARAM, "points length %zu != nof_elements %u * precompute_factor %u * %zu", points_len,
                    n, h->pf, point_size(h));
    if ((uint64_t)n * h->pf >= (1ull << 31)) return fail(BLZ_ERR_INVALID_PARAM, "too many points");
    if (!h->eng.can_accept())
        return fail(BLZ_ERR_INVALID_PARAM, "task queue full (%d in flight); call wait_result first", MSM_QUEUE_DEPTH);
    hipStream_t st = h->eng.stream;
    // Host buffers are staged on their own stream, so the PCIe transfer of this task overlaps the
    // accumulation of the task in flight (the reference's DMA writes overlap device compute the same
    // way).  The staging buffers are free: everything that read them (to-Montgomery,
    // digit sort) had completed when the previous set_data returned.
    hipStream_t cst = h->copy_stream;
    uint32_t npts = n * h->pf;

    if (have_points && has_hbm) {
        // msm_api.rs:203-206: load_data_to_hbm(points, addr, offset) first
        BLZ_TRY(arena_write(h->device, hbm_addr + hbm_off, points, points_len, on_device, st));
        h->bases_from_hbm = true;
        h->hbm_addr = hbm_addr;
    }
    if (has_hbm) {
        // bases come from the arena.  The reference's initialize() programs only hbm_point_addr.0
        // as the start address (msm_api.rs:84-95) while load_data_to_hbm writes at addr+offset
        // (msm_api.rs:312); both tests use offset 0.  Here the task reads where the load wrote.
        BLZ_TRY(arena_points_mont(h, hbm_addr + hbm_off, npts, &h->d_points_mont));
    } else {
        const size_t want_mont = (size_t)npts * mont_point_bytes(h->curve);
        BLZ_TRY(h->points_mont.reserve(want_mont ? want_mont : 16));
        if (on_device) {
            if (((uintptr_t)points) % 16) return fail(BLZ_ERR_INVALID_PARAM, "device points must be 16-byte aligned");
            BLZ_TRY(h->eng.points_to_mont(points, h->points_mont.p, npts));
        } else {
            BLZ_TRY(h->points_raw.reserve(want_pts ? want_pts : 16));
            if (want_pts) BLZ_HIP(hipMemcpyAsync(h->points_raw.p, points, want_pts, hipMemcpyHostToDevice, cst), BLZ_ERR_WRITE);
            BLZ_HIP(hipStreamSynchronize(cst), BLZ_ERR_WRITE);
            BLZ_TRY(h->eng.points_to_mont(h->points_raw.p, h->points_mont.p, npts));
        }
        h->d_points_mont = h->points_mont.p;
    }
    if (on_device) {
        if (((uintptr_t)scalars) % 16) return fail(BLZ_ERR_INVALID_PARAM, "device scalars must be 16-byte aligned");
        h->d_scalars = scalars;
    } else {
        BLZ_TRY(h->scalars_buf.reserve(scalars_len ? scalars_len : 16));
        if (scalars_len) BLZ_HIP(hipMemcpyAsync(h->scalars_buf.p, scalars, scalars_len, hipMemcpyHostToDevice, cst), BLZ_ERR_WRITE);
        h->d_scalars = h->scalars_buf.p;
        // the caller may drop its buffers as soon as we return (set_data is synchronous: utils.rs:71)
        BLZ_HIP(hipStreamSynchronize(cst), BLZ_ERR_WRITE);
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
    int rc = h->eng.init(device_id, curve);
    if (rc == BLZ_OK && hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess)
        rc = fail(BLZ_ERR_UNKNOWN, "copy stream creation failed");
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
    h->eng.destroy();
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    h->scalars_buf.release();
    h->points_raw.release();
    h->points_mont.release();
    delete h;
}

int blz_msm_loaded_binary_parameters(blz_msm* h, uint32_t out[2]) {
    if (!h || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    // image id: 'MI35'; parameters packed like MSMImageParametrs (msm_api.rs:333-347, msb0 numbering):
    // [31:28]... the reference decodes with packed_struct; we expose curve in bits 11:4 and the
    // number of "EC adders" (compute units) in bits 3:0 scaled by 16.
    out[0] = 0x4D493335u;
    uint32_t curve_code = h->curve == BLZ_BLS377 ? 0u : h->curve == BLZ_BN254 ? 1u : 2u;
    out[1] = (curve_code << 4) | 0x0u;
    return BLZ_OK;
}

int blz_msm_initialize(blz_msm* h, uint32_t nof_elements, int has_hbm, uint64_t hbm_addr, uint64_t hbm_off) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
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
    if (h->in_flight.empty()) {
        if (!h->results.empty()) return BLZ_OK;  // RESULT_VALID already set
        return fail(BLZ_ERR_INVALID_PARAM, "wait_result with no task in flight (the reference would spin forever)");
    }
    // tasks complete in submission order: wait for the oldest, move its bytes to the result queue
    blz_msm::Pending p = h->in_flight.front();
    h->in_flight.pop_front();
    blz_msm::Res r;
    r.bytes.resize(result_size(h));
    r.label = p.label;
    int rc = h->eng.finish(p.slot, r.bytes.data());
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
    BLZ_TRY(arena_write(h->device, addr + off, points, len, false, h->eng.stream));
    h->bases_from_hbm = true;  // msm_api.rs:301-311 flips BASES_SOURCE and programs the address
    h->hbm_addr = addr;
    return BLZ_OK;
}

int blz_msm_load_data_to_hbm_device(blz_msm* h, const void* d_points, size_t len, uint64_t addr, uint64_t off) {
    if (!h || (!d_points && len)) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    BLZ_TRY(arena_write(h->device, addr + off, d_points, len, true, h->eng.stream));
    h->bases_from_hbm = true;
    h->hbm_addr = addr;
    return BLZ_OK;
}

int blz_msm_get_data_from_hbm(blz_msm* h, uint8_t* out, size_t len, uint64_t addr, uint64_t off) {
    if (!h || (!out && len)) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    BLZ_TRY(use_device(h->device));
    Arena& A = arena_for(h->device);
    std::lock_guard<std::mutex> lk(A.mu);
    ArenaExtent* e = arena_find(A, addr + off, len);
    if (!e) return fail(BLZ_ERR_READ, "no loaded extent covers [%llu, +%zu)", (unsigned long long)(addr + off), len);
    BLZ_HIP(hipMemcpy(out, (const char*)e->raw + (addr + off - e->start), len, hipMemcpyDeviceToHost), BLZ_ERR_READ);
    return BLZ_OK;
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
    BLZ_TRY(h->eng.sync_all());
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
    return h->eng.combine_partials(partials, count, out);
}

}  // extern "C"
