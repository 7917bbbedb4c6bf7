// Radix-2^k NTT over a scalar field (BLS12-381 Fr by default; BLS12-377 / BN254 Fr on request) for gfx950 (device task of SURVEY.md a16: what the
// bitstream behind src/ingo_ntt/ntt_hw_code.rs:6-83 computes on one 2^27 x 32 B buffer).
//
//   X[k] = sum_i x[i] w^(ik),  natural order in and out,  w = 7^((r-1)/2^log_size)
//
// n = A*B*C (each <= 512).  With i = i0 + A i1 + AB i2 and k = k2 + C k1 + CB k0:
//   pass 1: C-point NTTs over i2 (stride AB), then * w^(A i1 k2)
//   pass 2: B-point NTTs over i1 (stride A),  then * w^(i0 (k2 + C k1))
//   pass 3: A-point NTTs over i0 (contiguous), written to the natural address k2 + C k1 + CB k0
// A pass stages a tile of COLS adjacent columns x radix rows in LDS (COLS*32 B contiguous per row
// access), runs the radix-2 stages there and applies the inter-pass twiddle on the way out.
// Data stay in plain canonical form; only twiddles are in Montgomery form (mont_mul(x, tR) = x t),
// so there is no conversion pass.  Algorithmic traffic 2 x 4 GiB; this 3-pass form moves 3x that.
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"
#include "ntt_engine.hpp"

namespace blz {


// NTTBanks::preprocess / postprocess (src/ingo_ntt/ntt_data.rs:80-156) as device permutations of
// 32-byte elements, from the closed forms of the reference loops:
//   preprocess : element e = 512 blk + j  ->  bank (blk%2)*8 + j%8, slot (blk/2)*64 + j/8
//   postprocess: output a = 512 isub + i, isub = ic*G + g + 2G*blk
//                <- bank ((ic ^ (g&1))*8 + i%8), slot (g*Bg + blk)*64 + i/8      (G groups, Bg blocks each)
__global__ __launch_bounds__(256) void k_ntt_banks_pre(const uint4* __restrict__ in, uint4* __restrict__ banks, uint64_t n) {
    uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    uint64_t blk = e >> 9, j = e & 511;
    uint64_t bank = (blk & 1) * 8 + (j & 7), slot = (blk >> 1) * 64 + (j >> 3);
    uint64_t dst = bank * (n >> 4) + slot;
    banks[2 * dst] = in[2 * e];
    banks[2 * dst + 1] = in[2 * e + 1];
}
__global__ __launch_bounds__(256) void k_ntt_banks_post(const uint4* __restrict__ banks, uint4* __restrict__ out, uint64_t n,
                                                        uint64_t G, uint64_t Bg) {
    uint64_t a = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= n) return;
    uint64_t isub = a >> 9, i = a & 511;
    uint64_t blk = isub / (2 * G), ic = (isub % (2 * G)) / G, g = isub % G;
    uint64_t bank = ((ic ^ (g & 1)) * 8) + (i & 7), slot = (g * Bg + blk) * 64 + (i >> 3);
    uint64_t src = bank * (n >> 4) + slot;
    out[2 * a] = banks[2 * src];
    out[2 * a + 1] = banks[2 * src + 1];
}

#ifdef BLZ_EXPERIMENT_KNOBS
__global__ __launch_bounds__(256) void k_copy16(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
#endif

}  // namespace blz

using namespace blz;

struct blz_ntt {
    int device = 0;
    int field = BLZ_BLS381;  // scalar field of this curve (enum blz_curve)
    const NttFieldOps* ops = nullptr;
    int logn = 27;
    int inverse = 0;
    bool force_generic = false;  // (experiment builds, BLAZE_NTT_GENERIC=1: radix-2-in-LDS kernel for every pass)
    NttGeom geom{};
    int cols_log[3] = {0, 0, 0};
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // host<->buffer traffic, concurrent with the compute stream
    hipStream_t copy_stream2 = nullptr; // blz_ntt_exchange: the host -> device direction, while copy_stream carries device -> host
    uint32_t flags = 0;                 // blz_ntt_new_ex2 / _ex3
    bool has_root = false;              // blz_ntt_new_ex3: the caller's primitive 2^logn-th root (canonical little-endian words)
    uint8_t root[32] = {};
    hipEvent_t xchg_ev[4] = {nullptr, nullptr, nullptr, nullptr};   // blz_ntt_exchange with pinned host buffers: piece k has left
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    DevBuf buf[2], scratch, tables, tables_rr, table_b;
    NttTables T{};
    NttTablesRR TR{};
    bool in_flight = false;
    int in_flight_buf = -1;   // buffer under transform while in_flight
    float last_ms = 0.f;
    bool wedged = false;      // a wait ran into BLAZE_WAIT_TIMEOUT_MS: only reset / free are accepted (common.hpp)
};

#define BLZ_NTT_LIVE(h)                                                                                          \
    do {                                                                                                         \
        if ((h)->wedged)                                                                                         \
            return fail(BLZ_ERR_UNKNOWN, "handle is wedged: an earlier wait timed out (BLAZE_WAIT_TIMEOUT_MS); only " \
                                         "reset / free are accepted");                                           \
    } while (0)
#define BLZ_NTT_WAIT(h, expr)                      \
    do {                                           \
        blz::wait_clear();                         \
        int rc__ = (expr);                         \
        if (rc__ != BLZ_OK) {                      \
            if (blz::wait_timed_out()) (h)->wedged = true; \
            return rc__;                           \
        }                                          \
    } while (0)

namespace {

size_t ntt_bytes(const blz_ntt* h) { return (size_t)32 << h->logn; }

const NttFieldOps* ntt_ops_for(int field) {
    switch (field) {
        case BLZ_BLS377: return &ntt_ops_bls377();
        case BLZ_BLS381: return &ntt_ops_bls381();
        case BLZ_BN254: return &ntt_ops_bn254();
    }
    return nullptr;
}

int ntt_setup(blz_ntt* h) {
    // split logn into three radices, largest last-pass first so the contiguous pass is wide
    int l = h->logn;
    int la = l > 9 ? 9 : l;
    int lb = (l - la) > 9 ? 9 : (l - la);
    int lc = l - la - lb;
    if (lc > 9) return fail(BLZ_ERR_INVALID_PARAM, "log_size %d not supported (max 27)", l);
    h->geom = NttGeom{la, lb, lc, l, lc ? 1 : lb ? 2 : 3, (h->flags & BLZ_NTT_BITREV_INPUT) ? 1 : 0, (h->flags & BLZ_NTT_BITREV_OUTPUT) ? 1 : 0};
    // columns per tile: LDS = radix * (COLS + 1) * 32 B <= 160 KiB, and COLS <= extent of the column index
    auto pick = [](int lr, int lcols_avail) {
        int c = 17 - 5 - lr;  // log2(128 KiB / 32 B / radix)
        if (c > 3) c = 3;
        if (c > lcols_avail) c = lcols_avail;
        if (c < 0) c = 0;
        return c;
    };
    h->cols_log[0] = pick(lc, la);  // pass 1: cols i0 (< A)
    h->cols_log[1] = pick(lb, la);  // pass 2: cols i0 (< A)
    h->cols_log[2] = pick(la, lc);  // pass 3: cols k2 (< C)
    size_t tb = (size_t)(3 * 512 + 3 * 512 + 1 + 1 + 2) * 32;   // wpass x 3, t0..t2, ninv, wbase, (the caller's root | the root check's flag)
    BLZ_TRY(h->tables.reserve(tb));
    uint32_t* p = h->tables.as<uint32_t>();
    for (int i = 0; i < 3; ++i) { h->T.wpass[i] = p; p += 512 * 8; }
    h->T.t0 = p; p += 512 * 8;
    h->T.t1 = p; p += 512 * 8;
    h->T.t2 = p; p += 512 * 8;
    h->T.ninv = h->inverse ? p : nullptr; p += 8;
    h->T.wbase = p; p += 8;
    uint32_t* const d_user_root = p; p += 8;
    uint32_t* const d_root_flag = p; p += 8;
    BLZ_HIP(hipMemsetAsync(d_root_flag, 0, 4, h->stream), BLZ_ERR_UNKNOWN);
    if (h->has_root) BLZ_HIP(hipMemcpyAsync(d_user_root, h->root, 32, hipMemcpyHostToDevice, h->stream), BLZ_ERR_WRITE);
    // the boundary table exists only where all three passes run the 512-point kernel (2^27), so that the factor it
    // splits off is re-joined by the same kernel in pass 2
    const bool want_ta = lc == 9 && !h->force_generic && exp_knob("BLAZE_NTT_TABLE", 1) != 0;
    BLZ_TRY(h->tables_rr.reserve(NTT_RR_TABLE_BYTES + (want_ta ? NTT_RR_BOUNDARY_BYTES : 0)));
    {
        uint32_t* q = h->tables_rr.as<uint32_t>();
        for (int i = 0; i < 3; ++i) { h->TR.wpass[i] = q; q += 2 * 512 * NTT_RR_ENTRY_DWORDS; }   // Shoup entries
        h->TR.t0 = q; q += 512 * NTT_RR_ENTRY_DWORDS;
        h->TR.t1 = q; q += 512 * NTT_RR_ENTRY_DWORDS;
        h->TR.t2 = q; q += 512 * NTT_RR_ENTRY_DWORDS;
        h->TR.fin = h->inverse ? q : nullptr; q += NTT_RR_ENTRY_DWORDS;
        h->TR.ts2 = q; q += 2 * 512 * NTT_RR_ENTRY_DWORDS;   // Shoup entries
        h->TR.tA = want_ta ? q : nullptr;
        // pass 2's boundary factors, one per element (4 GiB at 2^27: the transform's buffers are 8): only beside tA, whose
        // split of pass 1's factor it completes
        const bool want_tb = want_ta && !(h->flags & BLZ_NTT_NO_FACTOR_TABLE) && exp_knob("BLAZE_NTT_TB", 1) != 0;
        h->TR.tB = nullptr;
        if (want_tb) {
            // (an optimisation, not a need: a device too full for it steps the factors as smaller transforms do)
            if (h->table_b.reserve(ntt_bytes(h), true) == BLZ_OK) h->TR.tB = h->table_b.as<uint32_t>();
            else BLZ_LOG(1, "NTT: no memory for the boundary-factor table (%zu bytes): pass 2 steps its factors", ntt_bytes(h));
        }
        // pass 1 tile order (ntt_rr.hip.hpp): 0 plain, 1 + s: 2^s adjacent column groups back to back (s = 3), + 16 b: b bits of
        // i1 walked first (default 7)
        h->TR.swz = (la == 9 && lb == 9) ? (uint32_t)exp_knob("BLAZE_NTT_SWZ", 4) : 0u;
        if ((h->TR.swz & 15u) > 6u) h->TR.swz = 6u;
    }
    BLZ_TRY(h->ops->setup(h->stream, h->T, h->TR, h->geom, h->inverse, h->has_root ? d_user_root : nullptr, d_root_flag));
    if (h->has_root) {
        // the caller's root was checked on the device ahead of the tables built from it (k_ntt_root)
        uint32_t bad = 0;
        BLZ_TRY(sync_stream_bounded(h->stream, "NTT set-up: root check"));
        BLZ_HIP(hipMemcpy(&bad, d_root_flag, 4, hipMemcpyDeviceToHost), BLZ_ERR_READ);
        if (bad)
            return fail(BLZ_ERR_INVALID_PARAM, bad == 1 ? "the root is not a canonical element of the field (>= r)"
                                                        : "the root is not a primitive 2^%d-th root of unity (root^(2^%d) != -1)", h->logn, h->logn - 1);
    }
    // both transform buffers exist from the start, zero-filled, like the card's two HBM buffers: the reference's
    // double-buffer loop opens with start_process on a buffer nobody wrote and result on the other
    // (tests/integration_ntt.rs:102-136, cycle 0)
    for (int b = 0; b < 2; ++b) {
        BLZ_TRY(h->buf[b].reserve(ntt_bytes(h), true));   // (a transform's size is fixed: no growth slack - it was 1.5 GiB at 2^27)
        BLZ_HIP(hipMemsetAsync(h->buf[b].p, 0, ntt_bytes(h), h->stream), BLZ_ERR_UNKNOWN);
    }
    BLZ_TRY(h->scratch.reserve(ntt_bytes(h), true));
    BLZ_TRY(sync_stream_bounded(h->stream, "NTT set-up: tables and zero-filled buffers"));
    return BLZ_OK;
}

int launch_pass(blz_ntt* h, int pass, const void* in, void* out) {
    return h->ops->pass(pass, h->stream, in, out, h->geom, h->T, h->TR, h->cols_log[pass - 1], h->force_generic);
}

}  // namespace

extern "C" {

int blz_ntt_new(int device_id, int log_size, blz_ntt** out) { return blz_ntt_new_ex(device_id, log_size, 0, out); }

int blz_ntt_new_ex(int device_id, int log_size, int inverse, blz_ntt** out) {
    return blz_ntt_new_field(device_id, BLZ_BLS381, log_size, inverse, out);
}

int blz_ntt_new_field(int device_id, int field, int log_size, int inverse, blz_ntt** out) {
    return blz_ntt_new_ex2(device_id, field, log_size, inverse, 0u, out);
}

int blz_ntt_new_ex2(int device_id, int field, int log_size, int inverse, uint32_t flags, blz_ntt** out) {
    if (flags & ~(uint32_t)BLZ_NTT_NO_FACTOR_TABLE) return fail(BLZ_ERR_INVALID_PARAM, "unknown NTT flags 0x%x", flags);
    return blz_ntt_new_ex3(device_id, field, log_size, flags | (inverse ? BLZ_NTT_INVERSE : 0u), nullptr, out);
}

int blz_ntt_new_ex3(int device_id, int field, int log_size, uint32_t flags, const uint8_t* root, blz_ntt** out) {
    if (!out) return fail(BLZ_ERR_INVALID_PARAM, "null out");
    *out = nullptr;
    if (flags & ~(uint32_t)(BLZ_NTT_NO_FACTOR_TABLE | BLZ_NTT_INVERSE | BLZ_NTT_BITREV_INPUT | BLZ_NTT_BITREV_OUTPUT))
        return fail(BLZ_ERR_INVALID_PARAM, "unknown NTT flags 0x%x", flags);
    const int inverse = (flags & BLZ_NTT_INVERSE) ? 1 : 0;
    const NttFieldOps* ops = ntt_ops_for(field);
    if (!ops) return fail(BLZ_ERR_INVALID_PARAM, "unknown field %d", field);
    if (log_size < 1 || log_size > 27) return fail(BLZ_ERR_INVALID_PARAM, "log_size %d out of range [1,27]", log_size);
    if (log_size > ops->two_adicity)
        return fail(BLZ_ERR_INVALID_PARAM, "log_size %d exceeds the two-adicity %d of the field", log_size, ops->two_adicity);
    BLZ_TRY(use_device(device_id));
    blz_ntt* h = new blz_ntt();
    h->device = device_id;
    h->field = field;
    h->ops = ops;
    h->logn = log_size;
    h->inverse = inverse ? 1 : 0;
    h->flags = flags;
    if (root) {
        h->has_root = true;
        memcpy(h->root, root, 32);
    }
    h->force_generic = exp_knob("BLAZE_NTT_GENERIC", 0) != 0;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->copy_stream2, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&h->xchg_ev[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    int rc = e == hipSuccess ? ntt_setup(h) : fail_hip(BLZ_ERR_UNKNOWN, "stream/event creation failed: %s", hipGetErrorString(e));
    if (rc != BLZ_OK) {
        blz_ntt_free(h);
        return rc;
    }
    *out = h;
    return BLZ_OK;
}

void blz_ntt_free(blz_ntt* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream && (sync_stream_bounded(h->stream, "free: NTT stream") != BLZ_OK ||
                      sync_stream_bounded(h->copy_stream, "free: NTT copy stream") != BLZ_OK ||
                      (h->copy_stream2 && sync_stream_bounded(h->copy_stream2, "free: NTT copy stream") != BLZ_OK))) {
        BLZ_LOG(0, "NTT handle freed while its device work is wedged: buffers and streams are leaked");
        delete h;
        return;
    }
    h->buf[0].release(); h->buf[1].release(); h->scratch.release(); h->tables.release(); h->tables_rr.release(); h->table_b.release();
    for (auto& e : h->xchg_ev)
        if (e) (void)hipEventDestroy(e);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->copy_stream2) (void)hipStreamDestroy(h->copy_stream2);
    delete h;
}

int blz_ntt_initialize(blz_ntt* h) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    return BLZ_OK;  // ntt_api.rs:37-56 writes debug-program registers; nothing to program here
}

static int ntt_set_data_common(blz_ntt* h, size_t buf_host, const void* data, size_t len, bool on_device) {
    if (!h || !data) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    if (buf_host > 1) return fail(BLZ_ERR_INVALID_PARAM, "buf_host must be 0 or 1");
    if (len != ntt_bytes(h)) return fail(BLZ_ERR_INVALID_PARAM, "data length %zu != %zu", len, ntt_bytes(h));
    BLZ_NTT_LIVE(h);
    if (h->in_flight && h->in_flight_buf == (int)buf_host)
        return fail(BLZ_ERR_INVALID_PARAM, "buffer %zu is being transformed; call wait_result first", buf_host);
    BLZ_TRY(use_device(h->device));
    // a dedicated copy stream: the compute stream may be busy on the other buffer (double-buffer
    // contract, tests/integration_ntt.rs:102-136).  Synchronised before returning: set_data is
    // blocking, and a device-to-device hipMemcpy alone is not ordered against other streams.
    BLZ_HIP(hipMemcpyAsync(h->buf[buf_host].p, data, len, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                           h->copy_stream), BLZ_ERR_WRITE);
    BLZ_NTT_WAIT(h, sync_stream_bounded(h->copy_stream, "set_data: copy into the NTT buffer"));
    return BLZ_OK;
}

int blz_ntt_set_data(blz_ntt* h, size_t buf_host, const uint8_t* data, size_t len) {
    return ntt_set_data_common(h, buf_host, data, len, false);
}
int blz_ntt_set_data_device(blz_ntt* h, size_t buf_host, const void* d_data, size_t len) {
    return ntt_set_data_common(h, buf_host, d_data, len, true);
}

int blz_ntt_start_process(blz_ntt* h, size_t buf_kernel) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    if (buf_kernel > 1) return fail(BLZ_ERR_INVALID_PARAM, "buf_kernel must be 0 or 1");
    BLZ_NTT_LIVE(h);
    if (h->in_flight) return fail(BLZ_ERR_INVALID_PARAM, "a transform is already running; call wait_result first");
    BLZ_TRY(use_device(h->device));
    void* b = h->buf[buf_kernel].p;
    void* s = h->scratch.p;
    BLZ_HIP(hipEventRecord(h->ev0, h->stream), BLZ_ERR_UNKNOWN);
    const void* cur = b;
    if (h->geom.logC) { BLZ_TRY(launch_pass(h, 1, cur, s)); cur = s; }
    if (h->geom.logB) { BLZ_TRY(launch_pass(h, 2, cur, s)); cur = s; }
    if (cur == b) {  // single pass: keep it out of place through the scratch
        BLZ_HIP(hipMemcpyAsync(s, b, ntt_bytes(h), hipMemcpyDeviceToDevice, h->stream), BLZ_ERR_UNKNOWN);
        cur = s;
    }
    BLZ_TRY(launch_pass(h, 3, cur, b));
    BLZ_HIP(hipEventRecord(h->ev1, h->stream), BLZ_ERR_UNKNOWN);
    h->in_flight = true;
    h->in_flight_buf = (int)buf_kernel;
    return BLZ_OK;
}

int blz_ntt_wait_result(blz_ntt* h) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    BLZ_NTT_LIVE(h);
    if (!h->in_flight) return fail(BLZ_ERR_INVALID_PARAM, "wait_result with no transform in flight");
    BLZ_TRY(use_device(h->device));
    // bounded (BLAZE_WAIT_TIMEOUT_MS): the reference polls the status register without a deadline (ntt_api.rs:89-108)
    BLZ_NTT_WAIT(h, sync_event_bounded(h->ev1, "wait_result: NTT"));
    (void)hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1);
    h->in_flight = false;
    return BLZ_OK;
}

static int ntt_result_common(blz_ntt* h, size_t buf, void* out, size_t out_cap, bool on_device) {
    if (!h || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    if (buf > 1) return fail(BLZ_ERR_INVALID_PARAM, "buffer index must be 0 or 1");
    if (h->in_flight && h->in_flight_buf == (int)buf)
        return fail(BLZ_ERR_INVALID_PARAM, "buffer %zu is being transformed; call wait_result first", buf);
    if (out_cap < ntt_bytes(h)) return fail(BLZ_ERR_INVALID_PARAM, "output buffer too small");
    BLZ_NTT_LIVE(h);
    BLZ_TRY(use_device(h->device));
    BLZ_HIP(hipMemcpyAsync(out, h->buf[buf].p, ntt_bytes(h), on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                           h->copy_stream), BLZ_ERR_READ);
    BLZ_NTT_WAIT(h, sync_stream_bounded(h->copy_stream, "result: copy out of the NTT buffer"));
    return BLZ_OK;
}
int blz_ntt_result(blz_ntt* h, size_t buf, uint8_t* out, size_t out_cap) { return ntt_result_common(h, buf, out, out_cap, false); }
int blz_ntt_result_device(blz_ntt* h, size_t buf, void* d_out, size_t out_cap) { return ntt_result_common(h, buf, d_out, out_cap, true); }

// result(buf) and set_data(buf) of the reference's double-buffered loop (tests/integration_ntt.rs:102-136: while the kernel runs
// on the other buffer the host READS the previous result out of `buf` and WRITES the next input into it) as ONE call that uses
// the link in both directions at once.  Called one after the other, the two copies are the whole cycle - 76 + 76.5 ms around a
// hidden 14 ms kernel at 2^27 - and each leaves the opposite direction of the full-duplex link idle.  Here the buffer goes in
// pieces: piece k leaves for prev_out on copy_stream, and as soon as it has, piece k of next_in lands in its place on
// copy_stream2, while piece k + 1 is already leaving.
//   * host buffers the runtime knows as pinned (blz_host_malloc, hipHostMalloc, hipHostRegister): every copy is a true
//     asynchronous DMA; the pieces are chained with events and the caller waits once, at the end;
//   * pageable host memory: hipMemcpyAsync returns only when the runtime has staged the copy, so the two directions are driven
//     by two host threads (the caller's: device -> host; a helper: host -> device, which waits for a piece's departure
//     through a counter).
static bool host_ptr_is_pinned(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();   // an ordinary malloc'ed pointer: "invalid value"
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

int blz_ntt_exchange(blz_ntt* h, size_t buf, const uint8_t* next_in, size_t in_len, uint8_t* prev_out, size_t out_cap) {
    if (!h || !next_in || !prev_out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    if (buf > 1) return fail(BLZ_ERR_INVALID_PARAM, "buffer index must be 0 or 1");
    const size_t total = ntt_bytes(h);
    if (in_len != total) return fail(BLZ_ERR_INVALID_PARAM, "data length %zu != %zu", in_len, total);
    if (out_cap < total) return fail(BLZ_ERR_INVALID_PARAM, "output buffer too small");
    BLZ_NTT_LIVE(h);
    if (h->in_flight && h->in_flight_buf == (int)buf)
        return fail(BLZ_ERR_INVALID_PARAM, "buffer %zu is being transformed; call wait_result first", buf);
    BLZ_TRY(use_device(h->device));
    char* dbuf = (char*)h->buf[buf].p;
    // Pieces of 64 MiB (1.2 ms of link; smaller ones pay more per-copy overhead than they hide: 16 MiB 93.3 ms per cycle, 64 MiB
    // 91.3, 256 MiB 93.6, 1 GiB 105.6), except at the ends: the host -> device direction can only start once the first piece has
    // left and runs alone behind the last one, so the first and the last pieces ramp 8 / 16 / 32 MiB.
    const size_t piece = (size_t)exp_knob("BLAZE_NTT_XCHG_MB", 64) << 20;
    std::vector<size_t> cut;   // piece k = bytes [cut[k], cut[k + 1])
    {
        const size_t ramp0 = exp_knob("BLAZE_NTT_XCHG_RAMP", 1) != 0 ? (size_t)8 << 20 : piece;
        std::vector<size_t> head, tail;
        size_t lo = 0, hi = total;
        for (size_t r = ramp0; r < piece && hi - lo > 4 * piece; r <<= 1) {
            head.push_back(lo);
            lo += r;
            hi -= r;
            tail.push_back(hi);
        }
        cut = head;
        for (size_t o = lo; o < hi; o += piece) cut.push_back(o);
        for (size_t i = tail.size(); i-- > 0;) cut.push_back(tail[i]);
        cut.push_back(total);
    }
    const size_t npieces = cut.size() - 1;
#ifdef BLZ_EXPERIMENT_KNOBS
    // (experiments, profiles/r05_ntt_exchange_variants.txt: a copy KERNEL for one direction instead of the copy engine)
    if (exp_knob("BLAZE_NTT_XCHG_KERNEL", 0) != 0 && host_ptr_is_pinned(next_in) && host_ptr_is_pinned(prev_out)) {
        const int mode = exp_knob("BLAZE_NTT_XCHG_KERNEL", 0);   // 1: host -> device by kernel, 2: device -> host by kernel, 3: both
        const int blocks = exp_knob("BLAZE_NTT_XCHG_BLOCKS", 64);
        for (size_t k = 0; k < npieces; ++k) {
            const size_t o = cut[k], len = cut[k + 1] - cut[k];
            hipEvent_t ev = h->xchg_ev[k % 4];
            if (mode & 2) hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, h->copy_stream, (uint4*)(prev_out + o), (const uint4*)(dbuf + o), len / 16);
            else BLZ_HIP(hipMemcpyAsync(prev_out + o, dbuf + o, len, hipMemcpyDeviceToHost, h->copy_stream), BLZ_ERR_READ);
            BLZ_HIP(hipEventRecord(ev, h->copy_stream), BLZ_ERR_READ);
            BLZ_HIP(hipStreamWaitEvent(h->copy_stream2, ev, 0), BLZ_ERR_WRITE);
            if (mode & 1) hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, h->copy_stream2, (uint4*)(dbuf + o), (const uint4*)(next_in + o), len / 16);
            else BLZ_HIP(hipMemcpyAsync(dbuf + o, next_in + o, len, hipMemcpyHostToDevice, h->copy_stream2), BLZ_ERR_WRITE);
        }
        BLZ_NTT_WAIT(h, sync_stream_bounded(h->copy_stream, "exchange: copy out of the NTT buffer"));
        BLZ_NTT_WAIT(h, sync_stream_bounded(h->copy_stream2, "exchange: copy into the NTT buffer"));
        return BLZ_OK;
    }
#endif
    if (exp_knob("BLAZE_NTT_XCHG_PINNED", 1) != 0 && host_ptr_is_pinned(next_in) && host_ptr_is_pinned(prev_out)) {
        // Whatever goes wrong while the pieces are being enqueued, BOTH copy streams are drained (bounded) before the call
        // returns: the header lets the caller drop next_in / prev_out on return, and pieces already enqueued keep moving bytes
        // between them and the device until they are through.
        int rc = BLZ_OK;
        std::string err;
        auto step = [&](hipError_t e, int code, const char* what) {
            if (e == hipSuccess) return true;
            (void)hipGetLastError();
            rc = code;
            err = std::string("exchange: ") + what + " failed: " + hipGetErrorString(e);
            return false;
        };
        for (size_t k = 0; k < npieces && rc == BLZ_OK; ++k) {
            const size_t o = cut[k], len = cut[k + 1] - cut[k];
            hipEvent_t ev = h->xchg_ev[k % 4];
            if (!step(hipMemcpyAsync(prev_out + o, dbuf + o, len, hipMemcpyDeviceToHost, h->copy_stream), BLZ_ERR_READ, "device -> host copy")) break;
            if (!step(hipEventRecord(ev, h->copy_stream), BLZ_ERR_READ, "event record")) break;
            if (!step(hipStreamWaitEvent(h->copy_stream2, ev, 0), BLZ_ERR_WRITE, "stream wait")) break;   // (captures this record: the event is free to be re-recorded)
            if (!step(hipMemcpyAsync(dbuf + o, next_in + o, len, hipMemcpyHostToDevice, h->copy_stream2), BLZ_ERR_WRITE, "host -> device copy")) break;
        }
        bool timed_out = false;
        std::string werr;
        int wrc = BLZ_OK;
        for (hipStream_t st : {h->copy_stream, h->copy_stream2}) {
            blz::wait_clear();
            const int r = sync_stream_bounded(st, st == h->copy_stream ? "exchange: copy out of the NTT buffer" : "exchange: copy into the NTT buffer");
            if (r != BLZ_OK) {
                if (blz::wait_timed_out()) timed_out = true;
                if (wrc == BLZ_OK) { wrc = r; werr = blz_last_error_message(); }
            }
        }
        if (timed_out) h->wedged = true;   // (bytes may still be moving: reset / free only)
        if (rc != BLZ_OK) return fail(rc, "%s", err.c_str());
        if (wrc != BLZ_OK) return fail(wrc, "%s", werr.c_str());
        return BLZ_OK;
    }
    std::atomic<size_t> departed{0};     // pieces [0, departed) are in prev_out
    std::atomic<bool> abort_in{false};
    int rc_in = BLZ_OK;
    bool timed_out_in = false;
    std::string err_in;
    const int dev = h->device;
    hipStream_t st_in = h->copy_stream2;
    auto writer_fn = [&, dev, st_in] {
        if (hipSetDevice(dev) != hipSuccess) { rc_in = BLZ_ERR_FILE; err_in = "hipSetDevice failed on the exchange's helper thread"; return; }
        for (size_t k = 0; k < npieces; ++k) {
            while (departed.load(std::memory_order_acquire) <= k) {
                if (abort_in.load(std::memory_order_acquire)) return;
                std::this_thread::yield();
            }
            const size_t o = cut[k], len = cut[k + 1] - cut[k];
            if (hipMemcpyAsync(dbuf + o, next_in + o, len, hipMemcpyHostToDevice, st_in) != hipSuccess) {
                (void)hipGetLastError();
                rc_in = BLZ_ERR_WRITE;
                err_in = "exchange: host -> device copy failed";
                return;
            }
        }
        blz::wait_clear();
        rc_in = sync_stream_bounded(st_in, "exchange: copy into the NTT buffer");
        if (rc_in != BLZ_OK) { err_in = blz_last_error_message(); timed_out_in = blz::wait_timed_out(); }
    };
    std::thread writer;
    try {
        writer = std::thread(writer_fn);
    } catch (...) {
        // no helper thread to be had (the process is at its thread limit): the two-call sequence, one direction after the other -
        // nothing may unwind across the C ABI
        BLZ_LOG(1, "exchange: no helper thread: result and set_data one after the other");
        BLZ_TRY(ntt_result_common(h, buf, prev_out, out_cap, false));
        return ntt_set_data_common(h, buf, next_in, in_len, false);
    }
    int rc_out = BLZ_OK;
    bool timed_out_out = false;
    for (size_t k = 0; k < npieces && rc_out == BLZ_OK; ++k) {
        const size_t o = cut[k], len = cut[k + 1] - cut[k];
        if (hipMemcpyAsync(prev_out + o, dbuf + o, len, hipMemcpyDeviceToHost, h->copy_stream) != hipSuccess) {
            (void)hipGetLastError();
            rc_out = fail_hip(BLZ_ERR_READ, "exchange: device -> host copy failed");
            break;
        }
        // the piece must have LEFT before its place is overwritten: a host-side wait (bounded), which pageable copies have paid already
        blz::wait_clear();
        rc_out = sync_stream_bounded(h->copy_stream, "exchange: copy out of the NTT buffer");
        if (rc_out != BLZ_OK) { timed_out_out = blz::wait_timed_out(); break; }
        departed.store(k + 1, std::memory_order_release);
    }
    if (rc_out != BLZ_OK) abort_in.store(true, std::memory_order_release);
    writer.join();
    if (timed_out_in || timed_out_out) h->wedged = true;
    if (rc_out != BLZ_OK) return rc_out;
    if (rc_in != BLZ_OK) return fail(rc_in, "%s", err_in.c_str());
    return BLZ_OK;
}

// out = {device bytes this handle holds (two transform buffers, scratch, twiddle and factor tables), 1 when pass 2 READS its
// boundary factors from the per-element table (2^27 transforms with memory for it) / 0 when it steps them, 1 when pass 1 reads the
// column-independent boundary table, log_size}
int blz_ntt_info(blz_ntt* h, uint64_t out[4]) {
    if (!h || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    out[0] = (uint64_t)(h->buf[0].cap + h->buf[1].cap + h->scratch.cap + h->tables.cap + h->tables_rr.cap + h->table_b.cap);
    out[1] = h->TR.tB ? 1u : 0u;
    out[2] = h->TR.tA ? 1u : 0u;
    out[3] = (uint64_t)h->logn;
    return BLZ_OK;
}

int blz_ntt_reset(blz_ntt* h) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    BLZ_TRY(use_device(h->device));
    BLZ_TRY(sync_stream_bounded(h->stream, "reset: NTT stream"));
    BLZ_TRY(sync_stream_bounded(h->copy_stream, "reset: NTT copy stream"));
    BLZ_TRY(sync_stream_bounded(h->copy_stream2, "reset: NTT copy stream"));
    h->in_flight = false;
    h->wedged = false;
    return BLZ_OK;
}

int blz_ntt_stream(blz_ntt* h, void** hip_stream, int* device_id) {
    if (!h || !hip_stream) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    if (device_id) *device_id = h->device;
    *hip_stream = (void*)h->stream;
    return BLZ_OK;
}

int blz_ntt_last_kernel_ms(blz_ntt* h, float* out) {
    if (!h || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    *out = h->last_ms;
    return BLZ_OK;
}

static int banks_geometry(blz_ntt* h, uint64_t& n, uint64_t& G, uint64_t& Bg) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    if (h->logn < 10) return fail(BLZ_ERR_INVALID_PARAM, "bank layout needs log_size >= 10 (512-element blocks in pairs)");
    n = 1ull << h->logn;
    G = n >= (1ull << 18) ? n >> 18 : 1;   // the reference shape: 2^27 -> 512 groups of 256 block pairs
    Bg = (n >> 10) / G;
    return BLZ_OK;
}
int blz_ntt_banks_preprocess_device(blz_ntt* h, const void* d_in, void* d_banks) {
    uint64_t n, G, Bg;
    BLZ_TRY(banks_geometry(h, n, G, Bg));
    if (!d_in || !d_banks) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    BLZ_TRY(use_device(h->device));
    hipLaunchKernelGGL(k_ntt_banks_pre, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->copy_stream, (const uint4*)d_in,
                       (uint4*)d_banks, n);
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    BLZ_TRY(sync_stream_bounded(h->copy_stream, "banks preprocess"));
    return BLZ_OK;
}
int blz_ntt_banks_postprocess_device(blz_ntt* h, const void* d_banks, void* d_out) {
    uint64_t n, G, Bg;
    BLZ_TRY(banks_geometry(h, n, G, Bg));
    if (!d_out || !d_banks) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    BLZ_TRY(use_device(h->device));
    hipLaunchKernelGGL(k_ntt_banks_post, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->copy_stream, (const uint4*)d_banks,
                       (uint4*)d_out, n, G, Bg);
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    BLZ_TRY(sync_stream_bounded(h->copy_stream, "banks postprocess"));
    return BLZ_OK;
}

}  // extern "C"
