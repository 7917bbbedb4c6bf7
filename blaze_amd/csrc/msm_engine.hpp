// Host-side interface of the MSM device pipeline (implemented in msm.hip).
#pragma once
#include "common.hpp"

namespace blz {

// Window layout.  Windows may have two widths (cmin and cmin+1 bits, signed digits) plus a top window
// that takes the rest: with W fixed by the scalar width, the bucket count is what the widths can still
// trade (12 windows over 257 bits: 3 x 22 + 8 x 21 + 23 bits hold 18.9 M bucket slots, 12 x 22 hold
// 25.2 M).  Bucket g of window w is boff[w] + (|digit| - 1).  Downstream of the sort the bucket space is
// flat; the bucket reduce walks it as Wv "virtual windows" of V = 2^(cmin-1) buckets each (every
// window's bucket count is a multiple of V) and k_finish stitches them (msm_impl.hip.hpp).
constexpr int MSM_MAX_W = 96;
// entries (points x windows) of one task: u32 indices, and room for the kernels' strided walks to step past the end without wrapping
constexpr uint64_t MSM_MAX_ENTRIES = (1ull << 32) - (1ull << 26);
struct MsmPlan {
    uint32_t npts = 0;   // points in the sum (n * precompute_factor)
    int sbits = 256;     // scalar width per point: 256 (pf=1), 32 (pf=8 chunk) or 64 (pf=8, checked-table plan: two chunks per even base)
    int c = 0;           // widest lower window (reported as window_bits)
    int W = 0;           // windows; their widths sum to >= sbits+1 (signed digits carry into the top one)
    uint8_t width[MSM_MAX_W] = {};     // bits of window w, low to high
    uint32_t boff[MSM_MAX_W + 1] = {}; // first bucket of window w; boff[W] = G
    uint32_t Bw = 0;     // V: buckets per virtual window = 2^(cmin-1)
    int Wv = 0;          // G / V
    uint64_t G = 0;      // bucket slots
    uint32_t L = 0;      // max run length handled by one accumulate unit
    // Window-table tasks (msm_impl.hip.hpp k_build_window_table): the points array holds, per base, its W multiples 2^(c j) P (point-major), so
    // every window's digit d of base i is an addition of +-table[i W + j] into bucket |d| - 1 of ONE bucket set shared
    // by all windows: boff[w] = 0 for every window, G = 2^(c-1), entries carry i W + j.  The reduce sees a single window.
    bool table = false;
    int ebits = 0;       // significant bits the planner assumed for the scalars (bit length of r, or of the range / chunk)
    int base_bit = 0;    // scalar-range tasks (run() bit_lo): the windows start at this bit of the scalar; the result carries 2^base_bit
    double cost = 0;     // the planner's estimate for this plan (ns; host-side comparisons only)
};
MsmPlan make_plan(uint32_t npts, int sbits, int ebits, int force_c);
// window-table geometry for npts bases: the window width c (16..26) whose W = ceil(257 / c) windows make the cheapest task
// (fewer windows = fewer additions, wider windows = more buckets to reduce); BLAZE_MSM_TABLE_C forces c.  0: none fits
// (entries are indexed with 30 bits: npts W < 2^30)
// need_bits: the scalar bits the windows have to cover plus one for the last digit carry: 257, or hi - lo + 1 for a handle
// with a scalar range [lo, hi) (its table holds 2^(lo + c j) P: the range's weight is in the points, too)
int table_window_bits(uint32_t npts, int need_bits = 257);
inline int table_windows(int c, int need_bits = 257) { return (need_bits + c - 1) / c; }
MsmPlan make_table_plan(uint32_t npts, int c, int need_bits = 257);

// Task queue (msm_hw_code.rs:19-25: the device has a task queue and a result queue): up to
// MSM_QUEUE_DEPTH tasks may be in flight.  The throughput-bound part of a task (sort, bucket
// accumulation, first bucket-reduce level) runs on `stream`; its latency-bound tail (upper reduce
// levels, Horner, inversion: a few lanes for ~3 ms) runs on `tail_stream`, so it overlaps the next
// task's sort and accumulation instead of idling the chip.  Everything the tail touches is per slot.
constexpr int MSM_QUEUE_DEPTH = 2;
constexpr int MSM_MAX_SLICES = 64;
struct MsmSlot {
    // 0 start, 1 sort done, 2 accumulate done, 5..6 the accumulate kernel alone (stream);
    // 3 reduce done, 4 finish done (tail_stream)
    hipEvent_t ev[8] = {};
    hipEvent_t ev_l0 = nullptr;    // stream: level-0 reduce written -> tail may start (and this slot's sort outputs are free)
    hipEvent_t ev_sorted = nullptr;  // the sort stage (sort, scans, unit lists) of this slot's task is complete
    hipEvent_t ev_s0 = nullptr, ev_s1 = nullptr;  // timing of a hidden sort stage on sort_stream
    bool sort_hidden = false;      // this task's sort ran on sort_stream
    bool l0_recorded = false;      // ev_l0 has been recorded at least once
    hipEvent_t ev_done = nullptr;  // tail_stream: result bytes are in result_h
    DevBuf lvlA[2], lvlC[2];
    uint8_t* result_h = nullptr;   // pinned result bytes
    uint32_t* stats_h = nullptr;   // pinned: [0] total units, [1] max bucket count, [2] total entries (read in finish())
    uint64_t max_units = 0;        // the bound the launches of this task were sized by
    MsmPlan plan;
    // piecewise tasks (msm.hip begin()): one event pair per piece around its k_accumulate_cont launch; finish() sums them
    hipEvent_t slice_ev[2 * 64] = {};
    int slices = 1;
    bool accum_timed = false;
    bool busy = false;             // enqueued, result not collected yet
    // state of a task between begin() and end() (msm.hip)
    bool open = false;             // begun, not every piece enqueued yet
    bool phased = false;           // the caller enqueues the pieces as their data lands (inputs_event is recorded by end())
    bool use_s3 = false, ranged = false;
    uint32_t npts = 0, pts_per_slice = 0;
    int sbits = 0, bit_lo = 0, bit_hi = 0;
    hipEvent_t task_inputs_event = nullptr;   // the caller's inputs_event, until it has been recorded
    // A piecewise task on an otherwise idle handle sorts piece k + 1 UNDERNEATH the accumulation of piece k, like a stream
    // of tasks does: the pieces alternate between this slot's SortBufs and the other slot's (which nobody uses while that
    // slot is idle), sorts on sort_stream, one pair of events per buffer set.
    bool pingpong = false;
    hipEvent_t ev_sorted_pp[2] = {nullptr, nullptr};   // piece's sort stage complete (index: piece & 1)
    hipEvent_t ev_acc_pp[2] = {nullptr, nullptr};      // the accumulation that read the piece's sort outputs is through
    bool acc_pp_recorded[2] = {false, false};
};

struct MsmEngine {
    int device = 0;
    int curve = 0;
    // Layout / arithmetic of the Montgomery point copy: 0 = the curve's default; 1 = BN254's 32-bit-limb twin
    // (Fq_BN254_W32, msm_bn254w.hip).  BN254's 9 x 29 reduced radix wins where runs are short (pf = 1: the first
    // addition of a run is the cheap affine + affine one, the level-0 reduce is cheaper: 2^26 in 69.7 instead of 76.9
    // ms) and loses on the precompute shapes (pf = 8: 2^29 points, 32 GiB of bases, runs of 8192: 100.5 against 95.3 ms:
    // faster arithmetic only exposes the translation-bound gathers - profiles/r03_bn254_*.txt), so the handle picks by
    // precompute factor.  format_id() keys the arena's Montgomery shadows.
    int repr = 0;
    int format_id() const { return curve | (repr << 8); }
    hipStream_t stream = nullptr, tail_stream = nullptr, aux_stream = nullptr;  // aux: combine_partials
    MsmSlot slots[MSM_QUEUE_DEPTH];
    int cur = 0;                   // slot of the task being enqueued
    // What the digit sort of a task hands to its accumulation / reduce: one set per task slot, so that the sort of task
    // k + 1 can run (on sort_stream, underneath task k's accumulation: msm_sort3.hip) while task k still reads its own.
    struct SortBufs {
        DevBuf count, off, unit_off, unit_bucket, unit_order, lenhist, entries, stats;
        DevBuf range_scalars;   // scalar-range tasks: words [bit_lo / 32, bit_hi / 32) of every scalar, zero-extended to 32 bytes
    };
    SortBufs sbuf[MSM_QUEUE_DEPTH];
    int sb_sel = 0;                // the set the piece being enqueued uses: its slot's, or (ping-pong pieces) the other slot's
    SortBufs& sb() { return sbuf[sb_sel]; }
    hipStream_t sort_stream = nullptr;   // hidden sorts
    hipStream_t sort_st = nullptr;       // the stream the CURRENT task's sort stage is being enqueued on (stream or sort_stream)
    hipEvent_t last_sort_done = nullptr; // sorts share their scratch (coarse, inter, inter2, ...): each waits for the one before
    DevBuf coarse, inter, inter2, slice_map, partial, blocksums, result, sort3_tabs;
    DevBuf bucket_sums, bucket_ident;   // piecewise tasks: the bucket sums carried across pieces; identity unit_off for the reduce
    // A task in four steps (msm.hip): begin() plans it and takes a slot; per piece sort_slice() (needs the piece's scalars)
    // and accumulate_slice() (needs its points); end() enqueues the bucket reduce and the tail.  msm_stage.hip uses the steps
    // for host buffers (DMA mode): a piece is handed to the device as soon as it has crossed the PCIe link.  `pieces` > 1:
    // the pieces share one bucket space, the sums are carried from piece to piece (k_accumulate_cont).
    int begin(uint32_t npts, int sbits, int* slot, int table_c, int bit_lo, int bit_hi, int pieces, bool phased);
    int sort_slice(int slot, int piece, const void* d_scalars, uint32_t np);
    int accumulate_slice(int slot, int piece, const void* d_points_mont);
    int end(int slot);
    void abandon(int slot);   // give up a task between begin() and end() (a copy failed)
    hipEvent_t inputs_event = nullptr;   // set by the caller of run() / begin(): recorded once the task has read
                                         // its scalars / raw points
    uint8_t* combine_h = nullptr;  // pinned bytes of combine_partials
    MsmPlan last_plan;
    float last_ms[8] = {};
    bool last_sort_hidden = false;   // of the last task collected by finish(): its sort stage ran on sort_stream
    bool recent_hot[2] = {false, false};   // the last two collected tasks piled entries into a few buckets (begin()'s guard)
    uint32_t sort_slices = 1, sort_nc = 0;  // geometry of the last LDS sort (msm_sort.hip)
    int sort_cl = 0;
    void* sort_inter_fine = nullptr;  // u16 fine digits of the sort intermediate (second half of `inter`)

    int init(int device_id, int curve_id, int precompute_factor);
    bool destroy();   // false: the streams never drained (wedged device work): everything was leaked instead of freed
    // device result bytes of slot s / scratch of combine_partials inside `result`
    uint32_t* slot_result(int s) { return result.as<uint32_t>() + (size_t)s * 64; }
    bool can_accept() const;
    MsmPlan plan_for(uint32_t npts, int sbits) const;   // the plan run() will use (host-side only)
    // raw wire-format points (x||y canonical LE) -> Montgomery AoS at mont_point_bytes() stride (never in place)
    int points_to_mont(const void* d_raw, void* d_mont, uint32_t npts);
    int points_to_mont_even(const void* d_raw, void* d_mont, uint32_t nq);
    int check_precompute(const void* d_raw, uint64_t nelem, uint32_t* flag, hipStream_t st);
    // window table of npts wire-format points (msm_impl.hip.hpp k_build_window_table): see MsmCurveOps::build_table
    int build_table(const void* d_raw, void* d_table, uint32_t npts, int c, int W, int base_shift, void* scratch, uint32_t* flag,
                    hipStream_t st);
    size_t table_scratch_bytes(int W) const;
    // enqueue the whole pipeline; *slot identifies the task for finish().  Fails when both slots are busy.
    // table_c > 0: d_points_mont is the window table of the npts bases (table_windows(table_c) entries per base)
    // bit_hi > bit_lo (multiples of 32, pf = 1, no table): only bits [bit_lo, bit_hi) of every scalar take part and the result
    // is 2^bit_lo x their sum - one shard of a job split by scalar chunk (blz_msm_set_scalar_range)
    int run(const void* d_points_mont, const void* d_scalars, uint32_t npts, int sbits, int* slot, int table_c = 0, int bit_lo = 0,
            int bit_hi = 0);
    MsmPlan plan_for_range(uint32_t npts, int bit_lo, int bit_hi) const;
    // wait for task `slot`, copy the result out (result_size bytes), collect its phase timings
    int finish(int slot, uint8_t* out);
    // add `count` partial results (host bytes, or device bytes already ordered on aux_stream) on the device,
    // normalised output
    int combine_partials(const uint8_t* partials, size_t count, uint8_t* out, bool on_device);
    int sync_all();
};

// arena diet: conversions keyed by the FORMAT of a Montgomery copy (MsmEngine::format_id(): curve | repr << 8), for callers
// without an engine of that format at hand (arena.hip)
int msm_points_from_mont(int format_id, const void* d_mont, void* d_raw, uint64_t npts, hipStream_t st);
int msm_points_all_canonical(int format_id, const void* d_raw, uint64_t npts, uint32_t* flag, hipStream_t st);
// A few result / statistics words from device memory into PINNED host memory, stored by a one-wave kernel instead of a
// device -> host hipMemcpyAsync: the copy engines (SDMA) are shared with every large transfer on the device - the next task's
// 2 GiB upload, another process's traffic, the driver's own wipes of freed memory - and a 144-byte read-back queued behind
// one of those holds a finished task's result for as long as the transfer takes (seen: 0.7 s in bench.py's config 4 leg behind
// the release of 75 GiB).  dwords <= 64.
int copy_words_to_pinned(void* host_pinned, const void* d_src, uint32_t dwords, hipStream_t st);
size_t fq_bytes(int curve);
size_t mont_point_bytes(int curve);  // stride of the Montgomery point array the pipeline reads (msm_impl.hip.hpp MONT_STRIDE)
// two-level LDS-privatised digit sort (msm_sort.hip): fills count[], then (after the scan) entries[]
int msm_sort_lds(MsmEngine& E, const void* d_scalars, uint32_t npts, int sbits);
// three-level, small-footprint digit sort built to run underneath another task's k_accumulate (msm_sort3.hip): fills
// count[] and entries[] of the current slot's SortBufs on E.sort_st; msm_sort3_ok: does the plan qualify
bool msm_sort3_ok(const MsmPlan& P, int sbits);
int msm_sort3_max_vgprs();   // the largest register count among the three-level sort's kernels (0 if unknown)
int msm_sort3(MsmEngine& E, const void* d_scalars, uint32_t npts, int sbits);
// the same sort for window-table plans (shared bucket set, entries = point * W + window)
bool msm_sort3t_ok(const MsmPlan& P);
int msm_sort3t_max_vgprs();
int msm_sort3t(MsmEngine& E, const void* d_scalars, uint32_t npts);
int msm_sort_lds_scatter(MsmEngine& E);
// the whole sort stage of a small task (digits, bucket scan, entries, unit lists, stats) in one block (msm_sort_tiny.hip)
bool msm_sort_tiny_ok(const MsmPlan& P, uint32_t npts, int sbits);
int msm_sort_tiny(MsmEngine& E, const void* d_scalars, uint32_t npts, int sbits, uint32_t max_units);
int launch_fill_units(MsmEngine& E, uint32_t units);  // unit->bucket map + length-ordered unit list

// per-curve entry points (one translation unit per curve: msm_<curve>.hip)
struct MsmCurveOps {
    int (*points_to_mont)(MsmEngine&, const void* d_raw, void* d_mont, uint32_t npts);
    int (*emit_infinity)(MsmEngine&);
    // phase 1 after a digit sort: unit lists, k_accumulate, k_combine_units (at most max_units units; the real count is on
    // the device).  slice >= 0: piece of a piecewise task - k_accumulate_cont, bracketed by the piece's events.
    int (*run_accumulate)(MsmEngine&, const void* d_pts, uint32_t max_units, int slice);
    // piecewise tasks: bucket_sums[g] += the piece's sum of bucket g where its run needed several units
    int (*merge_buckets)(MsmEngine&);
    // phases 2 - 3 over bucket sums found at sums[unit_off[g]] (unit_off[g + 1] > unit_off[g], else the bucket is empty)
    int (*run_reduce)(MsmEngine&, const void* sums, const void* unit_off);
    int partial_dwords;   // dwords of one unit / bucket sum in `partial`
    // VGPRs of k_accumulate as compiled (hipFuncGetAttributes): what the hidden sort has to fit beside
    int (*accumulate_vgprs)();
    // window table of npts wire-format points (msm_impl.hip.hpp k_build_window_table): table[i W + j] = 2^(base_shift + c j) P_i in the
    // Montgomery point format, on `st`; scratch: table_scratch_bytes(W) bytes; *flag (device u32) is set when a multiple came
    // out as infinity
    int (*build_table)(MsmEngine&, const void* d_raw, void* d_table, uint32_t npts, int c, int W, int base_shift, void* scratch,
                       uint32_t* flag, hipStream_t st);
    size_t (*table_scratch_bytes)(int W);
    int (*combine)(MsmEngine&, const uint8_t* partials, size_t count, uint8_t* out, bool on_device);
    // checked-table plan of precompute handles (msm_impl.hip.hpp k_check_precompute / k_points_to_mont_even):
    // Montgomery copy of the even bases of nq / 4 elements (raw: 8 wire-format points per element; nq = 4 per element);
    // *flag (device u32) is raised when the table of nelem elements is not B_j = 2^32 B_(j-1) over on-curve B_0
    int (*points_to_mont_even)(MsmEngine&, const void* d_raw, void* d_mont, uint32_t nq);
    int (*check_precompute)(MsmEngine&, const void* d_raw, uint64_t nelem, uint32_t* flag, hipStream_t st);
    // arena diet (arena.hip): the Montgomery copy back to wire-format points; *flag raised when a raw coordinate is >= q
    int (*points_from_mont)(const void* d_mont, void* d_raw, uint64_t npts, hipStream_t st);
    int (*points_all_canonical)(const void* d_raw, uint64_t npts, uint32_t* flag, hipStream_t st);
};
const MsmCurveOps& msm_ops_bls377();
const MsmCurveOps& msm_ops_bls381();
const MsmCurveOps& msm_ops_bn254();
const MsmCurveOps& msm_ops_bn254_w32();

}  // namespace blz
