// The MSM handle behind the C ABI (include/blaze_hip.h blz_msm) and what the translation units that implement it share:
// msm_capi.hip (the DriverPrimitive entry points), msm_stage.hip (set_data: staging, pieces, streamed tasks),
// arena_tables.hip (Montgomery copies, window tables and table checks of arena extents), shard_layout.hip (host-side
// pricing of multi-GPU layouts), msm_comm.hip (RCCL exchange).
#pragma once
#include <deque>

#include "msm_engine.hpp"
#include "rccl_dyn.hpp"

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
    blz::DevBuf scalars_buf[2], points_raw[2], points_mont;
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
    // Tasks whose points arrive over the link and are consumed piece by piece (DMA mode: one-call and streamed tasks): a piece's raw
    // points and their Montgomery copy live in a slot of a small ring, not in buffers of the whole task's size - 4 x (96 + 128) bytes
    // x the piece instead of 48 + 64 GiB for the reference's largest shape (whose allocation alone took the first task 4 s).  A
    // slot's raw bytes may be overwritten once the to-Montgomery pass that read them is through (raw_read, recorded on the main
    // stream; the copy stream waits for it); its Montgomery copy is rewritten by a later piece's pass on the same main stream,
    // behind the accumulation that gathered from it.
    struct PieceRing {
        static constexpr int SLOTS = 4;
        blz::DevBuf raw, mont;
        uint64_t slot_pts = 0;                // points a slot holds (grows to the largest piece seen)
        hipEvent_t raw_read[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
        bool recorded[SLOTS] = {false, false, false, false};
        uint64_t next = 0;                    // pieces handed out so far: piece -> slot next % SLOTS
    } ring;
    // A task fed by SEVERAL set_data calls (msm_stage.hip stage_stream): the card takes a task's scalars and points through FIFOs and
    // counts elements against NUMBER_OF_MSM_ELEMENTS (msm_api.rs:155-202, msm_hw_code.rs:18-19), so any split of an armed task's
    // bytes over calls is the same task.  SURVEY.md 8(b): {armed_n, received}, launch when received == armed_n.
    struct Stream {
        bool open = false;
        int mode = 0;              // 1 scalars only, bases in the arena; 2 points + scalars (DMA mode); 3 points into the arena + scalars
        bool src_device = false;   // the slices are device pointers (set_data_device)
        uint32_t total = 0;        // elements of the armed task (initialize's nof_elements)
        uint32_t received = 0;     // elements delivered so far
        int set = -1;              // staging set the slices land in
        int slot = -1;             // engine slot of a task enqueued piece by piece; -1: launched whole when the last slice is in
        uint32_t per = 0;          // points per piece
        int pieces = 0, next_piece = 0;
        uint32_t ppe = 1;          // points per element as the engine counts them (1; 8; 4 on the checked-table plan)
        uint32_t npts = 0, done_pts = 0;   // points of the task / handed to the engine so far
        uint64_t ring_first = 0;   // mode 2 in pieces: ring piece number of the task's piece 0
        int sbits = 0;
        bool even = false;         // checked-table plan: pieces gather from the even-base copy
        uint64_t arena_pos = 0;    // modes 1 and 3: where the task's bases start
    } strm;
    blz::MsmEngine eng;
    // a wait ran into its deadline (BLAZE_WAIT_TIMEOUT_MS): device work of this handle may never complete, so nothing
    // new is queued behind it; reset (which waits, bounded, for the streams to drain) or free are the ways out
    bool wedged = false;
    // multi-GPU exchange (blz_msm_comm_*): one communicator rank per handle
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_size = 0;
    blz::DevBuf comm_buf;   // [send: one partial | recv: comm_size partials]
    // resident-base window table (blz_msm_set_window_table; off for new handles)
    int window_table = 0;   // 0 off, 1 where it pays (the BLS curves), 2 always
    // scalar range of this handle's tasks (blz_msm_set_scalar_range): bits [range_lo, range_hi) of every scalar; 0, 0 = all
    int range_lo = 0, range_hi = 0;
    uint64_t table_info[4] = {0, 0, 0, 0};   // of the last HBM task: table bytes, window bits, windows, build time (us)
    // checked-table plan of a precompute handle (blz_msm_set_precompute_plan; off for new handles)
    int precompute_plan = 0;
    uint64_t pc_info[4] = {0, 0, 0, 0};      // of the last HBM task: took the plan, check state of its bases, check time (us), bytes of the even-base copy
};

namespace blz {

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

inline size_t point_size(const blz_msm* h) { return blz_point_size(h->curve); }
inline size_t result_size(const blz_msm* h) { return blz_result_size(h->curve); }

// ---- arena_tables.hip
constexpr int TABLE_CHUNKS_PER_TASK = 4;           // ~22 ms on top of a 2^26 task's 117: 86 tasks until a 2^26 table is there
bool wants_table_mode(const blz_msm* h);
bool wants_table(const blz_msm* h);
int plan_repr_bn254(uint64_t nelem);
void task_repr_bn254pc(blz_msm* h, bool on_plan, uint64_t checked_elems);   // the arithmetic of a BN254 precompute handle's next task
int arena_points_mont(blz_msm* h, uint64_t pos, uint32_t npts, const void** out, bool even = false);
int arena_precompute_check(blz_msm* h, uint64_t pos, uint32_t nelem, bool* ok, uint64_t* checked_elems = nullptr);
int arena_points_table(blz_msm* h, uint64_t pos, uint32_t npts, const void** out, int* c_out, int chunk_budget);
int resolve_arena_task(blz_msm* h, uint64_t pos, uint32_t n, bool allow_table, bool allow_plan, uint32_t* npts, int* sbits, int* table_c);
// ---- msm_stage.hip
int launch_if_ready(blz_msm* h);
int stage_stream(blz_msm* h, bool have_points, const void* points, size_t points_len, const void* scalars, size_t scalars_len, uint32_t m,
                 int has_hbm, uint64_t hbm_addr, uint64_t hbm_off, bool on_device);
int ring_reserve(blz_msm* h, uint32_t piece_pts);   // the ring's slots hold pieces of piece_pts points
void stream_abandon(blz_msm* h);   // give up a half-fed task (reset, a failed slice)
int stage_common(blz_msm* h, bool have_points, const void* points, size_t points_len, const void* scalars, size_t scalars_len, uint32_t n,
                 int has_hbm, uint64_t hbm_addr, uint64_t hbm_off, bool on_device);

}  // namespace blz
