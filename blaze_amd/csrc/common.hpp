// Shared host-side plumbing of libblaze_hip: error convention, HIP checks, device arena.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/blaze_hip.h"
#include "../../include/blaze_hip_aux.h"

namespace blz {

void set_last_error(const char* fmt, ...);
int log_level();  // BLAZE_LOG=0..3 (env), the RUST_LOG analogue of README.md:150

#define BLZ_LOG(lvl, ...)                                  \
    do {                                                   \
        if (::blz::log_level() >= (lvl)) {                 \
            fprintf(stderr, "[blaze_hip] " __VA_ARGS__);   \
            fputc('\n', stderr);                           \
        }                                                  \
    } while (0)

// HIP call -> error code of the given class on failure.  A failed runtime call (an allocation on a full device, say) also leaves
// its code behind as the thread's sticky "last error", where the next launch check - BLZ_HIP(hipGetLastError()) behind some kernel
// of some later, healthy call - would find it and fail for no reason: it is wiped HERE, where the runtime call failed and is
// reported (and in fail_hip below) - not in the generic reporter, which argument checks reach on threads that never touched the GPU.
#define BLZ_HIP(call, errcode)                                                                     \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            ::blz::set_last_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                                  __LINE__);                                                       \
            (void)hipGetLastError();                                                               \
            return (errcode);                                                                      \
        }                                                                                          \
    } while (0)

#define BLZ_TRY(expr)              \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != BLZ_OK) return rc__; \
    } while (0)

inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    set_last_error("%s", buf);
    return code;
}
// the report of a HIP runtime call that failed: fail() + the runtime's sticky error wiped (see BLZ_HIP)
inline int fail_hip(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    set_last_error("%s", buf);
    (void)hipGetLastError();
    return code;
}

// Runtime switches.  The shipped library reads EIGHT environment variables (README.md lists them; tests/test_host_logic.py
// compares that table with what the sources read): BLAZE_LOG, BLAZE_WAIT_TIMEOUT_MS, BLAZE_COMM_TIMEOUT_MS, BLAZE_SHARD,
// BLAZE_SORT_HIDE, BLAZE_MSM_PIECES, BLAZE_MSM_PLAN (+ BLAZE_HIP_LIB in the Python mirror).  Everything else that was ever
// swept - sort geometry, segment sizes, kernel variants that lost their A/B runs - is a compile-time experiment: a build
// with -DBLZ_EXPERIMENT_KNOBS (tools/ only) reads the same names from the environment, the shipped build folds the defaults.
int env_int(const char* name, int dflt);
#ifdef BLZ_EXPERIMENT_KNOBS
inline int exp_knob(const char* name, int dflt) { return env_int(name, dflt); }
#else
inline int exp_knob(const char*, int dflt) { return dflt; }
#endif
// BLAZE_MSM_PLAN="c=13,L=64,split_ns=0,table_c=26": overrides of the window planner for tests and sweeps (c: uniform window
// width; L: unit length; split_ns: the planner's price of a split top window - 0 lets mixed widths appear at sizes the
// oracle can check; table_c: window width of window tables).  Absent keys keep the planner's own choice.
int plan_override(const char* key, int dflt);

// number of usable devices, 0 if the runtime cannot see any
int device_count();
// make `device_id` current; BLZ_ERR_FILE if it does not exist (the reference unwraps the open()
// of /dev/xdma{id}_*: src/utils.rs:74)
int use_device(int device_id);

// Bounded waits ("failure detection": the reference's wait_result spins on a status register for ever,
// msm_api.rs:222-238 / ntt_api.rs:89-108).  Every host-side wait for device work polls against a deadline of
// BLAZE_WAIT_TIMEOUT_MS (default 120 000; read at each wait) and returns BLZ_ERR_UNKNOWN with a message when it
// expires; wait_timed_out() then tells the caller that the failure was the deadline, not a HIP error, so the handle
// can refuse further work until reset.
int wait_timeout_ms();
int sync_event_bounded(hipEvent_t ev, const char* what);
int sync_stream_bounded(hipStream_t st, const char* what);
int sync_device_bounded(const char* what);   // every stream of the current device (before buffers other handles' tasks may read are freed)
bool wait_timed_out();   // the last sync_*_bounded of this thread ended on its deadline
void wait_clear();       // forget it (before a call that may fail without ever reaching a wait)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (current device, kernel): the opt-in for more than
// 64 KiB of dynamic LDS is per device, and a host may open clients on several devices of one process.
int ensure_dynamic_lds(const void* kernel, int bytes);

// growable device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // growth frees the old allocation behind a BOUNDED drain of the device (common.hip); exact: no slack for later growth
    int reserve(size_t bytes, bool exact = false);
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// Per-device, process-global byte-addressed arena: the card's "HBM" of load_data_to_hbm / hbm_point_addr
// (src/ingo_msm/msm_api.rs:299-322), with the card's semantics: flat memory.  Writes that touch or overlap
// earlier ones extend the same extent (bytes outside the written range are kept), so a 48 GiB table loaded in
// pieces is one extent; only a gap that was never written separates extents, and reads / MSM tasks must not
// cross one.  Each extent may carry a Montgomery-form shadow of its points for one (curve, phase) - phase =
// byte offset of the point grid inside the extent - built lazily by the MSM engine; a write dirties only the
// byte span it touched, and the next task converts only the points of that span.  arena.hip.
struct ArenaExtent {
    uint64_t start = 0;
    size_t len = 0, cap = 0;     // bytes written / allocated (appends within cap are in place)
    void* raw = nullptr;
    bool imported = false;       // opened from another process's export (hipIpcOpenMemHandle): fixed size, read-only
    bool exported = false;       // other processes map this allocation (blz_arena_export): frozen - it must neither
                                 // move (grow) nor change under their private Montgomery shadows
    // shadow
    void* mont = nullptr;
    size_t mont_bytes = 0;
    int mont_curve = -1;
    uint32_t mont_phase = 0;
    uint64_t dirty_lo = 0, dirty_hi = 0;   // byte span (relative to start) whose points are stale in the shadow
    hipEvent_t shadow_ready = nullptr;     // recorded after the last conversion; consumers on other streams wait
    bool shadow_recorded = false;          // shadow_ready has been recorded at least once since the shadow was (re)built
    // window tables (msm_impl.hip.hpp k_build_window_table; opt-in per handle): for the points [first, +npts) of the grid at `phase`
    // the W multiples 2^(lo + c j) P, j < W, in the shadow's point format, point-major.  One per (bases, scalar range) that
    // a handle asked for - the ranks of a job sharded by scalar chunk each tabulate their own range - at most
    // MAX_TABLES per extent.  A write of up to TABLE_PATCH_MAX_POINTS bases has their rows re-tabulated by the next task; a
    // larger one (or one that moves the extent) drops the tables.
    struct WindowTable {
        void* p = nullptr;
        size_t bytes = 0;
        int format = -1, c = 0, W = 0;
        int lo = 0, hi = 256;              // scalar range the table serves: entry j = 2^(lo + c j) P
        uint32_t phase = 0;
        uint64_t first = 0, npts = 0;
        float build_ms = 0;
    };
    static constexpr size_t MAX_TABLES = 32;
    static constexpr uint64_t TABLE_PATCH_MAX_POINTS = 1u << 18;
    std::vector<WindowTable> tables;
    uint64_t tab_dirty_lo = 0, tab_dirty_hi = 0;   // bytes (relative to start) rewritten since the tables were built: their rows are
                                           // re-tabulated by the next task that asks for a table (arena_tables.hip arena_points_table)
    bool table_refused = false;            // a build failed (a base of even order, or no memory): no NEW build until the next write
    // Checked-table plan of precompute handles (arena_tables.hip arena_precompute_check; opt-in per handle): has the caller's x8 table
    // been compared, element by element, with what precompute_base_* produces (tests/msm/mod.rs:360-380: B_j = 2^32 B_(j-1), B_0
    // on the curve)?  For the points [first, +npts) of the grid at `phase`; a write re-arms the check (state 3: for the elements it touched).
    struct PrecompCheck {
        int state = 0;                     // 0 not checked, 1 consistent, 2 refuted, 3 consistent but for the elements a later write
                                           // touched (bytes [redo_lo, redo_hi) of the extent): only those are checked again
        int curve = -1;
        uint32_t phase = 0;
        uint64_t first = 0, npts = 0;
        uint64_t redo_lo = 0, redo_hi = 0;
        float ms = 0;                      // device time of the (last) check
        uint64_t gen = 0;                  // bumped by every commit of a check: one that ran unlocked commits only onto the record it read
    } pcheck;
    // Arena diet (blz_arena_set_policy, arena.hip): 0 raw bytes in place; 1 their canonical check is enqueued (diet_ev, diet_flag);
    // 2 raw DROPPED - the complete Montgomery copy is the only copy, and get_data_from_hbm / writes / exports / table builds
    // convert back from it; 3 refused until the next write (a coordinate >= q would not survive the round trip)
    int diet = 0;
    hipEvent_t diet_ev = nullptr;
    uint32_t* diet_flag = nullptr;
    uint64_t epoch = 0;                    // changes with every write into the extent (a check that ran unlocked commits only to the bytes it read)
    // A table being built (arena_tables.hip arena_points_table): in chunks, paced by the tasks over these bases (each enqueues a
    // few chunks on its own stream ahead of itself and takes the plain path); the task that finds `done` complete behind the
    // last chunk adopts the table.
    struct TableBuild {
        void* tab = nullptr;
        uint32_t* flag = nullptr;          // a word of the arena's build_flags, owned by the build: set when a multiple came out as infinity
        size_t bytes = 0;
        hipEvent_t done = nullptr, t0 = nullptr;
        int format = -1, c = 0, W = 0, lo = 0, hi = 256;
        uint32_t phase = 0;
        uint64_t first = 0, npts = 0;
        uint64_t next_chunk = 0;           // chunks [0, next_chunk) have been enqueued
        bool recorded = false;             // `done` has been recorded behind the last enqueued chunk
    } build;
};
struct Arena {
    std::mutex mu;
    std::vector<ArenaExtent> ext;
    void* build_scratch = nullptr;         // the builds' lane-private rows (chunks are chained through the build's event)
    size_t build_scratch_bytes = 0;
    hipEvent_t scratch_event = nullptr;    // recorded behind the last chunk that used the rows
    bool scratch_recorded = false;
    // 256 device flag words raised by kernels (a base of even order in a table build, a refuted table check, a non-canonical
    // coordinate).  A word is OWNED from arena_flag_acquire to arena_flag_release - by an extent's diet check for as long as the
    // extent lives, by a table build until it is adopted or dropped, by a check or a patch for the call - so a flag that is read
    // long after it was armed (a diet check parked until the next task touches the extent) is still the reader's.
    uint32_t* build_flags = nullptr;
    std::vector<uint16_t> flag_free;       // the words nobody owns (filled when build_flags is allocated)
    int policy = 0;                        // blz_arena_set_policy: bit 0 = drop the raw bytes of an extent once its Montgomery copy is complete
};
Arena& arena_for(int device_id);
// find extent containing [pos, pos+len); nullptr if none
ArenaExtent* arena_find(Arena& a, uint64_t pos, size_t len);
// write bytes (host or device source) at pos; extends / merges extents as needed; blocking
int arena_write(int device_id, uint64_t pos, const void* src, size_t len, bool src_is_device, hipStream_t st);
void arena_free_extent(Arena& a, ArenaExtent& x);
uint32_t* arena_flag_acquire(Arena& a);             // nullptr: no memory for the words, or all of them owned
void arena_flag_release(Arena& a, uint32_t*& f);    // (null-safe; nulls f)
uint64_t arena_next_epoch();
// arena diet: give a dieted extent its raw bytes back (converted from the Montgomery copy, on st; blocking, bounded); no-op otherwise
int arena_restore_raw(Arena& a, ArenaExtent& e, hipStream_t st);
// ... and, after a conversion pass of arena_points_mont, move the extent one step along check -> drop (caller holds the lock)
int arena_diet_step(Arena& a, ArenaExtent& e, size_t point_bytes, hipStream_t st);
// bytes [off, off + len) of an extent (relative to its start) into host memory, whichever copy holds them
int arena_read_bytes(Arena& a, ArenaExtent& e, uint64_t off, size_t len, void* out, hipStream_t st);
void arena_drop_table(Arena& a, ArenaExtent& x);   // the tables and a build in flight; the caller has drained the device
void arena_drop_build(Arena& a, ArenaExtent& x);   // a build in flight only

}  // namespace blz
