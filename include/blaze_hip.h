/* libblaze_hip - MI355X (gfx950) device path for blaze's MSM / NTT primitives.
 *
 * C ABI cut at the DriverPrimitive method level: one function per trait method per primitive
 * (reference trait: src/driver_client/dclient.rs:28-46).  Each entry cites the reference
 * interface it replaces.  Plain pointers and sizes only; the callee borrows host pointers for the
 * duration of the call; the caller provides output buffers.  A handle is not thread-safe; distinct
 * handles are independent (own stream and workspace).  Nothing here ever falls back to a CPU path:
 * if no HIP device is usable every constructor fails with BLZ_ERR_FILE.
 *
 * Return value of every int function: 0 = Ok, otherwise a DriverClientError code in the order of
 * the reference enum (src/error.rs:6-32); blz_last_error_message() gives the detail string
 * (the `offset` / `path` payload of the reference variants).
 */
#ifndef BLAZE_HIP_H
#define BLAZE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/error.rs:6-32, same order */
enum blz_error {
    BLZ_OK = 0,
    BLZ_ERR_WRITE = 1,               /* WriteError{offset,source}: host->device transfer failed      */
    BLZ_ERR_READ = 2,                /* ReadError{offset,source}: device->host transfer failed       */
    BLZ_ERR_HBICAP_NOT_READY = 3,    /* HBICAPNotReady: never produced (no bitstream to load)        */
    BLZ_ERR_INVALID_PARAM = 4,       /* InvalidPrimitiveParam: bad mode combination / sizes / state  */
    BLZ_ERR_CSV = 5,                 /* CsvError: never produced                                     */
    BLZ_ERR_LOAD_FAILED = 6,         /* LoadFailed{path}: librccl.so.1 could not be loaded (comm API) */
    BLZ_ERR_FILE = 7,                /* FileError: device could not be opened (no GPU / bad ordinal) */
    BLZ_ERR_UNKNOWN = 8              /* Unknown: kernel launch / runtime failure                     */
};

/* src/ingo_msm/msm_cfg.rs:4-8 and :11-14, declaration order */
enum blz_curve { BLZ_BLS377 = 0, BLZ_BLS381 = 1, BLZ_BN254 = 2 };
enum blz_mem { BLZ_HBM = 0, BLZ_DMA = 1 };

#define BLZ_PRECOMPUTE_FACTOR_BASE 1u /* src/ingo_msm/msm_api.rs:39 */
#define BLZ_PRECOMPUTE_FACTOR 8u      /* src/ingo_msm/msm_api.rs:40 */
#define BLZ_SCALAR_SIZE 32u           /* src/ingo_msm/msm_cfg.rs:48 */

typedef struct blz_msm blz_msm;
typedef struct blz_ntt blz_ntt;

const char* blz_last_error_message(void);
/* number of usable HIP devices (0 when there is none); never fails */
int blz_device_count(void);
/* sizes per curve: src/ingo_msm/msm_cfg.rs:44-92 (point 96/64, result 144/96) */
size_t blz_point_size(int curve);
size_t blz_result_size(int curve);

/* ------------------------------------------------------------------ MSM (src/ingo_msm/msm_api.rs) */

/* DriverClient::new(id, cfg) (dclient.rs:79-86) + MSMClient::new(MSMInit{mem_type,is_precompute,curve})
 * (msm_api.rs:44-55).  (BN254, HBM) is todo!() in the reference (msm_cfg.rs:38); here it is defined
 * by analogy (point 64 B, result 96 B). */
int blz_msm_new(int device_id, int mem_type, int is_precompute, int curve, blz_msm** out);
void blz_msm_free(blz_msm* h);

/* MSMClient::loaded_binary_parameters (msm_api.rs:57-70): [image_id, image_parameters].
 * image_parameters packs the fields of MSMImageParametrs (msm_api.rs:333-347). */
int blz_msm_loaded_binary_parameters(blz_msm* h, uint32_t out[2]);

/* MSMClient::initialize(MSMParams{nof_elements, hbm_point_addr}) (msm_api.rs:72-111).
 * has_hbm=0 <=> hbm_point_addr == None.  mem_type==HBM with has_hbm==0 panics in the reference
 * (unwrap at msm_api.rs:84) -> BLZ_ERR_INVALID_PARAM here. */
int blz_msm_initialize(blz_msm* h, uint32_t nof_elements, int has_hbm, uint64_t hbm_addr, uint64_t hbm_off);

/* MSMClient::start_process (msm_api.rs:113-120): push the configured task to the task queue.
 * The device has a task queue and a result queue (msm_hw_code.rs:19-25): up to TWO tasks may be in
 * flight (initialize / start_process / set_data twice before the first wait_result).  Results are
 * returned in submission order with their labels.  The few-lane tail of a task (upper bucket-reduce
 * levels, Horner, inversion) overlaps the sort + accumulation of the next one, so a stream of MSMs
 * should keep two in flight.  A third submission returns BLZ_ERR_INVALID_PARAM until a result was
 * collected with wait_result; blz_msm_is_engine_ready reports whether a task can be accepted. */
int blz_msm_start_process(blz_msm* h);

/* MSMClient::set_data(MSMInput{points, scalars, params}) (msm_api.rs:155-220).
 *   points == NULL, has_hbm      : scalars only, bases read from the device arena (:163-174)
 *   points != NULL, !has_hbm     : scalars + points streamed (:175-202)
 *   points != NULL, has_hbm      : load_data_to_hbm(points) then scalars (:203-216)
 *   points == NULL, !has_hbm     : silent no-op in the reference (falls through) -> no-op here
 * points_len must be nof_elements * precompute_factor * point_size, scalars_len nof_elements*32.
 * Largest task: points x windows <= 2^32 - 2^26 - 352 321 536 points at precompute_factor 1 (5 x the reference's largest shape;
 * 134 GiB of device memory on BN254), 2^31 - 2^25 points at precompute_factor 8 (2^30 run: 172 GiB); beyond: InvalidPrimitiveParam
 * before anything is copied.  hbm_addr + hbm_off must not wrap around 2^64.
 * Blocking: host buffers may be dropped when it returns (pwrite with O_SYNC, utils.rs:71).
 *
 * STREAMED TASKS.  The reference writes the input to the card's FIFOs in 2048-element chunks (:175-202) and the card counts
 * elements against NUMBER_OF_MSM_ELEMENTS, the register initialize() wrote (msm_hw_code.rs:18-19): a task's bytes may be split
 * over any number of set_data calls.  Same here ({armed_n, received}, launch when received == armed_n): with a
 * task queued (start_process), a set_data whose nof_elements is SMALLER than what the task still lacks is its next slice -
 * lengths are checked against the slice's own nof_elements; any slice sizes (2048-element cadence, ragged tails, zero);
 * every slice in the mode of the first (scalars only: the same hbm_point_addr, the address of the task's FIRST base, in every
 * slice; points + scalars; points + hbm_point_addr: slice k's table is loaded right behind slice k - 1's, its address must say
 * so).  The task is handed to the device piece by piece while later slices are still with the host (the pieces of a one-call
 * DMA-mode task) and is complete with the slice that brings received to nof_elements; more than that is refused and changes
 * nothing, start_process / wait_result with a half-fed task are refused (a task already in flight can still be waited for),
 * blz_msm_reset drops it.  A slice that fails in transfer loses the stream: the task stays queued and may be sent again from
 * its first element.  Results are byte-identical to the one-call task's (tests/test_gpu_msm_stream.py).  The reference's
 * largest DMA-mode shape - tests/integration_msm.rs:386-467, 2^26 elements x 8 bases = a 48 GiB host vector - runs from host
 * slices of any size this way.  blz_msm_stream_progress: {elements received, elements of the queued task}. */
int blz_msm_set_data(blz_msm* h, const uint8_t* points, size_t points_len, const uint8_t* scalars,
                     size_t scalars_len, uint32_t nof_elements, int has_hbm, uint64_t hbm_addr,
                     uint64_t hbm_off);

/* Same semantics with inputs already resident in this device's HBM (device pointers, borrowed until
 * wait_result returns; the slices of a streamed task are COPIED into the handle's staging set and may be dropped on
 * return).  No reference counterpart: the FPGA path has no device-pointer notion; this
 * is what a multi-GPU host or a pipeline that produced scalars on the GPU calls. */
int blz_msm_set_data_device(blz_msm* h, const void* d_points, size_t points_len, const void* d_scalars,
                            size_t scalars_len, uint32_t nof_elements, int has_hbm, uint64_t hbm_addr,
                            uint64_t hbm_off);

/* MSMClient::wait_result (msm_api.rs:222-238): block until the OLDEST task's result is valid.
 * The reference spins forever when nothing is armed; here that is BLZ_ERR_INVALID_PARAM.
 * BOUNDED: the reference polls RESULT_VALID without a deadline; every host-side wait of this library (wait_result,
 * the blocking copies of set_data / load_data_to_hbm / result, the exchange) polls against BLAZE_WAIT_TIMEOUT_MS
 * (environment, milliseconds, default 120000).  On expiry the call returns BLZ_ERR_UNKNOWN (message: "... timed
 * out after N ms"), the task stays queued and the handle turns RESET-ONLY: every call except blz_msm_reset /
 * blz_msm_free / blz_msm_result (of results already collected) fails with BLZ_ERR_UNKNOWN until a reset succeeds;
 * reset itself waits, bounded, for the handle's streams to drain.  Same for the NTT handle. */
int blz_msm_wait_result(blz_msm* h);

/* MSMClient::result (msm_api.rs:240-274): read result_point_size bytes + RESULT_LABEL, pop.
 * Layout Z | Y | X, canonical little-endian, homogeneous projective (tests/msm/mod.rs:397-403);
 * this build always emits the normalised point Z=1 (infinity: Z=0, Y=1, X=0). */
int blz_msm_result(blz_msm* h, uint8_t* out, size_t out_cap, size_t* out_len, uint32_t* label);

/* MSMClient::load_data_to_hbm / get_data_from_hbm (msm_api.rs:299-322): raw bytes at arena byte
 * offset addr+off.  The arena is per device and process-global (points persist across handles, as
 * they persist across clients on the card: tests/integration_msm_hbm.rs:51-56) and behaves like the
 * card's flat memory: a write keeps every byte outside its own range, writes that touch or overlap
 * earlier ones form one contiguous extent (a table may be loaded in pieces), and only a range that
 * was never written cannot be read or used as bases. */
int blz_msm_load_data_to_hbm(blz_msm* h, const uint8_t* points, size_t len, uint64_t addr, uint64_t off);
int blz_msm_load_data_to_hbm_device(blz_msm* h, const void* d_points, size_t len, uint64_t addr, uint64_t off);
int blz_msm_get_data_from_hbm(blz_msm* h, uint8_t* out, size_t len, uint64_t addr, uint64_t off);
/* drop every arena extent of a device (no reference counterpart; the card keeps HBM until reset) */
int blz_arena_release(int device_id);
/* Arena policy (per device, process-wide; default 0).  BLZ_ARENA_DROP_RAW: an extent keeps its bytes as loaded AND a Montgomery
 * copy of the points the tasks gather from (BLS: 96 + 128 bytes per base; 48 + 64 GiB for the reference's largest precompute
 * shape) although only get_data_from_hbm, a later write, an export or a table build ever read the former.  With the bit set,
 * once an extent's copy is complete and every coordinate has been seen to be canonical (< q: such points convert to Montgomery
 * form and back without loss; an extent that holds one that is not keeps its bytes), the raw bytes are freed; whoever needs them
 * again gets them converted back - get_data_from_hbm returns the same bytes as before, a write / blz_arena_export /
 * window-table build / precompute-table check first restores the whole extent (one pass).  Not applied to extents shared with
 * other processes, to extents whose point grid does not start at their first byte, under handles that asked for a window table,
 * or to the even-base copy of a checked precompute table (which is not the whole table).  blz_msm_memory_info shows the effect. */
#define BLZ_ARENA_DROP_RAW 1u
int blz_arena_set_policy(int device_id, uint32_t policy);
/* Cross-process arena (the card's HBM outlives the process that loaded it; GPU memory does not, so a holder
 * process keeps it): blz_arena_export writes one IPC handle per extent of this process to `registry_path`;
 * blz_arena_attach, in ANOTHER process, maps those extents at the same arena offsets (all of them or, on any
 * failure, none), after which hbm_point_addr / get_data_from_hbm address the holder's bytes.  Shared extents are
 * READ-ONLY on both sides from the export on (every process keeps a private Montgomery copy of the points and
 * tracks staleness locally): load_data_to_hbm into one returns WriteError, in the holder too; a load that only
 * touches one starts a separate extent.  blz_arena_release un-shares.  The extents live as long as the holder
 * does.  (HSA_ENABLE_IPC_MODE_LEGACY=0 where the host driver only supports dmabuf IPC.) */
int blz_arena_export(int device_id, const char* registry_path);
int blz_arena_attach(int device_id, const char* registry_path);

/* The HIP stream (hipStream_t, as void*) the handle's tasks are enqueued on, and its device ordinal (nullable): a host that
 * produces scalars - or consumes results - with kernels of its own orders them against the handle's tasks with events on this
 * stream instead of host-side waits.  The driver client of the reference holds file descriptors (dclient.rs:50-59); this is
 * their GPU counterpart.  The stream stays the library's: do not destroy it. */
int blz_msm_stream(blz_msm* h, void** hip_stream, int* device_id);
/* MSMClient::task_label / nof_elements / is_msm_engine_ready (msm_api.rs:278-297) */
int blz_msm_task_label(blz_msm* h, uint32_t* out);
int blz_msm_nof_elements(blz_msm* h, uint32_t* out);
int blz_msm_is_engine_ready(blz_msm* h, uint32_t* out);
/* streamed tasks (blz_msm_set_data): out = {elements the queued task has received so far, elements it was queued with}; {0, 0}
 * with nothing queued.  Diagnostic (the card's FIFO fill is not readable at all). */
int blz_msm_stream_progress(blz_msm* h, uint32_t out[2]);
/* DriverClient::reset (dclient.rs:88-93): drop armed task, queued results and staged data. */
int blz_msm_reset(blz_msm* h);

/* Phase timers of the last completed task, milliseconds (the device clock counters of
 * msm_hw_code.rs:35-46 LAST_TASK_PHASE{1,2,3}_TOTAL_CLOCKS read through get_api, msm_api.rs:324-330):
 * [0] whole device pipeline  [1] k_accumulate kernel alone  [2] digit sort (count+scan+scatter)
 * [3] bucket accumulation (phase 1)  [4] bucket reduce (phase 2)  [5] window combine + affine (phase 3)
 * [6] window bits c  [7] number of windows */
int blz_msm_last_timings(blz_msm* h, float out[8]);
/* 1 when the digit sort of the last completed task ran underneath the accumulation of the task before it (two tasks in
 * flight, the sort's kernels fit beside the accumulation's waves: DESIGN.md section 3), else 0.  Diagnostic. */
int blz_msm_last_sort_hidden(blz_msm* h, int* out);

/* Device memory behind a handle, bytes (get_api, msm_api.rs:324-330, dumps the card's registers; the one figure a GPU host
 * needs that the card never had to report): out = {engine workspace (grown to the largest task seen), staging buffers of this
 * handle, arena: raw bytes as loaded (allocated), arena: Montgomery copies of the bases, arena: window tables + build scratch,
 * total}.  The three arena figures are per DEVICE (every handle of the device reports the same). */
int blz_msm_memory_info(blz_msm* h, uint64_t out[6]);

/* precompute_base_* (tests/msm/mod.rs:360-380) on the device: for each of the n base points (x||y)
 * write PRECOMPUTE_FACTOR = 8 points P, 2^32 P, ..., 2^224 P contiguously to d_out (n*8 points).
 * The reference builds this table on the host before set_data / load_data_to_hbm. */
int blz_msm_precompute_bases_device(int device_id, int curve, const void* d_points, void* d_out, uint64_t n);

/* The window plan the pipeline would use for this input size (no device needed): out = {widest lower
 * window in bits, windows W, unit length L, bucket slots G}; widths (nullable, room for 96 bytes)
 * receives the W window widths, low to high.  The widths always sum to at least the scalar width + 1
 * (signed digits).  Diagnostic; no reference counterpart (the bitstream's plan is fixed). */
int blz_msm_plan(int curve, uint32_t nof_elements, int is_precompute, uint32_t out[4], uint8_t* widths);

/* Resident-base window table (opt-in; no reference counterpart - the closest is the caller-supplied x8 table of
 * MSMInit.is_precompute, msm_api.rs:40-50, which costs 16 window passes per element where the plain path needs 12).
 * enable: 0 off, 1 on where it was measured to pay (BLS12-377 / BLS12-381; BN254's 64-byte points are already
 * gathered at the memory system's rate and lose 4 % with a table, so 1 leaves BN254 on the plain path), 2 always.
 * With it, a pf = 1 handle whose bases live in the arena (hbm_point_addr) builds the table of their window multiples
 * 2^(c j) P, j < W = ceil(257 / c) - 1.9 s of the chip for 2^26 BLS12-381 bases, paced by the tasks (below), W x the memory
 * of the Montgomery copy (2^26: 10 x 8 GiB) - and keeps it with the arena extent: a later write of up to 2^18 bases has their rows
 * re-tabulated ahead of the next task, a larger one drops the table (the tasks that follow rebuild it).  Tasks over
 * those bases then add every window's digit into ONE bucket set:
 * 10 windows of 26 bits at 2^26 (671 M bucket additions, 2^25 buckets) instead of 12 windows of 21-23 bits (805 M).
 * Results are bit-identical to the plain path's.  Falls back to the plain path (silently; BLAZE_LOG=1 says why) when
 * the table does not fit the free memory, when a base has even order (a multiple at infinity cannot be tabulated;
 * never the case in the r-torsion), or for a task over a sub-range that wants a different window width.
 * A handle with a scalar range (blz_msm_set_scalar_range) tabulates 2^(bit_lo + c j) P for the windows of its range.
 * New handles start with 0 (off).  No other value of `enable` is accepted (InvalidPrimitiveParam). */
int blz_msm_set_window_table(blz_msm* h, int enable);
/* The table's build is never one lump inside a task: it is cut into chunks of 196 608 bases (~5.5 ms of the chip), every task
 * launched over the bases first enqueues four of them on its own stream and takes the plain path, like every task until the
 * last chunk has completed; the next task adopts the table.  Results are bit-identical either way.  A host that wants the
 * table in place before its first task calls this after load_data_to_hbm: it allocates the table (do this with the load: an
 * 80 GiB hipMalloc takes 0.3 ms on a clean device and seconds on one that has memory to scrub), enqueues the first chunks
 * (wait_ms = 0) or ALL the remaining ones (wait_ms != 0) for the nof_elements bases at hbm_addr + hbm_off and waits up to
 * wait_ms (< 0: BLAZE_WAIT_TIMEOUT_MS) for the build; *ready = 1 when the table is in place, 0 otherwise (still building, not opted in, no memory, a base of even
 * order, another handle's table serves the extent). */
int blz_msm_prepare_window_table(blz_msm* h, uint32_t nof_elements, uint64_t hbm_addr, uint64_t hbm_off, int wait_ms, int* ready);
/* out = {table bytes, window bits c, windows W, build time in microseconds} of the table the handle's last HBM task
 * used; zeros when it took the plain path */
int blz_msm_window_table_info(blz_msm* h, uint64_t out[4]);

/* Checked-table plan for precompute handles (opt-in; MSMInit.is_precompute, msm_api.rs:39-50).  The reference's precompute mode
 * has the CALLER supply, per element, the 8 bases B_j = 2^(32 j) P (precompute_base_*: tests/msm/mod.rs:360-380) and defines the
 * task as sum_i sum_j s_(i,j) B_(i,j) over the 32-bit chunks of the scalars.  Served literally - the default - that is an 8n-point
 * MSM of 32-bit scalars: 16 bucket additions per element where the same elements without a table need 12.  With enable = 1 a
 * handle whose bases live in the arena (hbm_point_addr) CHECKS the table once per load - on the device, base by base: B_(i,0) on the
 * curve and B_(i,j) == 2^32 B_(i,j-1) for j = 1..7 (32 doublings each, compared projectively; 0.7 s for 2^26 BN254 elements,
 * 1.3 s on the BLS curves) - and, if it holds, sums sum_i sum_k (s_(i,2k) + 2^32 s_(i,2k+1)) B_(i,2k) instead: 4n points with 64-bit
 * scalars, three windows of 22 / 22 / 21 bits, 12 additions per element into 3 shared bucket sets, over a Montgomery copy of
 * the even bases only (half the copy's memory).  The two sums are the same group element exactly when the check holds, and the
 * result is emitted normalised (Z = 1), so the bytes are identical to the exact path's.  A table that fails the check (any base
 * off the curve, any multiple that is not 2^32 times its predecessor, a multiple at infinity) keeps the exact path - silently;
 * BLAZE_LOG=1 says so, blz_msm_precompute_plan_info reports it.  A write into a consistent table has the next task check the elements
 * it touched (only those); a write into a refuted one, the whole range again.  The check runs inside the first set_data / start_process that launches a task over the bases (the call blocks
 * for it), or in blz_msm_prepare_precompute_plan for a host that wants to pay with the load.  Tasks that bring their own points
 * (DMA mode, and set_data with points AND an hbm address) always take the exact path: they are link-bound, and a check per task
 * would cost more than it saves.
 * New handles start with 0.  InvalidPrimitiveParam for a handle without is_precompute. */
int blz_msm_set_precompute_plan(blz_msm* h, int enable);
/* run the check (and build the even-base copy) now for the nof_elements elements at hbm_addr + hbm_off; *consistent = 1 when
 * tasks over them will take the plan, 0 otherwise (not opted in, table refuted, bases off the extent's element grid) */
int blz_msm_prepare_precompute_plan(blz_msm* h, uint32_t nof_elements, uint64_t hbm_addr, uint64_t hbm_off, int* consistent);
/* out = {1 if the handle's last HBM task took the plan, state of the check of its bases (0 not asked / not checked, 1 consistent,
 * 2 refuted), device time of the check in microseconds, bytes of the even-base Montgomery copy} */
int blz_msm_precompute_plan_info(blz_msm* h, uint64_t out[4]);

/* Sharding by scalar chunk (multi-GPU; no reference counterpart - README.md:20-22 leaves the split to a "management
 * layer").  A handle with a scalar range sums only bits [bit_lo, bit_hi) of every scalar it is given and returns
 * 2^bit_lo x that sum: the partial results of the ranges of a partition of [0, 256) add up to the full MSM, like the partial
 * results of element chunks do (blz_msm_combine_partials / blz_msm_all_gather_combine), and the two splits compose.
 * Why: Pippenger's cost per rank is (elements x windows) additions + (windows x 2^(c-1)) bucket slots to reduce, and a rank
 * that gets 1/8 of the ELEMENTS has to narrow its windows (2^23 elements: 15 windows of 19 bits - 126 M additions) where a
 * rank that gets 1/4 of the elements' BITS for half of the elements keeps the big job's windows (2^25 elements x 3
 * windows of 22 bits - 101 M).  32-bit aligned ranges, precompute_factor 1; (0, 0) or (0, 256) = the whole scalar. */
int blz_msm_set_scalar_range(blz_msm* h, uint32_t bit_lo, uint32_t bit_hi);
/* The split blz picks for `nranks` equal devices: out = {first element, element count, bit_lo, bit_hi} of `rank`.
 * R ranges of 256 / R bits x nranks / R element chunks, R in {1, 2, 4, 8} dividing nranks, chosen by the window planner's
 * cost estimate of a rank's task; rank = chunk * R + range.  BLAZE_SHARD=elements forces the element split (R = 1),
 * BLAZE_SHARD=bits the largest R.  Host-side only (no device needed). */
int blz_msm_shard_layout(int curve, uint32_t nof_elements, int nranks, int rank, uint32_t out[4]);
/* The same choice with the flow's transfers priced in.  A split into R scalar ranges makes the R ranks of a range group need
 * the SAME element chunk: R x the scalar bytes per rank on its PCIe link when the scalars come from host memory with every
 * task (the reference's HBM flow, tests/integration_msm_hbm.rs:57-100), R x the bases per rank in device memory - and on
 * the link too when the bases travel with every task (DMA flow).  With t = bytes over the rank's link / 56 GB/s (measured)
 * and c = the window planner's compute estimate, cost(R) = max(c + 0.15 t, 1.1 t) per task (a stream of tasks overlaps the
 * two, not for free: profiles/r04_shard_layouts.txt), and a layout whose per-rank bases and scalars exceed half of the
 * device memory is not considered.  R > 1 is taken only when it beats the element split by more than
 * 2 % (the simpler layout wins ties).  flags: BLZ_SHARD_SCALARS_FROM_HOST, BLZ_SHARD_BASES_FROM_HOST; 0 = everything
 * resident (what blz_msm_shard_layout prices).  out = {first element, element count, bit_lo, bit_hi, R, estimated compute us,
 * estimated link us, device MiB per rank (bases raw + Montgomery copy + scalars)}.  Host-side only. */
#define BLZ_SHARD_SCALARS_FROM_HOST 1u
#define BLZ_SHARD_BASES_FROM_HOST 2u
int blz_msm_shard_layout_ex(int curve, uint32_t nof_elements, int nranks, int rank, uint32_t flags, uint32_t out[8]);
/* the estimates of one candidate (R ranges) for tools and tests: out as above; InvalidPrimitiveParam if R does not divide nranks */
int blz_msm_shard_layout_candidate(int curve, uint32_t nof_elements, int nranks, int rank, uint32_t flags, int R, uint32_t out[8]);

/* Multi-GPU: add G partial results (each result_size bytes, as returned by blz_msm_result on each
 * rank, in rank order) on this handle's device and emit the normalised sum, for hosts that move the
 * partials themselves (blz_msm_all_gather_combine below does the exchange too). */
int blz_msm_combine_partials(blz_msm* h, const uint8_t* partials, size_t count, uint8_t* out, size_t out_cap);

/* Multi-GPU exchange inside the library (one process per GPU, or one handle per device of one process): an
 * RCCL communicator rank per handle, then  all-gather of the ranks' partial results over xGMI + rank-ordered
 * add on the device + normalise, so every rank returns the same bytes.  (RCCL's reduce operators are
 * arithmetic, not a group law: all-gather + local add instead of all-reduce)  The reference
 * leaves the multi-device layer to a "management layer" (README.md:20-22); this is that layer for a Rust / C++
 * host.  RCCL is resolved at run time (dlopen of librccl.so.1); without it these return LoadFailed.
 *   rank 0:        blz_comm_unique_id(id); the host ships the 128 bytes to the other ranks (MPI, a file, a pipe ...)
 *   every rank:    blz_msm_comm_init(h, rank, nranks, id)                        -- collective
 *   per MSM:       blz_msm_result(h, partial ...); blz_msm_all_gather_combine(h, partial, out, cap) -- collective
 *   every rank:    blz_msm_comm_free(h)   (blz_msm_free does it too)
 * One process driving several devices from one thread (one handle per device, rank i = handles[i]):
 *   once:          blz_msm_comm_init_all(handles, n)                             -- one RCCL group (ncclGroupStart/End)
 *   per MSM:       blz_msm_all_gather_combine_all(handles, n, partials, out, cap) -- partials / out: n x result_size
 * (blz_msm_comm_init / blz_msm_all_gather_combine are blocking rendezvous: called for several handles from ONE
 * thread they would wait for each other - use the _all forms there.)
 * Deadlines: the bring-up runs against BLAZE_COMM_TIMEOUT_MS (default 60000) and fails with BLZ_ERR_UNKNOWN when a
 * peer never arrives; the exchange is a bounded wait like wait_result (BLAZE_WAIT_TIMEOUT_MS). */
#define BLZ_COMM_ID_BYTES 128
int blz_comm_unique_id(uint8_t out[BLZ_COMM_ID_BYTES]);
int blz_msm_comm_init(blz_msm* h, int rank, int nranks, const uint8_t id[BLZ_COMM_ID_BYTES]);
int blz_msm_all_gather_combine(blz_msm* h, const uint8_t* partial, uint8_t* out, size_t out_cap);
int blz_msm_comm_init_all(blz_msm* const* handles, int n);
int blz_msm_all_gather_combine_all(blz_msm* const* handles, int n, const uint8_t* partials, uint8_t* out, size_t out_cap);
int blz_msm_comm_free(blz_msm* h);

/* ------------------------------------------------------------------ NTT (src/ingo_ntt/ntt_api.rs) */

/* DriverClient::new + NTTClient::new(NTT::Ntt, dclient) (ntt_api.rs:26-31).  log_size = 27 is the
 * reference shape (ntt_data.rs:65); smaller sizes exist for tests.  Field: BLS12-381 Fr
 * (BASELINE.json), forward transform, natural order in/out, omega = 2^log_size-th root derived from
 * the multiplicative generator 7. */
int blz_ntt_new(int device_id, int log_size, blz_ntt** out);
/* same, with the transform direction: inverse != 0 gives x[i] = n^-1 sum_k X[k] omega^(-ik) (natural order;
 * the reference has no such knob: SURVEY.md 8(f) rank 3) */
int blz_ntt_new_ex(int device_id, int log_size, int inverse, blz_ntt** out);
/* same over the scalar field of another curve (field = enum blz_curve: BLS12-377 Fr, generator 22,
 * two-adicity 47; BN254 Fr, generator 5, two-adicity 28).  The bitstream the
 * reference drives is built for one field; here the kernels are instantiated per field. */
int blz_ntt_new_field(int device_id, int field, int log_size, int inverse, blz_ntt** out);
/* same, with flags.  BLZ_NTT_NO_FACTOR_TABLE: never allocate the per-element boundary-factor table of a 2^27 transform's
 * second pass (n x 32 bytes: 4 GiB beside the handle's 12 GiB of buffers): the pass steps the factors along each lane's rows
 * instead - the kernel a memory-tight device gets anyway (the table is an optimisation worth ~2 %, allocated when it fits) and the
 * one every smaller transform runs.  blz_ntt_info says which one a handle got. */
#define BLZ_NTT_NO_FACTOR_TABLE 1u
int blz_ntt_new_ex2(int device_id, int field, int log_size, int inverse, uint32_t flags, blz_ntt** out);
/* The transform's CONVENTION, chosen by the caller.  The reference states none of it (NttInit {} is empty, ntt_api.rs:8-23; its
 * golden files are external, tests/integration_ntt.rs:15-18): this build's default is X[k] = sum_i x[i] w^(i k) with
 * w = g^((r - 1) / 2^log_size), g the field's multiplicative generator (7 / 22 / 5 for BLS12-381 / BLS12-377 / BN254 Fr), natural
 * order in and out.  A host that holds vectors made for the card says here what they assume:
 *   root   (nullable) 32 bytes, canonical little-endian: ANY primitive 2^log_size-th root of unity replaces w; checked on the
 *          device (root < r and root^(2^(log_size - 1)) == -1), else BLZ_ERR_INVALID_PARAM;
 *   flags  BLZ_NTT_NO_FACTOR_TABLE as above; BLZ_NTT_INVERSE: x = n^-1 sum_k X[k] w^(-i k) (the same w: the inverse of the
 *          handle without the flag); BLZ_NTT_BITREV_INPUT / BLZ_NTT_BITREV_OUTPUT: position p of the buffer passed to
 *          set_data / returned by result holds the element of index bitrev(p) over log_size bits (the orders decimation-in-time
 *          inputs / decimation-in-frequency outputs come in) - folded into the first pass's loads and the last pass's stores.
 * Handles made without root and order flags are byte-identical to blz_ntt_new_ex2's. */
#define BLZ_NTT_INVERSE 2u
#define BLZ_NTT_BITREV_INPUT 4u
#define BLZ_NTT_BITREV_OUTPUT 8u
int blz_ntt_new_ex3(int device_id, int field, int log_size, uint32_t flags, const uint8_t* root, blz_ntt** out);
/* out = {device bytes the handle holds (two transform buffers + scratch + twiddle / factor tables), 1 if pass 2 reads the
 * per-element factor table / 0 if it steps its factors, 1 if pass 1 reads the column-independent boundary table, log_size} */
int blz_ntt_info(blz_ntt* h, uint64_t out[4]);
void blz_ntt_free(blz_ntt* h);
/* NTTClient::initialize(NttInit{}) (ntt_api.rs:37-56) */
int blz_ntt_initialize(blz_ntt* h);
/* NTTClient::set_data(NTTInput{buf_host, data}) (ntt_api.rs:72-87): data = 2^log_size x 32 B LE, canonical field
 * elements; a word >= r is taken as the residue it represents (the output is canonical either way) */
int blz_ntt_set_data(blz_ntt* h, size_t buf_host, const uint8_t* data, size_t len);
int blz_ntt_set_data_device(blz_ntt* h, size_t buf_host, const void* d_data, size_t len);
/* NTTClient::start_process(Some(buf_kernel)) (ntt_api.rs:58-70): in-place transform of that buffer */
int blz_ntt_start_process(blz_ntt* h, size_t buf_kernel);
/* NTTClient::wait_result (ntt_api.rs:89-108) */
int blz_ntt_wait_result(blz_ntt* h);
/* NTTClient::result(Some(buf)) (ntt_api.rs:110-124) */
int blz_ntt_result(blz_ntt* h, size_t buf, uint8_t* out, size_t out_cap);
int blz_ntt_result_device(blz_ntt* h, size_t buf, void* d_out, size_t out_cap);
/* NTTClient::result(Some(buf)) followed by NTTClient::set_data(NTTInput{buf_host: buf, data}) - what every cycle of the
 * reference's double-buffered loop does while the kernel runs on the other buffer (tests/integration_ntt.rs:102-136) - as ONE
 * call that drives the link in both directions at once: the buffer leaves for prev_out piece by piece and next_in lands in the
 * places that have left.  Same preconditions as the two calls (the buffer must not be under transform); blocking; both host
 * buffers may be dropped / read when it returns.  in_len = 2^log_size x 32, out_cap >= that. */
int blz_ntt_exchange(blz_ntt* h, size_t buf, const uint8_t* next_in, size_t in_len, uint8_t* prev_out, size_t out_cap);
/* the HIP stream (hipStream_t, as void*) the handle's transforms run on, and its device ordinal (nullable); see blz_msm_stream */
int blz_ntt_stream(blz_ntt* h, void** hip_stream, int* device_id);
/* DriverClient::reset (dclient.rs:88-93) without the 100 ms sleep */
int blz_ntt_reset(blz_ntt* h);
/* kernel time of the last transform in ms (what benches/ntt_bench.rs:34-39 times, minus reset()) */
int blz_ntt_last_kernel_ms(blz_ntt* h, float* out);
/* NTTBanks::preprocess / postprocess (ntt_data.rs:80-156) as device permutations, for byte
 * compatibility with bank files of the FPGA flow; n = 2^log_size elements (log_size >= 10), 16 banks
 * contiguous (n/16 elements each); 2^27 uses the reference's 512 groups x 256 block pairs, smaller sizes
 * scale the group count (n >> 18, at least 1). */
int blz_ntt_banks_preprocess_device(blz_ntt* h, const void* d_in, void* d_banks);
int blz_ntt_banks_postprocess_device(blz_ntt* h, const void* d_banks, void* d_out);

/* ------------------------------------------------------------------ device / host memory helpers */

/* Device memory owned by the library (hosts without a tensor library of their own; bench.py and the tests use these). */
int blz_device_malloc(int device_id, size_t bytes, void** out);
int blz_device_free(int device_id, void* p);
/* Host memory the runtime can DMA from / to without staging (page-locked, mapped for `device_id`): copies out of and into such
 * buffers are truly asynchronous, which is what lets blz_ntt_exchange keep both directions of the link busy from one thread.
 * Any pointer works everywhere (pageable memory is staged by the runtime); these are for hosts that own their I/O vectors. */
int blz_host_malloc(int device_id, size_t bytes, void** out);
int blz_host_free(void* p);
int blz_memcpy_h2d(int device_id, void* d_dst, const void* src, size_t bytes);
int blz_memcpy_d2h(int device_id, void* dst, const void* d_src, size_t bytes);
#ifdef __cplusplus
}
#endif
#endif /* BLAZE_HIP_H */
