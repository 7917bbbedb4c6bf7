// Pippenger windowed-bucket G1 MSM for gfx950 (the device task of SURVEY.md a7: what the FPGA
// bitstream behind src/ingo_msm/msm_hw_code.rs:6-54 computes; kernels are new design).
//
// Pipeline (main stream, then the tail stream from the second reduce level on):
//   msm_sort.hip   signed window digits of every scalar, two-level LDS counting sort ->
//                  count[] per bucket, then (point index | sign) entries grouped by bucket
//   k_scan_*       exclusive scan: bucket offsets + accumulate-unit offsets (runs split at L)
//   k_fill_units   unit -> bucket map;  k_unit_* order the units by run length (descending)
//   k_accumulate   one lane per unit: gathers its run of points, XYZZ mixed adds   [phase 1]
//   k_combine_units   only when a bucket needed more than one unit
//   k_reduce_level Sum_b b*S_b per virtual window by segmented running sums        [phase 2]
//   k_finish(_row) stitch virtual windows, Horner over windows, one inversion, Z=1|y|x  [phase 3]
// A task is enqueued in steps - begin / per piece sort_slice + accumulate_slice / end - so that host buffers can be handed to
// the device piece by piece while they cross the PCIe link (pieces share one bucket space: k_accumulate_cont).
//
// HBM layout: points AoS Montgomery (one 128-B line per BLS point, 64 B per BN254 point), scalars raw
// 32 B LE, entries u32, bucket partials AoS XYZZ (4N dwords).  The arithmetic (v_mad_u64_u32) bounds
// the dominant kernels, not HBM.
#include "msm_engine.hpp"
#include <mutex>
#include "field.hip.hpp"
#include "msm_digits.hip.hpp"

namespace blz {

size_t fq_bytes(int curve) { return curve == BLZ_BN254 ? 32 : 48; }
size_t mont_point_bytes(int curve) { return curve == BLZ_BN254 ? 64 : 128; }

// ------------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------------
// Window choice by a cost model fitted to MI355X measurements (tools/sweep_c.py, 2^16..2^26 points):
//   0.163 ns per entry (digit sort + one mixed add), 0.62 ns per occupied bucket (two full adds in the
//   bucket reduce, unit bookkeeping), 0.03 ns per empty bucket.
// `ebits` is the expected significant width of the scalars (bit length of r: 255 / 253 / 254; 32 for a
// pf = 8 chunk).  It only steers the choice - W always covers sbits + 1 bits, so scalars above 2^ebits
// are still summed correctly - but it matters: the top window holds ebits - (W-1) c real bits, i.e.
// fewer occupied buckets than 2^(c-1), or (c = 17 on 255-bit scalars) nothing but the carry of the
// window below.  A window whose entries fall into a handful of buckets used to cost 0.1 - 0.4 ns per
// entry extra; with the run-splitting units, the cooperative fills and the quad-folded combine it is
// within noise (t_hot), so odd c are no longer avoided.
//
// Layouts searched: W windows, the lowest k of width cmin+1, the next W-1-k of width cmin, and a top
// window of max(cmin, what is left of sbits+1) bits (its upper bits are zero for canonical scalars, so
// its signed digits never go negative; only 2^(real bits) of its buckets are occupied).  k = 0 with a
// top window of cmin bits is the uniform plan; BLAZE_MSM_PLAN c= forces that one.
static MsmPlan search_plan(uint32_t npts, int sbits, int ebits, int force_c, int split_ns, int max_w) {
    MsmPlan best;
    double best_cost = 1e300;
    const double t_entry = 0.163, t_bucket = 0.62, t_empty = 0.03, t_hot = 0.01;
    const int need = sbits + 1;
    const double t_split = (double)split_ns;
    for (int cmin = 3; cmin <= 23; ++cmin) {
        if (force_c > 0 && cmin != force_c) continue;
        for (int W = 1; W <= max_w; ++W) {
            // entries are indexed with u32, and the kernels' strided walks over them (i += stride, up to 2^24 lanes) must not wrap:
            // at 2^32 - 4 entries they did - wrong sums after seconds of re-walking the bins (tests/probes/big_probe.py)
            if ((uint64_t)npts * W > MSM_MAX_ENTRIES) break;
            const int lower = W - 1;
            for (int k = 0; k <= lower; ++k) {
                if (force_c > 0 && k != 0) break;
                if (k > 0 && cmin + 1 > 23) break;
                const int low_bits = lower * cmin + k;
                if (low_bits >= need + cmin) break;            // a window too many
                int top = need - low_bits;
                if (top < cmin) top = cmin;
                if (top > 23) continue;
                if (top - cmin > 7) continue;                  // k_finish walks 2^(top - cmin) virtual windows in sequence
                if (force_c > 0 && top != cmin) continue;
                // cost over the windows
                double cost = 0;
                uint64_t G = 0;
                int off = 0;
                for (int w = 0; w < W; ++w) {
                    const int cw = w == lower ? top : (w < k ? cmin + 1 : cmin);
                    const double Bw = (double)(1ull << (cw - 1));
                    int t = ebits - off;                        // real scalar bits in this window
                    if (t > cw) t = cw;
                    double entries, active;
                    if (t >= cw) { entries = npts; active = Bw; }
                    else if (t > 0) { entries = npts; active = (double)(1ull << t) + 1; if (active > Bw) active = Bw; }
                    else if (t == 0) { entries = 0.5 * npts; active = 1; }   // carry of a full window below
                    else { entries = 0; active = 0; }
                    if (active > entries) active = entries;
                    cost += entries * t_entry + active * t_bucket + (Bw - active) * t_empty;
                    if (active > 0 && entries / active > 8192.0) cost += entries * t_hot;
                    G += 1ull << (cw - 1);
                    off += cw;
                    // a window of m > 1 virtual windows costs k_finish 2 (m - 1) + 1 more quad additions
                    // (~6 us each, sequential): irrelevant at 2^26, decisive below 2^21
                    const int m = 1 << (cw - cmin);
                    if (m > 1) cost += (2.0 * (m - 1) + 1.0) * t_split;
                }
                if (G > (1ull << 26)) continue;                 // workspace bound (partials: 192 B each)
                if (cost < best_cost) {
                    best_cost = cost;
                    best = MsmPlan();
                    best.npts = npts; best.sbits = sbits; best.W = W; best.G = G;
                    best.c = k > 0 ? cmin + 1 : cmin;
                    best.Bw = 1u << (cmin - 1);
                    uint32_t b = 0;
                    for (int w = 0; w < W; ++w) {
                        const int cw = w == lower ? top : (w < k ? cmin + 1 : cmin);
                        best.width[w] = (uint8_t)cw;
                        best.boff[w] = b;
                        b += 1u << (cw - 1);
                    }
                    best.boff[W] = b;
                    best.Wv = (int)(G >> (cmin - 1));
                    best.cost = cost;
                    best.ebits = ebits;
                }
            }
        }
    }
    // Unit length: a unit is one lane's sequential chain (L mixed adds of ~15 us each when the SIMD has
    // little else to run), so on small inputs long units ARE the runtime: 2^18 points, c = 13 spent
    // 3.7 of 7.2 ms waiting for the 256-long units of the short top window.  Keep the longest chain
    // below about half of the throughput-bound time of the whole accumulation.
    {
        double chain = (double)npts * best.W * 5.5e-6;
        uint32_t L = 16;
        while (L < 256 && 2.0 * L <= chain) L <<= 1;
        best.L = L;
    }
    return best;
}

// The search walks a few hundred thousand layouts (0.2 - 0.6 ms of host time): a task of 8192 elements - the reference's
// default size - is done on the device in 1.9 ms and asks for its plan twice (set_data's check, begin()), so the last few
// answers are kept.  Keyed on everything the search reads.
MsmPlan make_plan(uint32_t npts, int sbits, int ebits, int force_c) {
    if (ebits <= 0 || ebits > sbits) ebits = sbits;
    const int split_ns = plan_override("split_ns", 6000);  // BLAZE_MSM_PLAN: tests set 0 - mixed widths at any size
    struct Memo {
        uint32_t npts = 0;
        int sbits = 0, ebits = 0, force_c = 0, split_ns = 0;
        MsmPlan plan;
    };
    static std::mutex mu;
    static Memo memo[16];
    static unsigned next = 0;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const Memo& m : memo)
            if (m.sbits == sbits && m.npts == npts && m.ebits == ebits && m.force_c == force_c && m.split_ns == split_ns) return m.plan;
    }
    // The 64-bit chunks of a checked precompute table (arena_tables.hip resolve_arena_task): the cost model, fitted to 256-bit scalars,
    // moves from four windows of 16 / 17 bits to three of 22 / 22 / 21 at 2^24.5 points; measured (BN254, same box), the three-window
    // plan already wins at 2^24 points (5.60 against 6.53 ms per MSM) and loses at 2^22 (2.67 against 1.81): four windows leave
    // 163 K buckets of hundreds of entries - a few units per lane - and the accumulation's last wave of units idles the chip.
    const int max_w = (sbits == 64 && force_c == 0 && npts >= (3u << 22)) ? 3 : MSM_MAX_W;
    const MsmPlan P = search_plan(npts, sbits, ebits, force_c, split_ns, max_w);
    std::lock_guard<std::mutex> lk(mu);
    Memo& m = memo[next++ % 16];
    m.npts = npts; m.sbits = sbits; m.ebits = ebits; m.force_c = force_c; m.split_ns = split_ns;
    m.plan = P;
    return P;
}

// Window-table plans (msm_engine.hpp MsmPlan::table, msm_impl.hip.hpp k_build_window_table).  Costs fitted to the round-3 kernels on the table
// shape: 0.130 ns per entry (one mixed addition; the sort is hidden), 0.34 ns per bucket slot (two full additions in the
// level-0 reduce - 12.6 ms for the 2^25 buckets of c = 26 - scans, unit lists).  2^26 bases: c = 26, 10 windows (671 M
// additions instead of the 805 M of the 12-window plan without a table: 106.6 ms per MSM against 116.2; c = 24, 11 windows:
// 108.7); 2^24: c = 24 (30.0 against 33.2 ms); 2^23: c = 22 (17.2 against 18.3 ms).
int table_window_bits(uint32_t npts, int need_bits) {
    const int forced = plan_override("table_c", 0);
    int best = 0;
    double best_cost = 1e300;
    for (int c = 16; c <= 26; ++c) {
        if (forced > 0 && c != forced) continue;
        const int W = table_windows(c, need_bits);
        if (c > 16 && table_windows(c - 1, need_bits) == W && forced <= 0) continue;   // same additions, twice the buckets: dominated
        if ((uint64_t)npts * W >= (1ull << 30)) continue;
        const double cost = (double)npts * W * 0.130 + (double)(1ull << (c - 1)) * 0.34;
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}

MsmPlan make_table_plan(uint32_t npts, int c, int need_bits) {
    MsmPlan P;
    if (c < 16 || c > 26 || need_bits < 2 || need_bits > 257) return P;
    const int W = table_windows(c, need_bits);
    if ((uint64_t)npts * W >= (1ull << 30)) return P;
    P.npts = npts; P.sbits = need_bits - 1; P.c = c; P.W = W; P.table = true;
    P.G = 1ull << (c - 1);
    for (int w = 0; w < W; ++w) { P.width[w] = (uint8_t)c; P.boff[w] = 0; }
    P.boff[W] = (uint32_t)P.G;
    P.Bw = 1u << (c - 1 - 5);   // the one window is walked as 32 virtual windows (k_finish stitches them)
    P.Wv = 32;
    double chain = (double)npts * W * 5.5e-6;
    uint32_t L = 16;
    while (L < 256 && 2.0 * L <= chain) L <<= 1;
    P.L = L;
    return P;
}

// zero-fill (hipMemsetAsync's fill kernel took 0.7 ms for the 71 MB bucket-count array: ~100 GB/s)
__global__ __launch_bounds__(256) void k_zero(uint4* __restrict__ p, size_t n16) {
    __builtin_amdgcn_s_setprio(3);   // sort-stage kernel: may run underneath another task's accumulation (msm_sort3.hip)
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = z;
}

// scalar-range tasks: dst = words [w0, w0 + nw) of every 8-word scalar, zero-extended to 8 words
__global__ __launch_bounds__(256) void k_extract_range(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint64_t n, uint32_t w0,
                                                       uint32_t nw) {
    __builtin_amdgcn_s_setprio(3);   // sort-stage kernel
    const uint64_t total = n * 8;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        const uint32_t j = (uint32_t)(i & 7u);
        dst[i] = j < nw ? src[(i & ~(uint64_t)7) + w0 + j] : 0u;
    }
}

__global__ __launch_bounds__(64) void k_words_to_pinned(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint32_t n) {
    __builtin_amdgcn_s_setprio(3);
    if (threadIdx.x < n) __hip_atomic_store(dst + threadIdx.x, src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
int copy_words_to_pinned(void* host_pinned, const void* d_src, uint32_t dwords, hipStream_t st) {
    if (dwords > 64) return fail(BLZ_ERR_UNKNOWN, "copy_words_to_pinned: %u dwords", dwords);
    hipLaunchKernelGGL(k_words_to_pinned, dim3(1), dim3(64), 0, st, (uint32_t*)host_pinned, (const uint32_t*)d_src, dwords);
    BLZ_HIP(hipGetLastError(), BLZ_ERR_READ);
    return BLZ_OK;
}

// p[i] = i: the "every bucket has exactly one sum, at its own index" unit_off of piecewise tasks (begin() / end())
__global__ __launch_bounds__(256) void k_iota(uint32_t* __restrict__ p, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) p[i] = (uint32_t)i;
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of (count, units(count)) packed in one u64: low = entries, high = units
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = 256 * SCAN_ITEMS;

__device__ __forceinline__ uint64_t pack_cu(uint32_t cnt, uint32_t L) {
    return (uint64_t)cnt | ((uint64_t)((cnt + L - 1) / L) << 32);
}
__device__ __forceinline__ uint64_t block_sum_u64(uint64_t v, uint64_t* sh) {
    // wave reduce then across 4 waves
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    uint64_t t = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void k_scan_reduce(const uint32_t* __restrict__ count, uint64_t G, uint32_t L,
                                                     uint64_t* __restrict__ blocksums, uint32_t* __restrict__ stats) {
    __builtin_amdgcn_s_setprio(3);   // sort-stage kernel: may run underneath another task's accumulation (msm_sort3.hip)
    __shared__ uint64_t sh[4];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    uint64_t acc = 0;
    uint32_t mx = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
        if (i < G) {
            uint32_t cn = count[i];
            acc += pack_cu(cn, L);
            mx = cn > mx ? cn : mx;
        }
    }
    uint64_t tot = block_sum_u64(acc, sh);
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_down(mx, o, 64); mx = t > mx ? t : mx; }
    // (guarded: 5 x 10^4 unconditional atomics on one address cost 0.5 ms; the value only grows)
    if ((threadIdx.x & 63) == 0 && mx > *(volatile uint32_t*)&stats[1]) atomicMax(&stats[1], mx);
    if (threadIdx.x == 0) blocksums[blockIdx.x] = tot;
}

// single block: exclusive scan of blocksums in place; totals -> stats[0] (units), stats[2] (entries)
// (256 threads: the kernel belongs to the sort stage, which may have to fit beside another task's accumulation -
// a 1024-thread block needs four waves per SIMD at once and 4 x its registers of the ~100 VGPRs the accumulation leaves)
constexpr int SCAN_SUMS_THREADS = 256;
__global__ __launch_bounds__(SCAN_SUMS_THREADS) void k_scan_sums(uint64_t* __restrict__ blocksums, uint32_t nblocks,
                                                                 uint32_t* __restrict__ stats) {
    __builtin_amdgcn_s_setprio(3);   // sort-stage kernel: may run underneath another task's accumulation (msm_sort3.hip)
    __shared__ uint64_t sh[SCAN_SUMS_THREADS];
    __shared__ uint64_t carry_sh;
    if (threadIdx.x == 0) carry_sh = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += SCAN_SUMS_THREADS) {
        uint32_t i = base + threadIdx.x;
        uint64_t v = i < nblocks ? blocksums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < SCAN_SUMS_THREADS; o <<= 1) {
            uint64_t t = threadIdx.x >= (uint32_t)o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        uint64_t incl = sh[threadIdx.x];
        uint64_t carry = carry_sh;
        if (i < nblocks) blocksums[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == SCAN_SUMS_THREADS - 1) carry_sh = carry + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stats[0] = (uint32_t)(carry_sh >> 32);
        stats[2] = (uint32_t)(carry_sh & 0xffffffffu);
    }
}

// per-block exclusive scan with the block's base; writes off[], unit_off[] and turns count[] into
// the scatter cursor (count[g] = off[g]).  Element G (one past the end) receives the totals.
__global__ __launch_bounds__(256) void k_scan_final(uint32_t* __restrict__ count, uint64_t G, uint32_t L,
                                                    const uint64_t* __restrict__ blocksums,
                                                    uint32_t* __restrict__ off, uint32_t* __restrict__ unit_off) {
    __builtin_amdgcn_s_setprio(3);   // sort-stage kernel: may run underneath another task's accumulation (msm_sort3.hip)
    __shared__ uint64_t sh[256];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS];
    uint64_t tsum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        uint64_t i = base + k;
        v[k] = i < G ? pack_cu(count[i], L) : 0;
        tsum += v[k];
    }
    sh[threadIdx.x] = tsum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        uint64_t t = threadIdx.x >= (uint32_t)o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint64_t run = blocksums[blockIdx.x] + sh[threadIdx.x] - tsum;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        uint64_t i = base + k;
        if (i <= G) {
            off[i] = (uint32_t)(run & 0xffffffffu);
            unit_off[i] = (uint32_t)(run >> 32);
            if (i < G) count[i] = (uint32_t)(run & 0xffffffffu);
        }
        run += v[k];
    }
}

// ------------------------------------------------------------------------------------------------
// Units ordered by run length (descending), so the 64 lanes of a wave walk runs of equal length:
// with uniform scalars the runs are Poisson (mean n/2^(c-1)), and bucket order costs ~30% of the
// lanes to divergence.  Counting sort over <= 1025 length bins, LDS-privatised; both kernels walk the
// buckets in order (coalesced reads of off / unit_off), one lane per bucket.
// ------------------------------------------------------------------------------------------------
constexpr int MAX_L = 1024;
constexpr uint32_t UNITS_INLINE = 8;   // buckets with more units than this are filled by the whole block

// unit -> bucket map + histogram of unit lengths.  Grid-stride over chunks of 256 buckets: the LDS
// histogram is flushed once per block (one block per chunk meant 10^5 blocks x 257 global atomics on
// the same 257 addresses: 1.3 ms at 2^26, all of it atomic serialisation).
constexpr uint32_t UNIT_GRID = 2048;

__global__ __launch_bounds__(256) void k_fill_units(const uint32_t* __restrict__ off, const uint32_t* __restrict__ unit_off,
                                                    uint64_t G, uint32_t L, uint32_t* __restrict__ unit_bucket,
                                                    uint32_t* __restrict__ hist) {
    __builtin_amdgcn_s_setprio(3);   // sort-stage kernel: may run underneath another task's accumulation (msm_sort3.hip)
    __shared__ uint32_t sh[MAX_L + 1];
    __shared__ uint32_t big[256], nbig;
    for (uint32_t i = threadIdx.x; i <= L; i += 256) sh[i] = 0;
    const uint64_t nchunks = (G + 255) / 256;
    for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        if (threadIdx.x == 0) nbig = 0;
        __syncthreads();
        // a bucket of cnt entries has cnt / L full units and one unit of cnt % L entries
        const uint64_t g0 = chunk * 256;
        uint64_t g = g0 + threadIdx.x;
        if (g < G) {
            uint32_t u0 = unit_off[g], u1 = unit_off[g + 1];
            uint32_t cnt = off[g + 1] - off[g];
            uint32_t nfull = cnt / L, rem = cnt - nfull * L;
            if (nfull) atomicAdd(&sh[L], nfull);
            if (rem) atomicAdd(&sh[rem], 1u);
            if (u1 - u0 > UNITS_INLINE) big[atomicAdd(&nbig, 1u)] = threadIdx.x;   // hot bucket: the block fills it together
            else for (uint32_t u = u0; u < u1; ++u) unit_bucket[u] = (uint32_t)g;
        }
        __syncthreads();
        for (uint32_t b = 0; b < nbig; ++b) {
            uint64_t gb = g0 + big[b];
            uint32_t u0 = unit_off[gb], u1 = unit_off[gb + 1];
            for (uint32_t u = u0 + threadIdx.x; u < u1; u += 256) unit_bucket[u] = (uint32_t)gb;
        }
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i <= L; i += 256)
        if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

// cursor[len] = number of units strictly longer than len (descending order); one block
__global__ __launch_bounds__(256) void k_unit_len_scan(const uint32_t* __restrict__ hist, uint32_t L,
                                                       uint32_t* __restrict__ cursor) {
    __builtin_amdgcn_s_setprio(3);   // sort-stage kernel: may run underneath another task's accumulation (msm_sort3.hip)
    if (threadIdx.x != 0) return;
    uint32_t run = 0;
    for (int len = (int)L; len >= 0; --len) {
        cursor[len] = run;
        run += hist[len];
    }
}

__global__ __launch_bounds__(256) void k_unit_order(const uint32_t* __restrict__ off, const uint32_t* __restrict__ unit_off,
                                                    uint64_t G, uint32_t L, uint32_t* __restrict__ cursor,
                                                    uint32_t* __restrict__ unit_order) {
    __builtin_amdgcn_s_setprio(3);   // sort-stage kernel: may run underneath another task's accumulation (msm_sort3.hip)
    __shared__ uint32_t sh_cnt[MAX_L + 1];
    __shared__ uint32_t sh_base[MAX_L + 1];
    __shared__ uint32_t big[256], big_pos[256], nbig;
    for (uint32_t i = threadIdx.x; i <= L; i += 256) sh_cnt[i] = 0;
    __syncthreads();
    const uint64_t nchunks = (G + 255) / 256;
    // pass 1 over this block's chunks: how many units of each length
    for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        uint64_t g = chunk * 256 + threadIdx.x;
        if (g < G) {
            uint32_t cnt = off[g + 1] - off[g];
            uint32_t nfull = cnt / L, rem = cnt - nfull * L;
            if (nfull) atomicAdd(&sh_cnt[L], nfull);
            if (rem) atomicAdd(&sh_cnt[rem], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= L; i += 256) {
        uint32_t v = sh_cnt[i];
        sh_base[i] = v ? atomicAdd(&cursor[i], v) : 0u;
        sh_cnt[i] = 0;
    }
    // pass 2: place
    for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        if (threadIdx.x == 0) nbig = 0;
        __syncthreads();
        const uint64_t g0 = chunk * 256;
        uint64_t g = g0 + threadIdx.x;
        if (g < G) {
            uint32_t u0 = unit_off[g];
            uint32_t cnt = off[g + 1] - off[g];
            uint32_t nfull = cnt / L, rem = cnt - nfull * L;
            if (nfull) {
                uint32_t pos = sh_base[L] + atomicAdd(&sh_cnt[L], nfull);
                if (nfull > UNITS_INLINE) {
                    uint32_t q = atomicAdd(&nbig, 1u);
                    big[q] = threadIdx.x;
                    big_pos[q] = pos;
                } else {
                    for (uint32_t k = 0; k < nfull; ++k) unit_order[pos + k] = u0 + k;
                }
            }
            if (rem) unit_order[sh_base[rem] + atomicAdd(&sh_cnt[rem], 1u)] = u0 + nfull;
        }
        __syncthreads();
        for (uint32_t b = 0; b < nbig; ++b) {
            uint64_t gb = g0 + big[b];
            uint32_t ub = unit_off[gb];
            uint32_t nf = (off[gb + 1] - off[gb]) / L, pos = big_pos[b];
            for (uint32_t k = threadIdx.x; k < nf; k += 256) unit_order[pos + k] = ub + k;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------

static const MsmCurveOps* ops_for(int curve, int repr = 0) {
    switch (curve) {
        case BLZ_BLS377: return &msm_ops_bls377();
        case BLZ_BLS381: return &msm_ops_bls381();
        case BLZ_BN254: return repr ? &msm_ops_bn254_w32() : &msm_ops_bn254();
    }
    return nullptr;
}

int msm_points_from_mont(int format_id, const void* d_mont, void* d_raw, uint64_t npts, hipStream_t st) {
    const MsmCurveOps* ops = (format_id >> 16) ? nullptr : ops_for(format_id & 0xff, (format_id >> 8) & 0xff);   // (an even-base copy is not the whole table)
    if (!ops) return fail(BLZ_ERR_UNKNOWN, "no conversion back from Montgomery format 0x%x", format_id);
    return ops->points_from_mont(d_mont, d_raw, npts, st);
}
int msm_points_all_canonical(int format_id, const void* d_raw, uint64_t npts, uint32_t* flag, hipStream_t st) {
    const MsmCurveOps* ops = ops_for(format_id & 0xff, (format_id >> 8) & 0xff);
    if (!ops) return fail(BLZ_ERR_UNKNOWN, "unknown Montgomery format 0x%x", format_id);
    return ops->points_all_canonical(d_raw, npts, flag, st);
}

int launch_fill_units(MsmEngine& E, uint32_t U /* upper bound of the unit count */) {
    const uint64_t G = E.last_plan.G;
    const uint32_t L = E.last_plan.L;
    hipStream_t st = E.sort_st;
    BLZ_TRY(E.sb().unit_bucket.reserve(((size_t)U + 1) * 4));
    BLZ_TRY(E.sb().unit_order.reserve(((size_t)U + 1) * 4));
    BLZ_TRY(E.sb().lenhist.reserve(2 * (MAX_L + 1) * 4));
    uint32_t* hist = E.sb().lenhist.as<uint32_t>();
    uint32_t* cursor = hist + (MAX_L + 1);
    BLZ_HIP(hipMemsetAsync(hist, 0, 2 * (MAX_L + 1) * 4, st), BLZ_ERR_UNKNOWN);
    uint64_t nchunks = (G + 255) / 256;
    dim3 gg((uint32_t)(nchunks < UNIT_GRID ? nchunks : UNIT_GRID)), b(256);
    hipLaunchKernelGGL(k_fill_units, gg, b, 0, st, E.sb().off.as<uint32_t>(), E.sb().unit_off.as<uint32_t>(), G, L,
                       E.sb().unit_bucket.as<uint32_t>(), hist);
    hipLaunchKernelGGL(k_unit_len_scan, dim3(1), b, 0, st, hist, L, cursor);
    hipLaunchKernelGGL(k_unit_order, gg, b, 0, st, E.sb().off.as<uint32_t>(), E.sb().unit_off.as<uint32_t>(), G, L, cursor,
                       E.sb().unit_order.as<uint32_t>());
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

int MsmEngine::init(int device_id, int curve_id, int precompute_factor) {
    device = device_id;
    curve = curve_id;
    repr = 0;
    if (curve == BLZ_BN254) {
        // (experiment builds: BLAZE_BN254_REPR = 0 | 1 overrides the choice by precompute factor)
        repr = exp_knob("BLAZE_BN254_REPR", precompute_factor > 1 ? 1 : 0) ? 1 : 0;
    }
    if (!ops_for(curve, repr)) return fail(BLZ_ERR_INVALID_PARAM, "unknown curve %d", curve);
    BLZ_TRY(use_device(device));
    // The runtime multiplexes a process's streams over a few hardware queues PER PRIORITY LEVEL, and which queue a new stream
    // lands on depends on every stream the process ever created or destroyed: a handle opened after another one was closed had
    // its sort stream on the main stream's queue - the "hidden" sort then simply waits for the accumulation it should run
    // beneath (config 3: 91 -> 111 ms per MSM, silently).  Streams of another priority level live on queues of their own, so
    // the streams that must run BESIDE the main stream's long kernels - the hidden sort, the tail (reduce levels, Horner walk:
    // the result of task k is due while task k + 1 accumulates), the exchange / combine - are created at the high level; their
    // kernels are short and already raise their wave priority (s_setprio).  Same-box A/B on the headline: 116.7 / 117.0 ms.
    int prio_low = 0, prio_high = 0;
    BLZ_HIP(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high), BLZ_ERR_UNKNOWN);
    BLZ_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), BLZ_ERR_UNKNOWN);
    BLZ_HIP(hipStreamCreateWithPriority(&tail_stream, hipStreamNonBlocking, prio_high), BLZ_ERR_UNKNOWN);
    BLZ_HIP(hipStreamCreateWithPriority(&aux_stream, hipStreamNonBlocking, prio_high), BLZ_ERR_UNKNOWN);
    BLZ_HIP(hipStreamCreateWithPriority(&sort_stream, hipStreamNonBlocking, prio_high), BLZ_ERR_UNKNOWN);
    sort_st = stream;
    last_sort_done = nullptr;
    for (auto& S : slots) {
        for (auto& e : S.ev) BLZ_HIP(hipEventCreate(&e), BLZ_ERR_UNKNOWN);
        for (auto& e : S.slice_ev) BLZ_HIP(hipEventCreate(&e), BLZ_ERR_UNKNOWN);
        BLZ_HIP(hipEventCreateWithFlags(&S.ev_l0, hipEventDisableTiming), BLZ_ERR_UNKNOWN);
        BLZ_HIP(hipEventCreateWithFlags(&S.ev_done, hipEventDisableTiming), BLZ_ERR_UNKNOWN);
        BLZ_HIP(hipEventCreateWithFlags(&S.ev_sorted, hipEventDisableTiming), BLZ_ERR_UNKNOWN);
        for (int i = 0; i < 2; ++i) {
            BLZ_HIP(hipEventCreateWithFlags(&S.ev_sorted_pp[i], hipEventDisableTiming), BLZ_ERR_UNKNOWN);
            BLZ_HIP(hipEventCreateWithFlags(&S.ev_acc_pp[i], hipEventDisableTiming), BLZ_ERR_UNKNOWN);
        }
        BLZ_HIP(hipEventCreate(&S.ev_s0), BLZ_ERR_UNKNOWN);
        BLZ_HIP(hipEventCreate(&S.ev_s1), BLZ_ERR_UNKNOWN);
        BLZ_HIP(hipHostMalloc((void**)&S.result_h, 256), BLZ_ERR_UNKNOWN);
        BLZ_HIP(hipHostMalloc((void**)&S.stats_h, 64), BLZ_ERR_UNKNOWN);
        memset(S.stats_h, 0, 64);
    }
    BLZ_HIP(hipHostMalloc((void**)&combine_h, 256), BLZ_ERR_UNKNOWN);
    for (auto& B : sbuf) BLZ_TRY(B.stats.reserve(64));
    BLZ_TRY(result.reserve(256 * 64));
    return BLZ_OK;
}

bool MsmEngine::destroy() {
    if (!stream) return true;
    (void)hipSetDevice(device);
    if (sync_stream_bounded(stream, "free: main stream") != BLZ_OK || sync_stream_bounded(tail_stream, "free: tail stream") != BLZ_OK ||
        sync_stream_bounded(sort_stream, "free: sort stream") != BLZ_OK) {
        // work that never completes still references the buffers: freeing them would block (or fault); leak them
        BLZ_LOG(0, "MSM engine freed while its device work is wedged: workspace and streams are leaked");
        stream = tail_stream = aux_stream = sort_stream = nullptr;
        return false;
    }
    for (DevBuf* b : {&coarse, &inter, &inter2, &slice_map, &partial, &blocksums, &result, &sort3_tabs, &bucket_sums, &bucket_ident})
        b->release();
    for (auto& B : sbuf)
        for (DevBuf* b : {&B.count, &B.off, &B.unit_off, &B.unit_bucket, &B.unit_order, &B.lenhist, &B.entries, &B.stats, &B.range_scalars}) b->release();
    for (auto& S : slots) {
        for (DevBuf* b : {&S.lvlA[0], &S.lvlA[1], &S.lvlC[0], &S.lvlC[1]}) b->release();
        for (auto& e : S.ev)
            if (e) (void)hipEventDestroy(e);
        for (auto& e : S.slice_ev)
            if (e) (void)hipEventDestroy(e);
        if (S.ev_l0) (void)hipEventDestroy(S.ev_l0);
        if (S.ev_done) (void)hipEventDestroy(S.ev_done);
        if (S.ev_sorted) (void)hipEventDestroy(S.ev_sorted);
        for (int i = 0; i < 2; ++i) {
            if (S.ev_sorted_pp[i]) (void)hipEventDestroy(S.ev_sorted_pp[i]);
            if (S.ev_acc_pp[i]) (void)hipEventDestroy(S.ev_acc_pp[i]);
        }
        if (S.ev_s0) (void)hipEventDestroy(S.ev_s0);
        if (S.ev_s1) (void)hipEventDestroy(S.ev_s1);
        if (S.result_h) (void)hipHostFree(S.result_h);
        if (S.stats_h) (void)hipHostFree(S.stats_h);
        S = MsmSlot();
    }
    inputs_event = nullptr;
    if (combine_h) (void)hipHostFree(combine_h);
    combine_h = nullptr;
    (void)hipStreamDestroy(stream);
    (void)hipStreamDestroy(tail_stream);
    (void)hipStreamDestroy(aux_stream);
    (void)hipStreamDestroy(sort_stream);
    stream = tail_stream = aux_stream = sort_stream = nullptr;
    last_sort_done = nullptr;
    return true;
}

bool MsmEngine::can_accept() const {
    for (const auto& S : slots)
        if (!S.busy) return true;
    return false;
}

int MsmEngine::sync_all() {
    BLZ_TRY(use_device(device));
    BLZ_TRY(sync_stream_bounded(stream, "reset: main stream"));
    BLZ_TRY(sync_stream_bounded(tail_stream, "reset: tail stream"));
    BLZ_TRY(sync_stream_bounded(aux_stream, "reset: exchange stream"));
    BLZ_TRY(sync_stream_bounded(sort_stream, "reset: sort stream"));
    for (auto& S : slots) { S.busy = false; S.open = false; S.task_inputs_event = nullptr; }
    return BLZ_OK;
}

int MsmEngine::points_to_mont(const void* d_raw, void* d_mont, uint32_t npts) {
    BLZ_TRY(use_device(device));
    return ops_for(curve, repr)->points_to_mont(*this, d_raw, d_mont, npts);
}

int MsmEngine::points_to_mont_even(const void* d_raw, void* d_mont, uint32_t nq) {
    BLZ_TRY(use_device(device));
    return ops_for(curve, repr)->points_to_mont_even(*this, d_raw, d_mont, nq);
}
int MsmEngine::check_precompute(const void* d_raw, uint64_t nelem, uint32_t* flag, hipStream_t st) {
    BLZ_TRY(use_device(device));
    return ops_for(curve, 0)->check_precompute(*this, d_raw, nelem, flag, st);   // (reads wire-format points: always on the faster arithmetic)
}

int MsmEngine::build_table(const void* d_raw, void* d_table, uint32_t npts, int c, int W, int base_shift, void* scratch, uint32_t* flag,
                           hipStream_t st) {
    BLZ_TRY(use_device(device));
    return ops_for(curve, repr)->build_table(*this, d_raw, d_table, npts, c, W, base_shift, scratch, flag, st);
}
size_t MsmEngine::table_scratch_bytes(int W) const { return ops_for(curve, repr)->table_scratch_bytes(W); }

// do the hidden sort's waves (sv VGPRs) fit on a SIMD beside the accumulation's (av VGPRs each, as many as fit)?  Measured: the
// sort runs beside 2 x 200 + 72 = 472 registers and waits for the accumulation to END behind 2 x 208 + 72 = 488.
static bool sort_fits_beside(int av, int sv) {
    const int a = (av + 7) & ~7, s = (sv + 7) & ~7;
    int waves = 512 / a;
    if (waves > 4) waves = 4;
    if (waves < 1) waves = 1;
    return waves * a + s <= 512 - 32;
}

static const int kScalarFieldBits[3] = {253, 255, 254};  // bit length of r (BLS12-377 / 381 / BN254)

MsmPlan MsmEngine::plan_for(uint32_t npts, int sbits) const {
    const int ebits = sbits == 256 ? kScalarFieldBits[curve] : sbits;
    return make_plan(npts, sbits, ebits, plan_override("c", 0));
}

MsmPlan MsmEngine::plan_for_range(uint32_t npts, int bit_lo, int bit_hi) const {
    const int vbits = bit_hi - bit_lo;
    int ebits = kScalarFieldBits[curve] - bit_lo;   // real bits of canonical scalars inside the range
    if (ebits > vbits) ebits = vbits;
    if (ebits < 1) ebits = 1;
    MsmPlan P = make_plan(npts, vbits, ebits, plan_override("c", 0));
    P.base_bit = bit_lo;
    return P;
}

// A task is enqueued in four steps - begin(), then per piece sort_slice() and accumulate_slice(), then end() - so that a
// caller whose inputs arrive over time (msm_stage.hip: host buffers crossing the PCIe link piece by piece, the reference's
// own flow: msm_api.rs:175-202 streams interleaved chunks of scalars and points into the card while it computes) can hand
// each piece to the device when it has landed.  run() is the four steps back to back for inputs that are all there.
//
// A task of ONE piece is the classic pipeline: sort stage (hidden underneath the other slot's accumulation when there is
// one), k_accumulate into per-unit sums, unit folds, bucket reduce over sums[unit_off[g]].  A task of SEVERAL pieces sorts
// and accumulates piece by piece over the same bucket space: the bucket sums live in `bucket_sums`, indexed by bucket;
// k_accumulate_cont resumes a bucket's sum where the previous piece left it (msm_impl.hip.hpp), runs longer than L go through the
// unit folds and k_merge_buckets, and the reduce reads bucket_sums through the identity map.
int MsmEngine::begin(uint32_t npts, int sbits, int* slot_out, int table_c, int bit_lo, int bit_hi, int nslices, bool phased) {
    BLZ_TRY(use_device(device));
    const MsmCurveOps* ops = ops_for(curve, repr);
    hipStream_t st = stream;
    if (npts == 0) return fail(BLZ_ERR_INVALID_PARAM, "begin: empty task");
    // slots are handed out round-robin, so results complete in submission order
    int slot = (cur + 1) % MSM_QUEUE_DEPTH;
    if (slots[slot].busy) slot = (slot + 1) % MSM_QUEUE_DEPTH;
    if (slots[slot].busy) return fail(BLZ_ERR_INVALID_PARAM, "task queue full (%d tasks in flight)", MSM_QUEUE_DEPTH);
    MsmSlot& S = slots[slot];
    const int ebits = sbits == 256 ? kScalarFieldBits[curve] : sbits;
    const bool ranged = bit_hi > bit_lo && !(bit_lo == 0 && bit_hi >= sbits);
    if (ranged && (sbits != 256 || (bit_lo & 31) || (bit_hi & 31) || bit_hi > 256))
        return fail(BLZ_ERR_INVALID_PARAM, "scalar range [%d, %d): 32-bit aligned ranges of 256-bit scalars", bit_lo, bit_hi);
    // (a window table of a ranged handle holds 2^(lo + c j) P: the plan covers hi - lo bits + the carry, no closing doublings)
    MsmPlan P = table_c > 0 ? make_table_plan(npts, table_c, ranged ? bit_hi - bit_lo + 1 : 257)
                : ranged    ? plan_for_range(npts, bit_lo, bit_hi)
                            : make_plan(npts, sbits, ebits, plan_override("c", 0));
    if (P.c == 0) return fail(BLZ_ERR_INVALID_PARAM, "no window plan for npts=%u sbits=%d%s", npts, sbits, table_c > 0 ? " (window table)" : "");
    if (P.table && (sbits != 256 || !msm_sort3t_ok(P))) return fail(BLZ_ERR_INVALID_PARAM, "window-table task outside the sort's range (c=%d)", P.c);
    P.L = (uint32_t)plan_override("L", (int)P.L);
    if (P.L < 1) P.L = 1;
    if (P.L > (uint32_t)MAX_L) P.L = MAX_L;
    const uint64_t G = P.G;
    BLZ_LOG(2, "msm plan: npts=%u sbits=%d c=%d W=%d Bw=%u G=%llu L=%u pieces=%d", npts, sbits, P.c, P.W, P.Bw,
            (unsigned long long)G, P.L, nslices);
    if (nslices < 1 || P.table) nslices = 1;
    if (nslices > MSM_MAX_SLICES) nslices = MSM_MAX_SLICES;
    // pieces of whole 16-point groups (keeps every piece's scalars 16-byte aligned; pf = 8: whole elements)
    // A task whose pieces arrive over the link (phased) ends with HALF a piece: behind the last byte sit that piece's
    // accumulation, the bucket reduce and the tail, and the accumulation is the one part of it that shrinks with the piece
    // (config 2: 13.5 -> 13.0 ms, the lone 2^26 HBM flow 138.6 -> 133); the pieces before it are 1 / (2 k - 1) larger.
    const uint64_t halves = (phased && nslices >= 4) ? 2ull * nslices - 1 : 2ull * nslices;
    uint32_t per = (uint32_t)(((2ull * npts + halves - 1) / halves + 15) & ~(uint64_t)15);
    if (nslices > 1) nslices = (int)(((uint64_t)npts + per - 1) / per);
    const uint64_t max_entries = (uint64_t)(nslices > 1 ? per : npts) * P.W;
    const uint64_t max_units = G + max_entries / P.L + 1;
    if (max_units >= (1ull << 32)) return fail(BLZ_ERR_INVALID_PARAM, "unit bound %llu exceeds 32 bits", (unsigned long long)max_units);

    // The sort stage (digit sort, bucket / unit scans, unit lists) of a one-piece task.  When the handle's other task is
    // still in flight - its accumulation is running or about to - the stage goes to sort_stream with the small-footprint
    // three-level sort (msm_sort3.hip) and runs UNDERNEATH that accumulation; the accumulation of this task then starts
    // with its sort already done.  Its outputs are per slot (SortBufs); the scratch the sorts share is protected by
    // chaining every sort stage behind the one before (last_sort_done).  BLAZE_SORT_HIDE: 0 never, 1 when the other slot
    // is busy (default), 2 the three-level sort always (on the main stream when there is nothing to hide under: tests).
    const int hide_env = env_int("BLAZE_SORT_HIDE", 1);
    const MsmSlot& O = slots[(slot + 1) % MSM_QUEUE_DEPTH];
    bool s3 = P.table || (hide_env != 0 && nslices == 1 && msm_sort3_ok(P, sbits));   // (a table task has no other sort)
    bool fits = true;
    if (s3 && hide_env == 1 && !P.table) {
        // the three-level sort gives one block a whole level-2 bin: fine for the near-uniform digits of real scalars, a
        // cliff for inputs that pile entries into a few buckets (the reference harness repeats a 256-point tile).  The
        // handle's last two COLLECTED tasks say which kind it is being fed (finish(): their largest bucket against their own
        // mean - judged when the host has waited for the task, never read from a copy that may still be in flight)
        if (recent_hot[0] || recent_hot[1]) s3 = false;
    }
    if (s3 && hide_env == 1) {
        // ... and the sort only hides if its waves fit BESIDE the accumulation's: two of those per SIMD (registers are
        // allocated in eights) plus one of the sort's.  Measured: the sort runs beside 2 x 194..199 (-> 2 x 200) + 72 = 472
        // registers, and waits for the accumulation to END behind 2 x 206 (-> 208) + 72 = 488 and 2 x 211 (-> 216) + 72 = 504 -
        // the SIMD does not hand out all 512.  A build whose k_accumulate grew past that still works, it just sorts in
        // the open (round 3 saw 211 VGPRs cost 4 ms per step before this check existed).
        // (the SIMD runs as many accumulation waves as their allocation admits: two of the reduced-radix kernels', three of
        // the 32-bit-limb kernel's, whose allocation is padded to 136 for exactly this purpose)
        const int av = ops->accumulate_vgprs(), sv = P.table ? msm_sort3t_max_vgprs() : msm_sort3_max_vgprs();
        if (av > 0 && sv > 0 && !sort_fits_beside(av, sv)) fits = false;
        if (!fits && !P.table) s3 = false;
    }
    bool hide = s3 && O.busy && hide_env != 0 && fits;
    // pieces of a task on an otherwise idle handle: piece k + 1 sorts underneath the accumulation of piece k (ping-pong over
    // the two slots' sort buffers); same conditions as hiding a task's sort
    bool pingpong = false;
    if (nslices > 1 && hide_env != 0 && !O.busy && !P.table && msm_sort3_ok(P, sbits) && !(recent_hot[0] || recent_hot[1])) {
        const int av = ops->accumulate_vgprs(), sv = msm_sort3_max_vgprs();   // (k_accumulate_cont shares k_accumulate's register cap)
        if (!(av > 0 && sv > 0 && !sort_fits_beside(av, sv))) pingpong = true;
    }
    if (pingpong) { s3 = true; hide = true; }

    cur = slot;
    if (slot_out) *slot_out = slot;
    last_plan = S.plan = P;
    S.npts = npts;
    S.sbits = sbits;
    S.ranged = ranged;
    S.bit_lo = bit_lo;
    S.bit_hi = bit_hi;
    S.slices = nslices;
    S.pts_per_slice = nslices > 1 ? per : npts;
    S.max_units = max_units;
    S.use_s3 = s3 && (hide || hide_env == 2 || P.table);
    S.sort_hidden = hide;
    S.pingpong = pingpong;
    S.acc_pp_recorded[0] = S.acc_pp_recorded[1] = false;
    S.phased = phased;
    S.task_inputs_event = inputs_event;
    S.accum_timed = false;
    BLZ_HIP(hipEventRecord(S.ev[0], st), BLZ_ERR_UNKNOWN);
    for (int b = 0; b < MSM_QUEUE_DEPTH; ++b) {
        if (b != slot && !pingpong) continue;
        BLZ_TRY(sbuf[b].count.reserve((G + 1) * 4 + 16));
        BLZ_TRY(sbuf[b].off.reserve((G + 2) * 4));
        BLZ_TRY(sbuf[b].unit_off.reserve((G + 2) * 4));
        BLZ_TRY(sbuf[b].entries.reserve(max_entries * 4));
    }
    BLZ_TRY(blocksums.reserve((size_t)((G + 1 + SCAN_TILE - 1) / SCAN_TILE) * 8));
    if (nslices > 1) {
        const size_t sum_bytes = (size_t)ops->partial_dwords * 4;
        BLZ_TRY(bucket_sums.reserve((G + 1) * sum_bytes));
        BLZ_TRY(bucket_ident.reserve((G + 2) * 4));
        hipLaunchKernelGGL(k_zero, dim3(2048), dim3(256), 0, st, (uint4*)bucket_sums.p, ((G + 1) * sum_bytes + 15) / 16);   // all-zero = infinity
        BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    }
    S.open = true;
    S.busy = true;
    return BLZ_OK;
}

// the sort stage of piece `sl`: np points whose scalars start at d_scalars (np x sbits / 8 bytes)
int MsmEngine::sort_slice(int slot, int sl, const void* d_scalars, uint32_t np) {
    BLZ_TRY(use_device(device));
    if (slot < 0 || slot >= MSM_QUEUE_DEPTH || !slots[slot].open) return fail(BLZ_ERR_INVALID_PARAM, "slot %d has no task being enqueued", slot);
    MsmSlot& S = slots[slot];
    if (sl < 0 || sl >= S.slices || np == 0 || np > S.pts_per_slice) return fail(BLZ_ERR_INVALID_PARAM, "piece %d of %d with %u points", sl, S.slices, np);
    MsmEngine& E = *this;
    cur = slot;
    last_plan = S.plan;
    const MsmPlan& P = S.plan;
    const uint64_t G = P.G;
    const uint32_t nscan = (uint32_t)((G + 1 + SCAN_TILE - 1) / SCAN_TILE);
    hipStream_t st = stream;
    MsmSlot& O = slots[(slot + 1) % MSM_QUEUE_DEPTH];
    sb_sel = S.pingpong ? (slot + sl) % MSM_QUEUE_DEPTH : slot;
    SortBufs& B = sb();
    const bool hide = S.sort_hidden;
    sort_st = hide ? sort_stream : st;
    hipStream_t ss = sort_st;
    if (last_sort_done) BLZ_HIP(hipStreamWaitEvent(ss, last_sort_done, 0), BLZ_ERR_UNKNOWN);
    if (hide) {
        // whoever read this buffer set last must have let go of it: the accumulation two pieces back (ping-pong), else the
        // slot's previous task (its level-0 reduce read unit_off last) - and the OTHER slot's last task too if that was a
        // ping-pong task, which borrowed this slot's set
        if (S.pingpong && S.acc_pp_recorded[sl & 1]) {
            BLZ_HIP(hipStreamWaitEvent(ss, S.ev_acc_pp[sl & 1], 0), BLZ_ERR_UNKNOWN);
        } else {
            if (S.l0_recorded) BLZ_HIP(hipStreamWaitEvent(ss, S.ev_l0, 0), BLZ_ERR_UNKNOWN);
            if (O.pingpong && O.l0_recorded) BLZ_HIP(hipStreamWaitEvent(ss, O.ev_l0, 0), BLZ_ERR_UNKNOWN);
        }
    }
    BLZ_HIP(hipEventRecord(S.ev_s0, ss), BLZ_ERR_UNKNOWN);   // ev_s0 .. ev_s1: the sort stage, whichever stream it is on
    dim3 b256(256);
    const char* sc_s = (const char*)d_scalars;
    if (S.ranged) {
        // the range of every scalar as a scalar of its own (zero-extended): everything downstream reads 32-byte scalars and
        // walks the plan's windows from bit 0; k_finish puts the 2^bit_lo back (FinishPlan offsets)
        BLZ_TRY(B.range_scalars.reserve((size_t)np * 32 + 16));
        hipLaunchKernelGGL(k_extract_range, dim3(2048), dim3(256), 0, ss, (const uint32_t*)sc_s, B.range_scalars.as<uint32_t>(),
                           (uint64_t)np, (uint32_t)(S.bit_lo >> 5), (uint32_t)((S.bit_hi - S.bit_lo) >> 5));
        sc_s = (const char*)B.range_scalars.p;
    }
    BLZ_HIP(hipMemsetAsync(B.stats.p, 0, 64, ss), BLZ_ERR_UNKNOWN);
    // a small task's whole sort stage - digits, bucket scan, entries, unit lists - is one block's work (msm_sort_tiny.hip)
    const bool tiny = !S.use_s3 && S.slices == 1 && msm_sort_tiny_ok(P, np, S.sbits);
    if (S.use_s3) {
        BLZ_TRY(P.table ? msm_sort3t(E, sc_s, np) : msm_sort3(E, sc_s, np, S.sbits));   // count[] and entries[] in one go
    } else if (tiny) {
        BLZ_TRY(msm_sort_tiny(E, sc_s, np, S.sbits, (uint32_t)S.max_units));
    } else {
        const size_t n16 = ((G + 1) * 4 + 15) / 16;   // the reserve of begin() rounds the allocation up
        hipLaunchKernelGGL(k_zero, dim3(2048), dim3(256), 0, ss, (uint4*)B.count.p, n16);
        BLZ_TRY(msm_sort_lds(E, sc_s, np, S.sbits));
    }
    if (!tiny) {
        hipLaunchKernelGGL(k_scan_reduce, dim3(nscan), b256, 0, ss, B.count.as<uint32_t>(), G, P.L, blocksums.as<uint64_t>(),
                           B.stats.as<uint32_t>());
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_SUMS_THREADS), 0, ss, blocksums.as<uint64_t>(), nscan, B.stats.as<uint32_t>());
        hipLaunchKernelGGL(k_scan_final, dim3(nscan), b256, 0, ss, B.count.as<uint32_t>(), G, P.L, blocksums.as<uint64_t>(),
                           B.off.as<uint32_t>(), B.unit_off.as<uint32_t>());
        if (!S.use_s3) BLZ_TRY(msm_sort_lds_scatter(E));
    }
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    // No host round trip: the unit count stays on the device.  Buffers and grids are sized by the bound
    // (every bucket at most one short unit, plus entries / L full ones) and the kernels read the real count
    // from `stats`; the host copy below is for the log line, the sanity check of finish() and the hot-bucket guard.
    BLZ_TRY(copy_words_to_pinned(S.stats_h, B.stats.p, 4, ss));
    if (!tiny) BLZ_TRY(launch_fill_units(E, (uint32_t)S.max_units));
    hipEvent_t sorted = S.pingpong ? S.ev_sorted_pp[sl & 1] : S.ev_sorted;
    BLZ_HIP(hipEventRecord(sorted, ss), BLZ_ERR_UNKNOWN);
    last_sort_done = sorted;
    BLZ_HIP(hipEventRecord(S.ev_s1, ss), BLZ_ERR_UNKNOWN);
    // run() (inputs all there before the task began): the staged inputs have been consumed once the LAST sort has read
    // the scalars - and, in DMA mode, once the to-Montgomery pass, which msm_stage.hip enqueued on the MAIN stream ahead of
    // run(), has read the raw points (ev[0] sits behind it): only then may a later task's host -> device copies overwrite
    // this staging set (msm_stage.hip's copy stream waits for the event).  On the sort stream the wait comes last, behind
    // ev_sorted, so it delays nothing but the event.  (Phased tasks: end() records it.)
    if (!S.phased && S.task_inputs_event && sl + 1 == S.slices) {
        if (hide) BLZ_HIP(hipStreamWaitEvent(ss, S.ev[0], 0), BLZ_ERR_UNKNOWN);
        BLZ_HIP(hipEventRecord(S.task_inputs_event, ss), BLZ_ERR_UNKNOWN);
        S.task_inputs_event = nullptr;
    }
    return BLZ_OK;
}

// the accumulation of piece `sl` (its sort stage has been enqueued): d_pts = Montgomery points of the piece's first point
int MsmEngine::accumulate_slice(int slot, int sl, const void* d_pts) {
    BLZ_TRY(use_device(device));
    if (slot < 0 || slot >= MSM_QUEUE_DEPTH || !slots[slot].open) return fail(BLZ_ERR_INVALID_PARAM, "slot %d has no task being enqueued", slot);
    MsmSlot& S = slots[slot];
    const MsmCurveOps* ops = ops_for(curve, repr);
    cur = slot;
    last_plan = S.plan;
    sort_st = S.sort_hidden ? sort_stream : stream;
    sb_sel = S.pingpong ? (slot + sl) % MSM_QUEUE_DEPTH : slot;
    if (S.sort_hidden) BLZ_HIP(hipStreamWaitEvent(stream, S.pingpong ? S.ev_sorted_pp[sl & 1] : S.ev_sorted, 0), BLZ_ERR_UNKNOWN);
    BLZ_TRY(ops->run_accumulate(*this, d_pts, (uint32_t)S.max_units, S.slices > 1 ? sl : -1));
    if (S.slices > 1) BLZ_TRY(ops->merge_buckets(*this));
    if (S.pingpong) {
        BLZ_HIP(hipEventRecord(S.ev_acc_pp[sl & 1], stream), BLZ_ERR_UNKNOWN);
        S.acc_pp_recorded[sl & 1] = true;
    }
    return BLZ_OK;
}

// every piece is enqueued: bucket reduce + tail
int MsmEngine::end(int slot) {
    BLZ_TRY(use_device(device));
    if (slot < 0 || slot >= MSM_QUEUE_DEPTH || !slots[slot].open) return fail(BLZ_ERR_INVALID_PARAM, "slot %d has no task being enqueued", slot);
    MsmSlot& S = slots[slot];
    const MsmCurveOps* ops = ops_for(curve, repr);
    cur = slot;
    last_plan = S.plan;
    if (S.task_inputs_event) {
        // phased task: every reader of the staged inputs - the pieces' sorts and the caller's to-Montgomery passes - is on
        // the main stream or has been waited for by it
        BLZ_HIP(hipEventRecord(S.task_inputs_event, stream), BLZ_ERR_UNKNOWN);
        S.task_inputs_event = nullptr;
    }
    sb_sel = slot;
    if (S.slices > 1) {
        hipLaunchKernelGGL(k_iota, dim3(1024), dim3(256), 0, stream, bucket_ident.as<uint32_t>(), S.plan.G + 2);
        BLZ_TRY(ops->run_reduce(*this, bucket_sums.p, bucket_ident.p));
    } else {
        BLZ_TRY(ops->run_reduce(*this, partial.p, sb().unit_off.p));
    }
    S.l0_recorded = true;
    S.open = false;
    return BLZ_OK;
}

// give up a task whose inputs never arrived in full (a copy failed): what was enqueued runs to completion and is ignored
void MsmEngine::abandon(int slot) {
    if (slot < 0 || slot >= MSM_QUEUE_DEPTH) return;
    MsmSlot& S = slots[slot];
    if (S.busy && S.open) {
        S.busy = false;
        S.open = false;
        S.task_inputs_event = nullptr;
    }
}

int MsmEngine::run(const void* d_pts, const void* d_scalars, uint32_t npts, int sbits, int* slot_out, int table_c, int bit_lo,
                   int bit_hi) {
    BLZ_TRY(use_device(device));
    if (npts == 0) {
        const MsmCurveOps* ops = ops_for(curve, repr);
        hipStream_t st = stream;
        int slot = (cur + 1) % MSM_QUEUE_DEPTH;
        if (slots[slot].busy) slot = (slot + 1) % MSM_QUEUE_DEPTH;
        if (slots[slot].busy) return fail(BLZ_ERR_INVALID_PARAM, "task queue full (%d tasks in flight)", MSM_QUEUE_DEPTH);
        cur = slot;
        MsmSlot& S = slots[slot];
        if (slot_out) *slot_out = slot;
        BLZ_HIP(hipEventRecord(S.ev[0], st), BLZ_ERR_UNKNOWN);
        BLZ_TRY(ops->emit_infinity(*this));
        for (int i = 1; i <= 4; ++i) BLZ_HIP(hipEventRecord(S.ev[i], st), BLZ_ERR_UNKNOWN);
        BLZ_TRY(copy_words_to_pinned(S.result_h, slot_result(slot), (uint32_t)(3 * fq_bytes(curve) / 4), st));
        BLZ_HIP(hipEventRecord(S.ev_done, st), BLZ_ERR_UNKNOWN);
        last_plan = S.plan = MsmPlan();
        S.accum_timed = false;
        S.open = false;
        S.busy = true;
        return BLZ_OK;
    }
    // resident inputs: one piece (BLAZE_MSM_PIECES = n forces n: tests of the piecewise path at sizes the oracle checks)
    int pieces = env_int("BLAZE_MSM_PIECES", 1);
    int slot = -1;
    BLZ_TRY(begin(npts, sbits, &slot, table_c, bit_lo, bit_hi, pieces, false));
    if (slot_out) *slot_out = slot;
    MsmSlot& S = slots[slot];
    int rc = BLZ_OK;
    for (int sl = 0; sl < S.slices && rc == BLZ_OK; ++sl) {
        const uint32_t p0 = (uint32_t)sl * S.pts_per_slice;
        const uint32_t np = npts - p0 < S.pts_per_slice ? npts - p0 : S.pts_per_slice;
        rc = sort_slice(slot, sl, (const char*)d_scalars + (size_t)p0 * (sbits / 8), np);
        if (rc == BLZ_OK) rc = accumulate_slice(slot, sl, (const char*)d_pts + (size_t)p0 * mont_point_bytes(curve));
    }
    if (rc == BLZ_OK) rc = end(slot);
    if (rc != BLZ_OK) abandon(slot);
    return rc;
}

int MsmEngine::finish(int slot, uint8_t* out) {
    BLZ_TRY(use_device(device));
    if (slot < 0 || slot >= MSM_QUEUE_DEPTH || !slots[slot].busy) return fail(BLZ_ERR_INVALID_PARAM, "no task in slot %d", slot);
    if (slots[slot].open) return fail(BLZ_ERR_INVALID_PARAM, "task in slot %d never received all of its data", slot);
    MsmSlot& S = slots[slot];
    // bounded: a wedged kernel must not hang the host for ever (common.hpp); on expiry the slot stays busy
    BLZ_TRY(sync_event_bounded(S.ev_done, "wait_result: MSM task"));
    S.busy = false;
    if (S.plan.c) {
        BLZ_LOG(2, "msm: units=%u max_bucket=%u entries=%u", S.stats_h[0], S.stats_h[1], S.stats_h[2]);
        if ((uint64_t)S.stats_h[0] > S.max_units) return fail(BLZ_ERR_UNKNOWN, "unit count %u exceeds its bound", S.stats_h[0]);
        // hot-bucket guard of begin(): this task's largest bucket against the mean of the sort that produced it
        const uint64_t mean = (uint64_t)S.stats_h[2] / (S.plan.G ? S.plan.G : 1) + 1;
        recent_hot[1] = recent_hot[0];
        recent_hot[0] = (uint64_t)S.stats_h[1] > 64 * mean + 4096;
    }
    memcpy(out, S.result_h, 3 * fq_bytes(curve));
    float t = 0;
    (void)hipEventElapsedTime(&t, S.ev[0], S.ev[4]); last_ms[0] = t;
    last_ms[1] = 0;
    if (S.accum_timed && S.plan.c) {
        if (S.slices > 1) {   // piecewise task: the sum of the pieces' k_accumulate_cont launches
            for (int i = 0; i < S.slices; ++i)
                if (hipEventElapsedTime(&t, S.slice_ev[2 * i], S.slice_ev[2 * i + 1]) == hipSuccess) last_ms[1] += t;
        } else {
            (void)hipEventElapsedTime(&t, S.ev[5], S.ev[6]);
            last_ms[1] = t;
        }
    }
    last_sort_hidden = S.sort_hidden && S.plan.c != 0;
    if (S.plan.c) (void)hipEventElapsedTime(&t, S.ev_s0, S.ev_s1);   // the sort stage: on the main stream, or on sort_stream underneath the previous task
    else t = 0;
    last_ms[2] = t;
    (void)hipEventElapsedTime(&t, S.ev[1], S.ev[2]); last_ms[3] = t;
    (void)hipEventElapsedTime(&t, S.ev[2], S.ev[3]); last_ms[4] = t;
    (void)hipEventElapsedTime(&t, S.ev[3], S.ev[4]); last_ms[5] = t;
    last_ms[6] = (float)S.plan.c;
    last_ms[7] = (float)S.plan.W;
    return BLZ_OK;
}

int MsmEngine::combine_partials(const uint8_t* partials, size_t cnt, uint8_t* out, bool on_device) {
    BLZ_TRY(use_device(device));
    return ops_for(curve, repr)->combine(*this, partials, cnt, out, on_device);
}

}  // namespace blz
