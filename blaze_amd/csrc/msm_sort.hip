// Digit sort of the MSM pipeline: group (point index | sign) entries by (window, bucket).
//
// Two-level LDS-privatised counting sort (replaces one global atomic per entry, which ran at
// ~25 G atomics/s and cost 105 ms of a 2^26 MSM):
//   bucket index of a window = coarse bin | fine (cl bits, 11 by default); windows may differ in width
//   (MsmPlan), so window w owns coarse bins binoff[w] .. binoff[w+1]; G >> cl <= 24576 bins in all
//   k_coarse_count    per block: LDS histogram of its points' digits over all coarse bins, flushed
//                     with one global atomic per (block, non-empty bin)
//   k_coarse_scan     exclusive scan over the coarse bins (one block)
//   k_coarse_scatter_staged   16 scalars per lane in registers, window by window: LDS rank, bin-major
//                     LDS stage, slot-major copy-out of (entry, fine) pairs grouped by coarse bin
//                     (k_coarse_scatter: the unstaged form, BLAZE_SORT_STAGED=0)
//   k_slice_map       device-built work list: every coarse bin cut into slices of <= 65536 entries
//   k_fine_count      per (coarse bin, slice): LDS histogram over the 2^cl fine buckets -> count[]
//   (k_scan_* of msm.hip: bucket offsets + unit offsets)
//   k_fine_scatter    per (coarse bin, slice): LDS rank, reservation per (block, bucket) on the
//                     scatter cursor, bucket-major LDS stage, slot-major copy-out to the final runs
// HBM traffic: scalars read twice (32 B each), 8 B/entry written + read twice, 4 B/entry written.
#include "msm_engine.hpp"
#include "msm_digits.hip.hpp"

namespace blz {

// wave priority of the sort kernels (0..3): raised when they are meant to run underneath another kernel
#ifndef BLZ_SORT_PRIO
#define BLZ_SORT_PRIO 0
#endif
#define BLZ_SORT_SETPRIO() do { if (BLZ_SORT_PRIO) __builtin_amdgcn_s_setprio(BLZ_SORT_PRIO); } while (0)

#ifndef BLZ_SORT_THREADS
#define BLZ_SORT_THREADS 1024
#endif
constexpr int SORT_THREADS = BLZ_SORT_THREADS;
constexpr int FINE_THREADS = 512;
constexpr int SORT_UNROLL = 4;

struct SortGeom {
    int W, cl;                      // windows; fine bits of the bucket index (the same for every window)
    int chmax;                      // log2 of the largest window's coarse-bin count
    uint32_t NC;                    // total coarse bins = G >> cl
    uint32_t pts_per_block;
    uint8_t width[MSM_MAX_W];       // window widths (MsmPlan)
    uint32_t binoff[MSM_MAX_W + 1]; // first coarse bin of window w
};

// signed digit of window w (pops width[w] bits); carry runs through the windows of one scalar
template <int SW>
__device__ __forceinline__ int next_digit(ScalarWords<SW>& sw, const SortGeom& g, int w, uint32_t& carry) {
    const int cw = g.width[w];
    return sw.next(cw, (1u << cw) - 1u, 1u << (cw - 1), carry);
}

template <int SW>
__global__ __launch_bounds__(SORT_THREADS) void k_coarse_count(const uint32_t* __restrict__ scalars, uint32_t npts, SortGeom g,
                                                               uint32_t* __restrict__ coarse_count) {
    BLZ_SORT_SETPRIO();
    extern __shared__ __attribute__((aligned(16))) uint32_t sh[];
    for (uint32_t i = threadIdx.x; i < g.NC; i += SORT_THREADS) sh[i] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * g.pts_per_block;
    uint32_t end = base + g.pts_per_block;
    if (end > npts) end = npts;
    // SORT_UNROLL scalars in flight per lane before their digits are consumed (latency, not bandwidth,
    // bounds the one-scalar-per-iteration form)
    for (uint32_t p0 = base + threadIdx.x; p0 < end; p0 += SORT_UNROLL * SORT_THREADS) {
        ScalarWords<SW> sw[SORT_UNROLL];
#pragma unroll
        for (int u = 0; u < SORT_UNROLL; ++u) {
            uint32_t p = p0 + u * SORT_THREADS;
            sw[u].load(scalars, p < end ? p : p0);
        }
#pragma unroll
        for (int u = 0; u < SORT_UNROLL; ++u) {
            if (p0 + u * SORT_THREADS >= end) break;
            uint32_t carry = 0;
            for (int w = 0; w < g.W; ++w) {
                int d = next_digit(sw[u], g, w, carry);
                if (d != 0) {
                    uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
                    atomicAdd(&sh[g.binoff[w] + (b >> g.cl)], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < g.NC; i += SORT_THREADS) {
        uint32_t v = sh[i];
        if (v) atomicAdd(&coarse_count[i], v);
    }
}

// exclusive scan of coarse_count[NC] -> coarse_off[NC+1]; coarse_count becomes the scatter cursor
__global__ __launch_bounds__(1024) void k_coarse_scan(uint32_t* __restrict__ coarse_count, uint32_t NC,
                                                      uint32_t* __restrict__ coarse_off) {
    BLZ_SORT_SETPRIO();
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t carry_sh;
    if (threadIdx.x == 0) carry_sh = 0;
    __syncthreads();
    for (uint32_t base = 0; base < NC; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < NC ? coarse_count[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            uint32_t t = threadIdx.x >= (uint32_t)o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        uint32_t incl = sh[threadIdx.x], carry = carry_sh;
        if (i < NC) {
            coarse_off[i] = carry + incl - v;
            coarse_count[i] = carry + incl - v;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_sh = carry + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) coarse_off[NC] = carry_sh;
}

template <int SW>
__global__ __launch_bounds__(SORT_THREADS) void k_coarse_scatter(const uint32_t* __restrict__ scalars, uint32_t npts, SortGeom g,
                                                                 uint32_t* __restrict__ coarse_cursor,
                                                                 uint32_t* __restrict__ inter_idx,
                                                                 uint16_t* __restrict__ inter_fine) {
    BLZ_SORT_SETPRIO();
    extern __shared__ __attribute__((aligned(16))) uint32_t sh[];
    for (uint32_t i = threadIdx.x; i < g.NC; i += SORT_THREADS) sh[i] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * g.pts_per_block;
    uint32_t end = base + g.pts_per_block;
    if (end > npts) end = npts;
    // SORT_UNROLL scalars in flight per lane before their digits are consumed (latency, not bandwidth,
    // bounds the one-scalar-per-iteration form)
    for (uint32_t p0 = base + threadIdx.x; p0 < end; p0 += SORT_UNROLL * SORT_THREADS) {
        ScalarWords<SW> sw[SORT_UNROLL];
#pragma unroll
        for (int u = 0; u < SORT_UNROLL; ++u) {
            uint32_t p = p0 + u * SORT_THREADS;
            sw[u].load(scalars, p < end ? p : p0);
        }
#pragma unroll
        for (int u = 0; u < SORT_UNROLL; ++u) {
            if (p0 + u * SORT_THREADS >= end) break;
            uint32_t carry = 0;
            for (int w = 0; w < g.W; ++w) {
                int d = next_digit(sw[u], g, w, carry);
                if (d != 0) {
                    uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
                    atomicAdd(&sh[g.binoff[w] + (b >> g.cl)], 1u);
                }
            }
        }
    }
    __syncthreads();
    // one global reservation per (block, non-empty bin); the LDS counter becomes the write cursor
    for (uint32_t i = threadIdx.x; i < g.NC; i += SORT_THREADS) {
        uint32_t v = sh[i];
        sh[i] = v ? atomicAdd(&coarse_cursor[i], v) : 0u;
    }
    __syncthreads();
    const uint32_t fmask = (1u << g.cl) - 1u;
    for (uint32_t p0 = base + threadIdx.x; p0 < end; p0 += SORT_UNROLL * SORT_THREADS) {
        ScalarWords<SW> sw[SORT_UNROLL];
#pragma unroll
        for (int u = 0; u < SORT_UNROLL; ++u) {
            uint32_t p = p0 + u * SORT_THREADS;
            sw[u].load(scalars, p < end ? p : p0);
        }
#pragma unroll
        for (int u = 0; u < SORT_UNROLL; ++u) {
            const uint32_t p = p0 + u * SORT_THREADS;
            if (p >= end) break;
            uint32_t carry = 0;
            for (int w = 0; w < g.W; ++w) {
                int d = next_digit(sw[u], g, w, carry);
                if (d != 0) {
                    uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
                    uint32_t pos = atomicAdd(&sh[g.binoff[w] + (b >> g.cl)], 1u);
                    inter_idx[pos] = p | (d < 0 ? 0x80000000u : 0u);
                    inter_fine[pos] = (uint16_t)(b & fmask);
                }
            }
        }
    }
}

// Coarse scatter, window at a time, LDS-staged.  The kernel above issues one 8-byte store per entry to
// W * 2^ch scattered cursors, and the L2 takes ~100 G such requests per second whatever their size:
// 8.2 of its 9.6 ms at 2^26 (with the stores removed it runs in 1.4 ms).  Here a block keeps CS_T
// scalars per lane in registers and walks the windows: per window its entries are ranked with an LDS
// histogram over the window's 2^ch bins, laid out bin-major in an LDS stage, and copied out slot-major,
// so consecutive lanes write consecutive addresses of a bin's run (8 entries = 64 B on average).
#ifndef BLZ_CS_THREADS
#define BLZ_CS_THREADS 512
#endif
#ifndef BLZ_CS_T
#define BLZ_CS_T 16
#endif
constexpr int CS_THREADS = BLZ_CS_THREADS;
constexpr int CS_T = BLZ_CS_T;
// (CS_THREADS * CS_T = 8192 points per block: 64 KiB of staging)
constexpr int CS_T_SMALL = 2;             // inputs of up to CS_SMALL_PTS points: 1024 per block
constexpr uint32_t CS_SMALL_PTS = 1u << 19;

// (T scalars per lane: 16 where the input fills the chip with blocks of 8192 points, 2 below that - 2^16 points were 8 blocks
// walking 22 windows for 0.38 ms of a 2.4 ms MSM)
template <int SW, int T>
__global__ __launch_bounds__(CS_THREADS) void k_coarse_scatter_staged(const uint32_t* __restrict__ scalars, uint32_t npts,
                                                                      SortGeom g, uint32_t* __restrict__ coarse_cursor,
                                                                      uint32_t* __restrict__ inter_idx,
                                                                      uint16_t* __restrict__ inter_fine) {
    BLZ_SORT_SETPRIO();
    extern __shared__ __attribute__((aligned(16))) uint32_t sh[];
    const uint32_t nbmax = 1u << g.chmax;
    uint32_t* hist = sh;              // [nb] counts of this window
    uint32_t* lstart = sh + nbmax;       // [nb] first stage slot of the bin
    uint32_t* gbase = sh + 2 * nbmax;    // [nb] reserved global position of the bin's run
    uint2* stage = reinterpret_cast<uint2*>(sh + 3 * nbmax);  // [CS_THREADS * T]
    __shared__ uint32_t wave_tot[CS_THREADS / 64];
    __shared__ uint32_t total_sh;
    const uint32_t tid = threadIdx.x;
    const uint32_t base = blockIdx.x * (uint32_t)(CS_THREADS * T);
    const uint32_t fmask = (1u << g.cl) - 1u;
    ScalarWords<SW> sw[T];
    uint32_t carry[T];
#pragma unroll
    for (int u = 0; u < T; ++u) {
        uint32_t p = base + u * CS_THREADS + tid;
        sw[u].load(scalars, p < npts ? p : 0u);
        carry[u] = 0;
    }
    for (int w = 0; w < g.W; ++w) {
        const uint32_t nb = g.binoff[w + 1] - g.binoff[w];                     // bins of this window (a power of two)
        const uint32_t per = nb > (uint32_t)CS_THREADS ? nb / CS_THREADS : 1u;  // bins per lane in the scan
        for (uint32_t i = tid; i < nb; i += CS_THREADS) hist[i] = 0;
        __syncthreads();
        uint32_t key[T], rk[T];  // key = fine | bin << 12 | sign << 31;  rk = rank in the bin, ~0 = no entry
#pragma unroll
        for (int u = 0; u < T; ++u) {
            int d = next_digit(sw[u], g, w, carry[u]);
            rk[u] = ~0u;
            key[u] = 0;
            if (d != 0 && base + u * CS_THREADS + tid < npts) {
                uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
                uint32_t bin = b >> g.cl;
                key[u] = (b & fmask) | (bin << 12) | (d < 0 ? 0x80000000u : 0u);
                rk[u] = atomicAdd(&hist[bin], 1u);
            }
        }
        __syncthreads();
        {   // exclusive scan of hist + one global reservation per non-empty bin
            const uint32_t b0 = tid * per;
            uint32_t loc[4];
            uint32_t sum = 0;
            for (uint32_t q = 0; q < per; ++q) {
                uint32_t v = (b0 + q) < nb ? hist[b0 + q] : 0u;
                loc[q] = v;
                sum += v;
            }
            uint32_t incl = sum;
            for (int o = 1; o < 64; o <<= 1) {
                uint32_t t2 = __shfl_up(incl, o, 64);
                if ((tid & 63u) >= (uint32_t)o) incl += t2;
            }
            if ((tid & 63u) == 63u) wave_tot[tid >> 6] = incl;
            __syncthreads();
            uint32_t wbase = 0;
            for (uint32_t q = 0; q < (tid >> 6); ++q) wbase += wave_tot[q];
            uint32_t run = wbase + incl - sum;
            for (uint32_t q = 0; q < per; ++q) {
                if ((b0 + q) < nb) {
                    uint32_t v = loc[q];
                    lstart[b0 + q] = run;
                    gbase[b0 + q] = v ? atomicAdd(&coarse_cursor[g.binoff[w] + b0 + q], v) : 0u;
                    run += v;
                }
            }
            if (tid == CS_THREADS - 1) total_sh = wbase + incl;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < T; ++u) {
            if (rk[u] != ~0u) {
                uint32_t bin = (key[u] >> 12) & 0x7ffffu;
                uint32_t p = base + u * CS_THREADS + tid;
                stage[lstart[bin] + rk[u]] = make_uint2(p | (key[u] & 0x80000000u), (key[u] & 0xfffu) | (bin << 16));
            }
        }
        __syncthreads();
        const uint32_t total = total_sh;
        for (uint32_t slot = tid; slot < total; slot += CS_THREADS) {
            uint2 e = stage[slot];
            uint32_t bin = e.y >> 16;
            const uint32_t dst = gbase[bin] + (slot - lstart[bin]);
            inter_idx[dst] = e.x;
            inter_fine[dst] = (uint16_t)e.y;
        }
        __syncthreads();
    }
}

// Work list of the fine passes: coarse bin k is cut into ceil(size_k / SLICE) slices, so a bin that
// holds far more than the mean (the short top window puts 1/W of all entries into a handful of bins;
// the reference harness's repeated tile does the same everywhere) is spread over many blocks.  Built on
// the device (one block), no host round trip: the launch uses the bound entries/SLICE + NC.
constexpr uint32_t SLICE = 65536;

#ifndef BLZ_SLICEMAP_THREADS
#define BLZ_SLICEMAP_THREADS 1024
#endif
constexpr int SLICEMAP_THREADS = BLZ_SLICEMAP_THREADS;
__global__ __launch_bounds__(SLICEMAP_THREADS) void k_slice_map(const uint32_t* __restrict__ coarse_off, uint32_t NC,
                                                    uint2* __restrict__ slice_map, uint32_t* __restrict__ nslices) {
    BLZ_SORT_SETPRIO();
    __shared__ uint32_t sh[SLICEMAP_THREADS];
    __shared__ uint32_t carry_sh;
    if (threadIdx.x == 0) carry_sh = 0;
    __syncthreads();
    for (uint32_t base = 0; base < NC; base += SLICEMAP_THREADS) {
        uint32_t i = base + threadIdx.x;
        uint32_t size = i < NC ? coarse_off[i + 1] - coarse_off[i] : 0;
        uint32_t v = (size + SLICE - 1) / SLICE;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < SLICEMAP_THREADS; o <<= 1) {
            uint32_t t = threadIdx.x >= (uint32_t)o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        uint32_t incl = sh[threadIdx.x], carry = carry_sh;
        uint32_t first = carry + incl - v;
        for (uint32_t j = 0; j < v; ++j) slice_map[first + j] = make_uint2(i, j);
        __syncthreads();
        if (threadIdx.x == SLICEMAP_THREADS - 1) carry_sh = carry + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *nslices = carry_sh;
}

// entries [lo, hi) of work item `sid`; multi = the bin has other slices too
__device__ __forceinline__ bool slice_range(const uint32_t* coarse_off, const uint2* slice_map, const uint32_t* nslices,
                                            uint32_t sid, uint32_t& k, uint32_t& lo, uint32_t& hi, bool& multi) {
    if (sid >= *nslices) return false;
    uint2 m = slice_map[sid];
    k = m.x;
    uint32_t a = coarse_off[k], b = coarse_off[k + 1];
    lo = a + m.y * SLICE;
    hi = lo + SLICE < b ? lo + SLICE : b;
    multi = b - a > SLICE;
    return lo < hi;
}

__global__ __launch_bounds__(FINE_THREADS) void k_fine_count(const uint16_t* __restrict__ inter_fine, const uint32_t* __restrict__ coarse_off,
                                                             const uint2* __restrict__ slice_map, const uint32_t* __restrict__ nslices,
                                                             int cl, uint32_t* __restrict__ count) {
    BLZ_SORT_SETPRIO();
    extern __shared__ __attribute__((aligned(16))) uint32_t sh[];
    const uint32_t nf = 1u << cl;
    uint32_t k, lo, hi;
    bool multi;
    if (!slice_range(coarse_off, slice_map, nslices, blockIdx.x, k, lo, hi, multi)) return;
    for (uint32_t i = threadIdx.x; i < nf; i += FINE_THREADS) sh[i] = 0;
    __syncthreads();
    // 8 independent loads in flight per lane before the dependent LDS atomics (the one-load-per-
    // iteration form was HBM-latency bound: 1.2 TB/s)
    uint32_t j = lo + threadIdx.x;
    for (; j + 7 * FINE_THREADS < hi; j += 8 * FINE_THREADS) {
        uint32_t key[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) key[u] = inter_fine[j + u * FINE_THREADS];
#pragma unroll
        for (int u = 0; u < 8; ++u) atomicAdd(&sh[key[u]], 1u);
    }
    for (; j < hi; j += FINE_THREADS) atomicAdd(&sh[inter_fine[j]], 1u);
    __syncthreads();
    uint32_t* dst = count + ((size_t)k << cl);
    for (uint32_t i = threadIdx.x; i < nf; i += FINE_THREADS) {
        uint32_t v = sh[i];
        if (v) {
            if (!multi) dst[i] = v;
            else atomicAdd(&dst[i], v);
        }
    }
}

// Final scatter, LDS-staged: a block takes its slice in rounds of up to FS_ROUND entries held in
// registers; ranks come from an LDS histogram, the round is laid out bucket-major in an LDS stage
// (payload + 16-bit bucket id per slot) and copied out slot-major: consecutive lanes write consecutive
// entries of a bucket's run, and neighbouring buckets' runs are neighbours in memory too, so a store
// instruction touches a handful of lines instead of 64 (one-lane-per-bucket copy-out: 5.0 ms; the
// unstaged version wrote 17.6 GB for 3.2 GB of entries in the WRITE_SIZE counter).
#ifndef BLZ_FS_THREADS
#define BLZ_FS_THREADS 1024
#endif
#ifndef BLZ_FS_PER_THREAD
#define BLZ_FS_PER_THREAD 24
#endif
constexpr int FS_THREADS = BLZ_FS_THREADS;
constexpr int FS_PER_THREAD = BLZ_FS_PER_THREAD;
constexpr int FS_ROUND = FS_THREADS * FS_PER_THREAD;  // up to 24576 entries: 6 bytes of staging each

__global__ __launch_bounds__(FS_THREADS) void k_fine_scatter(const uint32_t* __restrict__ inter_idx, const uint16_t* __restrict__ inter_fine,
                                                             const uint32_t* __restrict__ coarse_off,
                                                             const uint2* __restrict__ slice_map, const uint32_t* __restrict__ nslices,
                                                             int cl, uint32_t round_cap,
                                                             uint32_t* __restrict__ cursor, uint32_t* __restrict__ entries) {
    BLZ_SORT_SETPRIO();
    extern __shared__ __attribute__((aligned(16))) uint32_t sh[];
    const uint32_t nf = 1u << cl;
    uint32_t* hist = sh;             // [nf]  counts, then first stage slot of the bucket
    uint32_t* gbase = sh + nf;       // [nf]  reserved global position of the bucket's run
    uint32_t* stage = sh + 2 * nf;   // [round_cap] payloads
    uint16_t* ids = reinterpret_cast<uint16_t*>(stage + round_cap);  // [round_cap] bucket of each slot
    __shared__ uint32_t wave_tot[FS_THREADS / 64];
    uint32_t k, lo, hi;
    bool multi;
    if (!slice_range(coarse_off, slice_map, nslices, blockIdx.x, k, lo, hi, multi)) return;
    uint32_t* cur = cursor + ((size_t)k << cl);
    const uint32_t per_thr_bins = (nf + FS_THREADS - 1) / FS_THREADS;
    for (uint32_t r0 = lo; r0 < hi; r0 += round_cap) {
        const uint32_t rn = (hi - r0) < round_cap ? (hi - r0) : round_cap;
        for (uint32_t i = threadIdx.x; i < nf; i += FS_THREADS) hist[i] = 0;
        __syncthreads();
        uint32_t ex[FS_PER_THREAD], ey[FS_PER_THREAD];
#pragma unroll
        for (int t = 0; t < FS_PER_THREAD; ++t) {
            uint32_t j = t * FS_THREADS + threadIdx.x;
            if (j < rn) {
                const uint32_t f = inter_fine[r0 + j];
                uint32_t rank = atomicAdd(&hist[f], 1u);
                ex[t] = inter_idx[r0 + j];
                ey[t] = f | (rank << 12);
            }
        }
        __syncthreads();
        // exclusive scan of hist (each lane owns per_thr_bins consecutive buckets) + global reservation
        {
            uint32_t b0 = threadIdx.x * per_thr_bins;
            uint32_t sum = 0;
            for (uint32_t q = 0; q < per_thr_bins; ++q) sum += (b0 + q) < nf ? hist[b0 + q] : 0;
            uint32_t incl = sum;
            for (int o = 1; o < 64; o <<= 1) {
                uint32_t t2 = __shfl_up(incl, o, 64);
                if ((threadIdx.x & 63) >= (uint32_t)o) incl += t2;
            }
            if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
            __syncthreads();
            uint32_t wbase = 0;
            for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) wbase += wave_tot[w];
            uint32_t run = wbase + incl - sum;
            for (uint32_t q = 0; q < per_thr_bins; ++q) {
                if ((b0 + q) < nf) {
                    uint32_t v = hist[b0 + q];   // only this lane touches its buckets between the barriers
                    hist[b0 + q] = run;
                    gbase[b0 + q] = v ? atomicAdd(&cur[b0 + q], v) : 0u;
                    run += v;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < FS_PER_THREAD; ++t) {
            uint32_t j = t * FS_THREADS + threadIdx.x;
            if (j < rn) {
                uint32_t f = ey[t] & 0xfffu, slot = hist[f] + (ey[t] >> 12);
                stage[slot] = ex[t];
                ids[slot] = (uint16_t)f;
            }
        }
        __syncthreads();
        for (uint32_t slot = threadIdx.x; slot < rn; slot += FS_THREADS) {
            uint32_t f = ids[slot];
            entries[gbase[f] + (slot - hist[f])] = stage[slot];
        }
        __syncthreads();
    }
}

// coarse / fine split of the bucket index
static SortGeom make_geom(const MsmPlan& P) {
    SortGeom g;
    g.W = P.W;
    int cmin = 32, cmax = 0;
    for (int w = 0; w < P.W; ++w) {
        cmin = P.width[w] < cmin ? P.width[w] : cmin;
        cmax = P.width[w] > cmax ? P.width[w] : cmax;
    }
    const int cl_pref = exp_knob("BLAZE_SORT_CL", 11);
    int cl = cmin - 1 < cl_pref ? cmin - 1 : cl_pref;    // every window needs >= 1 coarse bin
    while ((P.G >> cl) > 24576u && cl < cmin - 1) ++cl;
    g.cl = cl;
    g.chmax = cmax - 1 - cl;
    g.NC = (uint32_t)(P.G >> cl);
    for (int w = 0; w <= P.W; ++w) g.binoff[w] = P.boff[w] >> cl;
    for (int w = 0; w < P.W; ++w) g.width[w] = P.width[w];
    g.pts_per_block = 0;
    return g;
}

int msm_sort_lds(MsmEngine& E, const void* d_scalars, uint32_t npts, int sbits) {
    const MsmPlan& P = E.last_plan;
    hipStream_t st = E.sort_st;
    SortGeom g = make_geom(P);
    if (g.cl > 12 || g.NC > 24576u) return fail(BLZ_ERR_UNKNOWN, "sort geometry out of range (c=%d W=%d)", P.c, P.W);
    // points per block: enough entries per block to amortise the NC-sized LDS sweeps, enough blocks to fill the chip
    uint32_t ppb = (uint32_t)(((uint64_t)g.NC * (uint32_t)exp_knob("BLAZE_SORT_EPB", 64) + P.W - 1) / P.W);
    if (ppb < 4096) ppb = 4096;
    if (ppb > 65536) ppb = 65536;
    ppb = (ppb + SORT_THREADS - 1) / SORT_THREADS * SORT_THREADS;
    g.pts_per_block = ppb;
    const uint32_t nblk = (npts + ppb - 1) / ppb;
    const uint64_t max_entries = (uint64_t)npts * P.W;
    BLZ_TRY(E.coarse.reserve(((size_t)g.NC * 2 + 2) * 4));
    // (index | sign) as u32 and the fine digit as u16, in two arrays: the fine count reads only the second
    BLZ_TRY(E.inter.reserve(max_entries * 6 + 64));
    uint32_t* inter_idx = E.inter.as<uint32_t>();
    uint16_t* inter_fine = reinterpret_cast<uint16_t*>(inter_idx + max_entries);
    E.sort_inter_fine = inter_fine;
    uint32_t* coarse_count = E.coarse.as<uint32_t>();
    uint32_t* coarse_off = coarse_count + g.NC;
    BLZ_HIP(hipMemsetAsync(coarse_count, 0, (size_t)g.NC * 4, st), BLZ_ERR_UNKNOWN);
    const size_t lds = (size_t)g.NC * 4;
    BLZ_SW_DISPATCH(sbits, {
        BLZ_TRY(ensure_dynamic_lds((const void*)k_coarse_count<SW>, 128 * 1024));
        BLZ_TRY(ensure_dynamic_lds((const void*)k_coarse_scatter<SW>, 128 * 1024));
    });
    const uint32_t* sc = (const uint32_t*)d_scalars;
    BLZ_SW_DISPATCH(sbits, hipLaunchKernelGGL(k_coarse_count<SW>, dim3(nblk), dim3(SORT_THREADS), lds, st, sc, npts, g, coarse_count));
    hipLaunchKernelGGL(k_coarse_scan, dim3(1), dim3(1024), 0, st, coarse_count, g.NC, coarse_off);
    // BLAZE_SORT_STAGED=0 selects the unstaged coarse scatter (kept for A/B measurements)
    if (exp_knob("BLAZE_SORT_STAGED", 1) != 0 && g.chmax <= 11 && g.cl <= 12) {
        const bool small = npts <= CS_SMALL_PTS;
        const uint32_t pts_blk = (uint32_t)CS_THREADS * (small ? CS_T_SMALL : CS_T);
        const uint32_t nblk_cs = (npts + pts_blk - 1) / pts_blk;
        const size_t lds_cs = ((size_t)3 << g.chmax) * 4 + (size_t)pts_blk * 8;
        BLZ_SW_DISPATCH(sbits, {
            BLZ_TRY(ensure_dynamic_lds((const void*)k_coarse_scatter_staged<SW, CS_T>, 96 * 1024));
            BLZ_TRY(ensure_dynamic_lds((const void*)k_coarse_scatter_staged<SW, CS_T_SMALL>, 96 * 1024));
            if (small) hipLaunchKernelGGL((k_coarse_scatter_staged<SW, CS_T_SMALL>), dim3(nblk_cs), dim3(CS_THREADS), lds_cs, st, sc, npts, g, coarse_count, inter_idx, inter_fine);
            else hipLaunchKernelGGL((k_coarse_scatter_staged<SW, CS_T>), dim3(nblk_cs), dim3(CS_THREADS), lds_cs, st, sc, npts, g, coarse_count, inter_idx, inter_fine);
        });
    } else {
        BLZ_SW_DISPATCH(sbits, hipLaunchKernelGGL(k_coarse_scatter<SW>, dim3(nblk), dim3(SORT_THREADS), lds, st, sc, npts, g, coarse_count, inter_idx, inter_fine));
    }
    // work list of the fine passes (device-built, no host sync)
    const uint32_t max_slices = (uint32_t)(max_entries / SLICE) + g.NC + 1;
    BLZ_TRY(E.slice_map.reserve(((size_t)max_slices + 2) * 8));
    uint2* slice_map = E.slice_map.as<uint2>() + 1;             // element 0 holds the slice count
    uint32_t* nslices = E.slice_map.as<uint32_t>();
    hipLaunchKernelGGL(k_slice_map, dim3(1), dim3(SLICEMAP_THREADS), 0, st, coarse_off, g.NC, slice_map, nslices);
    E.sort_slices = max_slices;
    E.sort_cl = g.cl;
    E.sort_nc = g.NC;
    hipLaunchKernelGGL(k_fine_count, dim3(max_slices), dim3(FINE_THREADS), (size_t)4 << g.cl, st, inter_fine, coarse_off,
                       slice_map, nslices, g.cl, E.sb().count.as<uint32_t>());
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

int msm_sort_lds_scatter(MsmEngine& E) {
    hipStream_t st = E.sort_st;
    uint32_t* coarse_off = E.coarse.as<uint32_t>() + E.sort_nc;
    BLZ_TRY(ensure_dynamic_lds((const void*)k_fine_scatter, 158 * 1024));
    // staging entries per round (6 bytes each): what is left of the LDS after the two per-bucket arrays
    size_t budget = (size_t)exp_knob("BLAZE_SORT_FS_KB", 157) * 1024 - ((size_t)2 << E.sort_cl) * 4;
    uint32_t round_cap = (uint32_t)(budget / 6);
    if (round_cap > (uint32_t)FS_ROUND) round_cap = FS_ROUND;
    round_cap &= ~1023u;
    const size_t lds = ((size_t)2 << E.sort_cl) * 4 + (size_t)round_cap * 6;
    hipLaunchKernelGGL(k_fine_scatter, dim3(E.sort_slices), dim3(FS_THREADS), lds, st, E.inter.as<uint32_t>(), (const uint16_t*)E.sort_inter_fine, coarse_off,
                       E.slice_map.as<uint2>() + 1, E.slice_map.as<uint32_t>(), E.sort_cl, round_cap, E.sb().count.as<uint32_t>(),
                       E.sb().entries.as<uint32_t>());
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

}  // namespace blz
