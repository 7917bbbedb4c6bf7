// Three-level digit sort with a small footprint, built to run UNDERNEATH another task's bucket accumulation.
//
// k_accumulate (2 waves x 200 VGPRs per SIMD, no LDS) leaves 112 VGPRs per SIMD, four wave slots and the whole LDS of
// every CU idle, and ~85 % of the memory system: the digit sort of the NEXT task (msm_sort.hip: 10.9 ms of a 122 ms
// step at 2^26) fits in there - if its kernels fit.  The two-level sort's do not (512 / 1024-thread blocks holding 16 - 24
// entries per lane in registers), and shrinking them shrinks the runs they write: with 1024-way splits a 256-thread
// block produces pieces of one or two entries, the memory system drowns in lone stores, and the accumulation above it
// slows down by more than the sort costs alone (round 2 measured exactly that: profiles/r02_sort_under_accumulate.txt).
// Hence three levels of at most 256 / 128 / 128 ways, so that a block of 256 lanes with 12 - 16 entries per lane still
// writes pieces of 24 - 32 entries, every kernel at <= 96 VGPRs and 256 threads, wave priority raised (at equal priority
// the accumulation's older waves starve the sort's):
//
//   bucket g (flat over the windows)  =  level-1 bin (g >> 14)  |  level-2 digit (7 bits)  |  level-3 digit (7 bits)
//   k3_l1_count    scalars -> LDS histogram over the <= 2048 level-1 bins -> cnt1
//   k3_l1_scatter  3072 scalars per block parked in LDS (lane-private, word-major: the block's register file stays
//                  free), window by window: digits, LDS rank, bin-major stage, slot-major copy-out of
//                  (index | sign : u32, low 14 bits of the bucket : u16) grouped by level-1 bin
//   k3_l2_count / k3_l2_scatter   per slice of 4096 entries of a level-1 bin: 128-way split by the middle 7 bits;
//                  out: (index | sign : u32, low 7 bits : u8) grouped by level-2 bin (= 128 consecutive buckets)
//   k3_l3          one block per level-2 bin: bucket counts (-> count[], the scans of msm.hip turn them into off[] /
//                  unit_off[]) and the bin's entries in bucket order, staged in LDS and written as one contiguous run
//                  - the bin's position in entries[] is already final, no global bucket scan is needed before it
// HBM traffic ~26 GB at 2^26 (the two-level sort: 18 GB) - it is hidden, what counts is that it is coalesced.
// Same digits, same buckets as msm_sort.hip; the order of a bucket's entries differs (the group law does not care).
// Used when another task of the handle is in flight (msm.hip run()); a task with nothing to hide under keeps the
// two-level sort, which is faster when it has the chip to itself.  BLAZE_SORT_HIDE = 0 never / 2 always (tests).
#include "msm_engine.hpp"
#include "msm_digits.hip.hpp"

namespace blz {

constexpr int S3_THREADS = 256;
constexpr int S3_SH1 = 14;                      // level-1 bin = bucket >> 14
constexpr int S3_SH2 = 7;                       // level-2 bin = bucket >> 7
constexpr int S3_T = 12;                        // scalars per lane of the level-1 scatter
constexpr int S3_PB = S3_THREADS * S3_T;        // 3072 points per block: 96 KiB of scalars + 24 KiB of stage
constexpr uint32_t S3_SLICE2 = 4096;            // entries per level-2 work item (16 per lane)
constexpr int S3_T2 = S3_SLICE2 / S3_THREADS;
constexpr int S3_T3 = 28;                       // level 3: entries per lane held in registers
constexpr uint32_t S3_R3 = S3_THREADS * S3_T3;  // ... so a level-2 bin of up to 7168 entries takes one pass (mean at 2^26: 5.8 K, sigma 76)
constexpr uint32_t S3_MAXNB1 = 2048;            // level-1 bins in all (LDS histogram of k3_l1_count)
constexpr uint32_t S3_CNT_PTS = 16384;          // points per block of k3_l1_count

struct S3Geom {
    int W, prio;
    uint32_t NB1;
    uint8_t width[MSM_MAX_W];
    uint16_t bitoff[MSM_MAX_W];        // first scalar bit of window w
    uint16_t binoff1[MSM_MAX_W + 1];   // first level-1 bin of window w
};

// wave priority of the sort's kernels (s_setprio takes an immediate); BLAZE_SORT_PRIO = 0..3, default 3
__device__ __forceinline__ void s3_setprio(int p) {
    if (p >= 3) __builtin_amdgcn_s_setprio(3);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
}
#define S3_PRIO() s3_setprio(prio)

__device__ __forceinline__ uint32_t s3_wave_incl_scan(uint32_t v) {
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o, 64);
        if ((threadIdx.x & 63u) >= (uint32_t)o) v += t;
    }
    return v;
}
// exclusive scan of one value per thread over the block's 256 threads; returns the exclusive prefix, *total = sum
__device__ __forceinline__ uint32_t s3_block_excl_scan(uint32_t v, uint32_t* wave_tot, uint32_t* total) {
    const uint32_t incl = s3_wave_incl_scan(v);
    if ((threadIdx.x & 63u) == 63u) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t q = 0; q < (threadIdx.x >> 6); ++q) wbase += wave_tot[q];
    *total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    return wbase + incl - v;
}

// ---------------------------------------------------------------------------------------------- level 1
// NW: 32-bit words per scalar - 8 (256-bit scalars), 1 (the 32-bit chunks of a precompute handle: pf = 8) or 2 (its 64-bit
// chunks on the checked-table plan: arena_tables.hip)
template <int NW>
__global__ __launch_bounds__(S3_THREADS, 4) void k3_l1_count(const uint32_t* __restrict__ scalars, uint32_t npts, S3Geom g,
                                                            uint32_t* __restrict__ cnt1) {
    const int prio = g.prio;
    S3_PRIO();
    __shared__ uint32_t hist[S3_MAXNB1];
    for (uint32_t i = threadIdx.x; i < g.NB1; i += S3_THREADS) hist[i] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * S3_CNT_PTS;
    uint32_t end = base + S3_CNT_PTS;
    if (end > npts) end = npts;
    for (uint32_t p0 = base + threadIdx.x; p0 < end; p0 += 4 * S3_THREADS) {
        ScalarWords<NW> sw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t p = p0 + u * S3_THREADS;
            sw[u].load(scalars, p < end ? p : p0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (p0 + u * S3_THREADS >= end) break;
            uint32_t carry = 0;
            for (int w = 0; w < g.W; ++w) {
                const int cw = g.width[w];
                const int d = sw[u].next(cw, (1u << cw) - 1u, 1u << (cw - 1), carry);
                if (d != 0) {
                    const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
                    atomicAdd(&hist[g.binoff1[w] + (b >> S3_SH1)], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < g.NB1; i += S3_THREADS) {
        const uint32_t v = hist[i];
        if (v) atomicAdd(&cnt1[i], v);
    }
}

// exclusive scan of n <= 2048 counters (one block): off[0..n], cur[i] = off[i]
__global__ __launch_bounds__(S3_THREADS) void k3_scan_small(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t* __restrict__ off,
                                                           uint32_t* __restrict__ cur, int prio) {
    S3_PRIO();
    __shared__ uint32_t wave_tot[4];
    uint32_t v[8], sum = 0;
    const uint32_t b0 = threadIdx.x * 8u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        v[k] = b0 + k < n ? cnt[b0 + k] : 0u;
        sum += v[k];
    }
    uint32_t total;
    uint32_t run = s3_block_excl_scan(sum, wave_tot, &total);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (b0 + k < n) {
            off[b0 + k] = run;
            cur[b0 + k] = run;
        }
        run += v[k];
    }
    if (threadIdx.x == 0) off[n] = total;
}

template <int NW>
__global__ __launch_bounds__(S3_THREADS, 4) void k3_l1_scatter(const uint32_t* __restrict__ scalars, uint32_t npts, S3Geom g,
                                                              uint32_t* __restrict__ cur1, uint32_t* __restrict__ o_idx,
                                                              uint16_t* __restrict__ o_rem) {
    const int prio = g.prio;
    S3_PRIO();
    extern __shared__ __attribute__((aligned(16))) uint32_t sh[];
    uint32_t* sc = sh;                                          // [NW][S3_PB]: word j of the lane's scalar u at j * PB + u * 256 + tid
    uint2* stage = reinterpret_cast<uint2*>(sh + NW * S3_PB);   // [S3_PB]
    uint32_t* hist = sh + (NW + 2) * S3_PB;                     // [256]
    uint32_t* lstart = hist + 256;
    uint32_t* gbase = lstart + 256;
    __shared__ uint32_t wave_tot[4];
    const uint32_t tid = threadIdx.x;
    const uint32_t base = blockIdx.x * (uint32_t)S3_PB;
    // the scalars are lane-private: LDS is this block's register spill area, nobody else reads a lane's words
#pragma unroll
    for (int u = 0; u < S3_T; ++u) {
        const uint32_t p = base + u * S3_THREADS + tid;
        const uint32_t i = u * S3_THREADS + tid;
        if constexpr (NW == 8) {
            uint4 a = make_uint4(0, 0, 0, 0), b = make_uint4(0, 0, 0, 0);
            if (p < npts) {
                const uint4* q = reinterpret_cast<const uint4*>(scalars) + 2 * (size_t)p;
                a = q[0];
                b = q[1];
            }
            sc[0 * S3_PB + i] = a.x; sc[1 * S3_PB + i] = a.y; sc[2 * S3_PB + i] = a.z; sc[3 * S3_PB + i] = a.w;
            sc[4 * S3_PB + i] = b.x; sc[5 * S3_PB + i] = b.y; sc[6 * S3_PB + i] = b.z; sc[7 * S3_PB + i] = b.w;
        } else if constexpr (NW == 2) {
            uint2 a = make_uint2(0, 0);
            if (p < npts) a = reinterpret_cast<const uint2*>(scalars)[p];
            sc[0 * S3_PB + i] = a.x; sc[1 * S3_PB + i] = a.y;
        } else {
            sc[i] = p < npts ? scalars[p] : 0u;
        }
    }
    uint32_t carry = 0;   // bit u: the carry of the lane's scalar u into the next window
    for (int w = 0; w < g.W; ++w) {
        const uint32_t cw = g.width[w], off = g.bitoff[w];
        const uint32_t nb = (uint32_t)g.binoff1[w + 1] - (uint32_t)g.binoff1[w];   // <= 256
        const uint32_t j = off >> 5, shb = off & 31u, mask = (1u << cw) - 1u, half = 1u << (cw - 1);
        if (tid < nb) hist[tid] = 0;
        __syncthreads();
        uint32_t key[S3_T], rk[S3_T];   // key = low 14 bits | bin << 14 | sign << 31;  rk = rank in the bin, ~0 = no entry
#pragma unroll
        for (int u = 0; u < S3_T; ++u) {
            const uint32_t i = u * S3_THREADS + tid;
            const uint32_t lo = j < (uint32_t)NW ? sc[j * S3_PB + i] : 0u;
            const uint32_t hi = j + 1 < (uint32_t)NW ? sc[(j + 1) * S3_PB + i] : 0u;
            const uint32_t raw = (shb ? __builtin_amdgcn_alignbit(hi, lo, shb) : lo) & mask;
            const uint32_t v = raw + ((carry >> u) & 1u);
            int d;
            if (v > half) { d = (int)v - (int)(half << 1); carry |= 1u << u; }
            else { d = (int)v; carry &= ~(1u << u); }
            rk[u] = ~0u;
            key[u] = 0;
            if (d != 0 && base + i < npts) {
                const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
                const uint32_t bin = b >> S3_SH1;
                key[u] = (b & 0x3fffu) | (bin << 14) | (d < 0 ? 0x80000000u : 0u);
                rk[u] = atomicAdd(&hist[bin], 1u);
            }
        }
        __syncthreads();
        {   // exclusive scan over the window's bins (one per thread) + one global reservation per non-empty bin
            const uint32_t v = tid < nb ? hist[tid] : 0u;
            uint32_t total;
            const uint32_t excl = s3_block_excl_scan(v, wave_tot, &total);
            if (tid < nb) {
                lstart[tid] = excl;
                gbase[tid] = v ? atomicAdd(&cur1[(uint32_t)g.binoff1[w] + tid], v) : 0u;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < S3_T; ++u) {
                if (rk[u] != ~0u) {
                    const uint32_t bin = (key[u] >> 14) & 0x1ffu;
                    const uint32_t p = base + u * S3_THREADS + tid;
                    stage[lstart[bin] + rk[u]] = make_uint2(p | (key[u] & 0x80000000u), (key[u] & 0x3fffu) | (bin << 16));
                }
            }
            __syncthreads();
            for (uint32_t slot = tid; slot < total; slot += S3_THREADS) {
                const uint2 e = stage[slot];
                const uint32_t bin = e.y >> 16;
                const uint32_t dst = gbase[bin] + (slot - lstart[bin]);
                o_idx[dst] = e.x;
                o_rem[dst] = (uint16_t)e.y;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------- work lists and scans
// bin k (off[k] .. off[k+1]) cut into ceil(size / slice) work items (k, piece); one block, no host round trip
__global__ __launch_bounds__(S3_THREADS) void k3_slice_map(const uint32_t* __restrict__ off, uint32_t nbins, uint32_t slice,
                                                          uint2* __restrict__ map, uint32_t* __restrict__ nitems, int prio) {
    S3_PRIO();
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t carry_sh;
    if (threadIdx.x == 0) carry_sh = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nbins; base += S3_THREADS) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t size = i < nbins ? off[i + 1] - off[i] : 0u;
        const uint32_t v = (size + slice - 1) / slice;
        uint32_t total;
        const uint32_t excl = s3_block_excl_scan(v, wave_tot, &total);
        const uint32_t first = carry_sh + excl;
        for (uint32_t q = 0; q < v; ++q) map[first + q] = make_uint2(i, q);
        __syncthreads();
        if (threadIdx.x == 0) carry_sh += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *nitems = carry_sh;
}

// three-kernel exclusive scan of n counters (n <= 2048 * 256): block sums, their scan, final
__global__ __launch_bounds__(S3_THREADS) void k3_scan_a(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t* __restrict__ bsum, int prio) {
    S3_PRIO();
    __shared__ uint32_t wave_tot[4];
    const uint32_t b0 = blockIdx.x * 2048u + threadIdx.x * 8u;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += b0 + k < n ? cnt[b0 + k] : 0u;
    uint32_t total;
    (void)s3_block_excl_scan(sum, wave_tot, &total);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}
__global__ __launch_bounds__(S3_THREADS) void k3_scan_b(uint32_t* __restrict__ bsum, uint32_t nblocks, int prio) {   // nblocks <= 256
    S3_PRIO();
    __shared__ uint32_t wave_tot[4];
    const uint32_t v = threadIdx.x < nblocks ? bsum[threadIdx.x] : 0u;
    uint32_t total;
    const uint32_t excl = s3_block_excl_scan(v, wave_tot, &total);
    if (threadIdx.x < nblocks) bsum[threadIdx.x] = excl;
    if (threadIdx.x == 0) bsum[256] = total;
}
__global__ __launch_bounds__(S3_THREADS) void k3_scan_c(const uint32_t* __restrict__ cnt, uint32_t n, const uint32_t* __restrict__ bsum,
                                                       uint32_t* __restrict__ off, uint32_t* __restrict__ cur, int prio) {
    S3_PRIO();
    __shared__ uint32_t wave_tot[4];
    const uint32_t b0 = blockIdx.x * 2048u + threadIdx.x * 8u;
    uint32_t v[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        v[k] = b0 + k < n ? cnt[b0 + k] : 0u;
        sum += v[k];
    }
    uint32_t total;
    uint32_t run = bsum[blockIdx.x] + s3_block_excl_scan(sum, wave_tot, &total);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (b0 + k < n) {
            off[b0 + k] = run;
            cur[b0 + k] = run;
        }
        run += v[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) off[n] = bsum[256];
}

// ---------------------------------------------------------------------------------------------- level 2
__device__ __forceinline__ bool s3_item(const uint32_t* off1, const uint2* map, const uint32_t* nitems, uint32_t id, uint32_t slice,
                                        uint32_t& k, uint32_t& lo, uint32_t& hi) {
    if (id >= *nitems) return false;
    const uint2 m = map[id];
    k = m.x;
    const uint32_t a = off1[k], b = off1[k + 1];
    lo = a + m.y * slice;
    hi = lo + slice < b ? lo + slice : b;
    return lo < hi;
}

__global__ __launch_bounds__(S3_THREADS, 4) void k3_l2_count(const uint16_t* __restrict__ rem, const uint32_t* __restrict__ off1,
                                                            const uint2* __restrict__ map, const uint32_t* __restrict__ nitems,
                                                            uint32_t* __restrict__ cnt2, int prio) {
    S3_PRIO();
    __shared__ uint32_t hist[128];
    uint32_t k, lo, hi;
    if (!s3_item(off1, map, nitems, blockIdx.x, S3_SLICE2, k, lo, hi)) return;
    if (threadIdx.x < 128) hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t r[S3_T2];
#pragma unroll
    for (int t = 0; t < S3_T2; ++t) {
        const uint32_t i = lo + t * S3_THREADS + threadIdx.x;
        r[t] = i < hi ? rem[i] : 0xffffffffu;
    }
#pragma unroll
    for (int t = 0; t < S3_T2; ++t)
        if (r[t] != 0xffffffffu) atomicAdd(&hist[r[t] >> S3_SH2], 1u);
    __syncthreads();
    if (threadIdx.x < 128) {
        const uint32_t v = hist[threadIdx.x];
        if (v) atomicAdd(&cnt2[(k << 7) + threadIdx.x], v);
    }
}

__global__ __launch_bounds__(S3_THREADS, 4) void k3_l2_scatter(const uint32_t* __restrict__ i_idx, const uint16_t* __restrict__ i_rem,
                                                              const uint32_t* __restrict__ off1, const uint2* __restrict__ map,
                                                              const uint32_t* __restrict__ nitems, uint32_t* __restrict__ cur2,
                                                              uint32_t* __restrict__ o_idx, uint8_t* __restrict__ o_lo, int prio) {
    S3_PRIO();
    __shared__ uint2 stage[S3_SLICE2];
    __shared__ uint32_t hist[128], lstart[128], gbase[128];
    __shared__ uint32_t wave_tot[4];
    uint32_t k, lo, hi;
    if (!s3_item(off1, map, nitems, blockIdx.x, S3_SLICE2, k, lo, hi)) return;
    const uint32_t tid = threadIdx.x;
    if (tid < 128) hist[tid] = 0;
    __syncthreads();
    uint32_t ex[S3_T2], ky[S3_T2];   // ky = low 7 bits | level-2 digit << 8 | rank << 16; ~0 = no entry
#pragma unroll
    for (int t = 0; t < S3_T2; ++t) {
        const uint32_t i = lo + t * S3_THREADS + tid;
        ky[t] = 0xffffffffu;
        ex[t] = 0;
        if (i < hi) {
            const uint32_t r = i_rem[i];
            ex[t] = i_idx[i];
            const uint32_t h2 = r >> S3_SH2;
            const uint32_t rank = atomicAdd(&hist[h2], 1u);   // < 4096
            ky[t] = (r & 127u) | (h2 << 8) | (rank << 16);
        }
    }
    __syncthreads();
    const uint32_t v = tid < 128 ? hist[tid] : 0u;
    uint32_t total;
    const uint32_t excl = s3_block_excl_scan(v, wave_tot, &total);
    if (tid < 128) {
        lstart[tid] = excl;
        gbase[tid] = v ? atomicAdd(&cur2[(k << 7) + tid], v) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < S3_T2; ++t) {
        if (ky[t] != 0xffffffffu) {
            const uint32_t h2 = (ky[t] >> 8) & 127u;
            stage[lstart[h2] + (ky[t] >> 16)] = make_uint2(ex[t], ky[t] & 0x7fffu);
        }
    }
    __syncthreads();
    for (uint32_t slot = tid; slot < total; slot += S3_THREADS) {
        const uint2 e = stage[slot];
        const uint32_t h2 = (e.y >> 8) & 127u;
        const uint32_t dst = gbase[h2] + (slot - lstart[h2]);
        o_idx[dst] = e.x;
        o_lo[dst] = (uint8_t)(e.y & 127u);
    }
}

// ---------------------------------------------------------------------------------------------- level 3
// one block per level-2 bin (128 consecutive buckets): bucket counts, then the bin's entries in bucket order.  The bin's
// run [off2[j], off2[j+1]) of entries[] is final already: buckets are laid out in order and a bin is a whole group of
// them.  A bin of up to S3_R3 entries is read once (entries and ranks wait in registers while the counts are scanned),
// placed in an LDS image of the run and leaves as one contiguous copy; a larger one (hot buckets: the reference
// harness's repeated tile) is written entry by entry - correct, slow, and not what this path is chosen for.
__global__ __launch_bounds__(S3_THREADS, 4) void k3_l3(const uint32_t* __restrict__ i_idx, const uint8_t* __restrict__ i_lo,
                                                      const uint32_t* __restrict__ off2, uint32_t* __restrict__ count,
                                                      uint32_t* __restrict__ entries, int prio) {
    S3_PRIO();
    __shared__ uint32_t out[S3_R3];
    __shared__ uint32_t hist[128], cursor[128];
    __shared__ uint32_t wave_tot[4];
    const uint32_t j = blockIdx.x, tid = threadIdx.x;
    const uint32_t a = off2[j], b = off2[j + 1], s = b - a;
    if (tid < 128) hist[tid] = 0;
    __syncthreads();
    if (s <= S3_R3) {
        // one pass over the bin: every lane keeps its entries and their ranks in registers (all loads in flight at once)
        uint32_t ex[S3_T3], ky[S3_T3];   // ky = bucket (7 bits) | rank in the bucket << 8; ~0 = no entry
#pragma unroll
        for (int t = 0; t < S3_T3; ++t) {
            const uint32_t i = a + t * S3_THREADS + tid;
            ky[t] = 0xffffffffu;
            ex[t] = 0;
            if (i < b) {
                ky[t] = i_lo[i];
                ex[t] = i_idx[i];
            }
        }
#pragma unroll
        for (int t = 0; t < S3_T3; ++t)
            if (ky[t] != 0xffffffffu) ky[t] |= atomicAdd(&hist[ky[t]], 1u) << 8;
        __syncthreads();
        const uint32_t v = tid < 128 ? hist[tid] : 0u;
        uint32_t total;
        const uint32_t excl = s3_block_excl_scan(v, wave_tot, &total);
        if (tid < 128) {
            count[((size_t)j << 7) + tid] = v;
            cursor[tid] = excl;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < S3_T3; ++t)
            if (ky[t] != 0xffffffffu) out[cursor[ky[t] & 127u] + (ky[t] >> 8)] = ex[t];
        __syncthreads();
        for (uint32_t t = tid; t < s; t += S3_THREADS) entries[a + t] = out[t];
    } else {
        // a bin beyond the registers - the precompute shapes (2^29 points of 32-bit scalars in 2 x 2^16 buckets: a million
        // entries per bin, 8192 per bucket), hot buckets (the reference harness's repeated tile): count the whole bin, then
        // place it chunk by chunk - S3_R3 entries ranked in LDS, staged in bucket order, copied out as one contiguous piece
        // per bucket (a bucket's piece of a chunk is contiguous in the stage and at the bucket's cursor in entries[])
        for (uint32_t i = a + tid; i < b; i += S3_THREADS) atomicAdd(&hist[i_lo[i]], 1u);
        __syncthreads();
        const uint32_t v = tid < 128 ? hist[tid] : 0u;
        uint32_t total;
        const uint32_t excl = s3_block_excl_scan(v, wave_tot, &total);
        if (tid < 128) {
            count[((size_t)j << 7) + tid] = v;
            cursor[tid] = excl;      // where the bucket's next piece goes, relative to a
        }
        // (half the entries per lane of the one-pass path: this path must not raise the kernel's register count - the sort
        // only hides beside the accumulation at <= 72 VGPRs)
        constexpr int TB = S3_T3 / 2;
        constexpr uint32_t RB = S3_THREADS * TB;
        __shared__ uint32_t chist[128], cstart[128];
        __shared__ uint8_t cbkt[RB];
        for (uint32_t c0 = a; c0 < b; c0 += RB) {
            const uint32_t c1 = c0 + RB < b ? c0 + RB : b;
            if (tid < 128) chist[tid] = 0;
            __syncthreads();
            uint32_t ex[TB], ky[TB];
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                const uint32_t i = c0 + t * S3_THREADS + tid;
                ky[t] = 0xffffffffu;
                ex[t] = 0;
                if (i < c1) {
                    ky[t] = i_lo[i];
                    ex[t] = i_idx[i];
                }
            }
#pragma unroll
            for (int t = 0; t < TB; ++t)
                if (ky[t] != 0xffffffffu) ky[t] |= atomicAdd(&chist[ky[t]], 1u) << 8;
            __syncthreads();
            const uint32_t cv = tid < 128 ? chist[tid] : 0u;
            uint32_t ctotal;
            const uint32_t cexcl = s3_block_excl_scan(cv, wave_tot, &ctotal);
            if (tid < 128) cstart[tid] = cexcl;
            __syncthreads();
#pragma unroll
            for (int t = 0; t < TB; ++t)
                if (ky[t] != 0xffffffffu) {
                    const uint32_t slot = cstart[ky[t] & 127u] + (ky[t] >> 8);
                    out[slot] = ex[t];
                    cbkt[slot] = (uint8_t)(ky[t] & 127u);
                }
            __syncthreads();
            for (uint32_t slot = tid; slot < ctotal; slot += S3_THREADS) {
                const uint32_t bk = cbkt[slot];
                entries[a + cursor[bk] + (slot - cstart[bk])] = out[slot];
            }
            __syncthreads();
            if (tid < 128) cursor[tid] += chist[tid];
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------- host
bool msm_sort3_ok(const MsmPlan& P, int sbits) {
    if ((sbits != 256 && sbits != 64 && sbits != 32) || P.W < 1) return false;
    for (int w = 0; w < P.W; ++w)
        if (P.width[w] < S3_SH1 + 1 || P.width[w] > 23) return false;   // every window a whole number of level-1 bins, <= 256 of them
    if ((P.G >> S3_SH1) > S3_MAXNB1 || (P.G & ((1u << S3_SH1) - 1u))) return false;
    if ((P.G >> S3_SH2) > 2048u * 256u) return false;
    return true;
}

int msm_sort3_max_vgprs() {
    static int cached = -1;
    if (cached >= 0) return cached;
    int mx = 0;
    const void* ks[] = {(const void*)k3_l1_count<8>, (const void*)k3_l1_scatter<8>, (const void*)k3_l1_count<1>, (const void*)k3_l1_scatter<1>,
                        (const void*)k3_l1_count<2>, (const void*)k3_l1_scatter<2>, (const void*)k3_l2_count, (const void*)k3_l2_scatter, (const void*)k3_l3};
    for (const void* k : ks) {
        hipFuncAttributes a;
        if (hipFuncGetAttributes(&a, k) != hipSuccess) {
            (void)hipGetLastError();
            return cached = 0;
        }
        if (a.numRegs > mx) mx = a.numRegs;
    }
    return cached = mx;
}

int msm_sort3(MsmEngine& E, const void* d_scalars, uint32_t npts, int sbits) {
    const MsmPlan& P = E.last_plan;
    hipStream_t st = E.sort_st;
    MsmEngine::SortBufs& B = E.sb();
    S3Geom g;
    g.W = P.W;
    g.prio = exp_knob("BLAZE_SORT_PRIO", 3);
    g.NB1 = (uint32_t)(P.G >> S3_SH1);
    uint32_t bit = 0;
    for (int w = 0; w < P.W; ++w) {
        g.width[w] = P.width[w];
        g.bitoff[w] = (uint16_t)bit;
        g.binoff1[w] = (uint16_t)(P.boff[w] >> S3_SH1);
        bit += P.width[w];
    }
    g.binoff1[P.W] = (uint16_t)(P.boff[P.W] >> S3_SH1);
    const uint32_t NB1 = g.NB1, NB2 = (uint32_t)(P.G >> S3_SH2);
    const uint64_t max_entries = (uint64_t)npts * P.W;
    const uint32_t max_items = (uint32_t)(max_entries / S3_SLICE2) + NB1 + 1;
    const uint32_t nsb = (NB2 + 2047u) / 2048u;
    // tables: cnt1 | off1 (+1) | cur1 | cnt2 | off2 (+1) | cur2 | bsum (257) | nitems (1, padded) | map (uint2 per item)
    const size_t tab_dw = (size_t)3 * (NB1 + 2) + (size_t)3 * (NB2 + 2) + 260 + 4 + 2 * ((size_t)max_items + 2);
    BLZ_TRY(E.sort3_tabs.reserve(tab_dw * 4));
    uint32_t* cnt1 = E.sort3_tabs.as<uint32_t>();
    uint32_t* off1 = cnt1 + NB1 + 2;
    uint32_t* cur1 = off1 + NB1 + 2;
    uint32_t* cnt2 = cur1 + NB1 + 2;
    uint32_t* off2 = cnt2 + NB2 + 2;
    uint32_t* cur2 = off2 + NB2 + 2;
    uint32_t* bsum = cur2 + NB2 + 2;
    uint32_t* nitems = bsum + 260;
    uint2* map = reinterpret_cast<uint2*>(nitems + 4);
    BLZ_TRY(E.inter.reserve(max_entries * 6 + 64));
    BLZ_TRY(E.inter2.reserve(max_entries * 5 + 64));
    uint32_t* i1_idx = E.inter.as<uint32_t>();
    uint16_t* i1_rem = reinterpret_cast<uint16_t*>(i1_idx + max_entries);
    uint32_t* i2_idx = E.inter2.as<uint32_t>();
    uint8_t* i2_lo = reinterpret_cast<uint8_t*>(i2_idx + max_entries);
    const uint32_t* sc = (const uint32_t*)d_scalars;
    const dim3 blk(S3_THREADS);

    BLZ_HIP(hipMemsetAsync(cnt1, 0, (size_t)(NB1 + 2) * 4, st), BLZ_ERR_UNKNOWN);
    BLZ_HIP(hipMemsetAsync(cnt2, 0, (size_t)(NB2 + 2) * 4, st), BLZ_ERR_UNKNOWN);
    BLZ_SW_DISPATCH(sbits, hipLaunchKernelGGL(k3_l1_count<SW>, dim3((npts + S3_CNT_PTS - 1) / S3_CNT_PTS), blk, 0, st, sc, npts, g, cnt1));
    hipLaunchKernelGGL(k3_scan_small, dim3(1), blk, 0, st, cnt1, NB1, off1, cur1, g.prio);
    BLZ_SW_DISPATCH(sbits, {
        const size_t lds1 = (size_t)((SW + 2) * S3_PB + 3 * 256) * 4;
        BLZ_TRY(ensure_dynamic_lds((const void*)k3_l1_scatter<SW>, (int)lds1));
        hipLaunchKernelGGL(k3_l1_scatter<SW>, dim3((npts + S3_PB - 1) / S3_PB), blk, lds1, st, sc, npts, g, cur1, i1_idx, i1_rem);
    });
    hipLaunchKernelGGL(k3_slice_map, dim3(1), blk, 0, st, off1, NB1, S3_SLICE2, map, nitems, g.prio);
    hipLaunchKernelGGL(k3_l2_count, dim3(max_items), blk, 0, st, i1_rem, off1, map, nitems, cnt2, g.prio);
    hipLaunchKernelGGL(k3_scan_a, dim3(nsb), blk, 0, st, cnt2, NB2, bsum, g.prio);
    hipLaunchKernelGGL(k3_scan_b, dim3(1), blk, 0, st, bsum, nsb, g.prio);
    hipLaunchKernelGGL(k3_scan_c, dim3(nsb), blk, 0, st, cnt2, NB2, bsum, off2, cur2, g.prio);
    hipLaunchKernelGGL(k3_l2_scatter, dim3(max_items), blk, 0, st, i1_idx, i1_rem, off1, map, nitems, cur2, i2_idx, i2_lo, g.prio);
    hipLaunchKernelGGL(k3_l3, dim3(NB2), blk, 0, st, i2_idx, i2_lo, off2, B.count.as<uint32_t>(), B.entries.as<uint32_t>(), g.prio);
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

// ================================================================================================ window-table tasks
// The same three levels for plans whose windows share ONE bucket set (MsmPlan::table, msm_impl.hip.hpp k_build_window_table): W windows of c bits,
// G = 2^(c-1) buckets, an entry carries (base * W + window) | sign.  What changes against the kernels above:
//   * every window scatters into the same <= 256 level-1 bins (bucket >> sh1, sh1 = c - 9), so the level-1 remainder is up
//     to 17 bits wide: 16 travel in the u16 side array, the lowest one (xs = 1) in bit 30 of the index word (indices stay
//     below 2^30: make_table_plan);
//   * the split of the remainder between level 2 (b2 <= 8 bits) and the final level (b3 bits, one block per 2^b3 buckets) is
//     chosen per task so that a final block holds about 6 K entries: wide windows have few entries per bucket (2^26 bases,
//     c = 26: 20), and 128 buckets per block would leave the blocks with 2.5 K entries and 2^18 of them;
//   * the final level keeps bins of up to S3_R3 entries in registers as above; a larger one (c = 26: 512 buckets, 10 K
//     entries) is counted and placed in two passes over an LDS image of the run.
struct S3TGeom {
    int W, prio, c;
    uint32_t sh1, b2, b3, xs, NB1;
};
constexpr uint32_t S3T_IMG = 12288;   // entries of the final level's LDS image (dynamic LDS: 48 KiB)

__global__ __launch_bounds__(S3_THREADS, 4) void k3t_l1_count(const uint32_t* __restrict__ scalars, uint32_t npts, S3TGeom g,
                                                             uint32_t* __restrict__ cnt1) {
    const int prio = g.prio;
    S3_PRIO();
    __shared__ uint32_t hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * S3_CNT_PTS;
    uint32_t end = base + S3_CNT_PTS;
    if (end > npts) end = npts;
    const int cw = g.c;
    for (uint32_t p0 = base + threadIdx.x; p0 < end; p0 += 4 * S3_THREADS) {
        ScalarWords<8> sw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t p = p0 + u * S3_THREADS;
            sw[u].load(scalars, p < end ? p : p0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (p0 + u * S3_THREADS >= end) break;
            uint32_t carry = 0;
            for (int w = 0; w < g.W; ++w) {
                const int d = sw[u].next(cw, (1u << cw) - 1u, 1u << (cw - 1), carry);
                if (d != 0) {
                    const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
                    atomicAdd(&hist[b >> g.sh1], 1u);
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < g.NB1) {
        const uint32_t v = hist[threadIdx.x];
        if (v) atomicAdd(&cnt1[threadIdx.x], v);
    }
}

__global__ __launch_bounds__(S3_THREADS, 4) void k3t_l1_scatter(const uint32_t* __restrict__ scalars, uint32_t npts, S3TGeom g,
                                                               uint32_t* __restrict__ cur1, uint32_t* __restrict__ o_idx,
                                                               uint16_t* __restrict__ o_rem) {
    const int prio = g.prio;
    S3_PRIO();
    extern __shared__ __attribute__((aligned(16))) uint32_t sh[];
    uint32_t* sc = sh;                                          // [8][S3_PB]: word j of the lane's scalar u at j * PB + u * 256 + tid
    uint2* stage = reinterpret_cast<uint2*>(sh + 8 * S3_PB);    // [S3_PB]
    uint32_t* hist = sh + 10 * S3_PB;                           // [256]
    uint32_t* lstart = hist + 256;
    uint32_t* gbase = lstart + 256;
    __shared__ uint32_t wave_tot[4];
    const uint32_t tid = threadIdx.x;
    const uint32_t base = blockIdx.x * (uint32_t)S3_PB;
#pragma unroll
    for (int u = 0; u < S3_T; ++u) {
        const uint32_t p = base + u * S3_THREADS + tid;
        uint4 a = make_uint4(0, 0, 0, 0), b = make_uint4(0, 0, 0, 0);
        if (p < npts) {
            const uint4* q = reinterpret_cast<const uint4*>(scalars) + 2 * (size_t)p;
            a = q[0];
            b = q[1];
        }
        const uint32_t i = u * S3_THREADS + tid;
        sc[0 * S3_PB + i] = a.x; sc[1 * S3_PB + i] = a.y; sc[2 * S3_PB + i] = a.z; sc[3 * S3_PB + i] = a.w;
        sc[4 * S3_PB + i] = b.x; sc[5 * S3_PB + i] = b.y; sc[6 * S3_PB + i] = b.z; sc[7 * S3_PB + i] = b.w;
    }
    const uint32_t cw = (uint32_t)g.c, mask = (1u << cw) - 1u, half = 1u << (cw - 1);
    const uint32_t nb = g.NB1, sh1 = g.sh1, lowmask = (1u << sh1) - 1u, W = (uint32_t)g.W;
    uint32_t carry = 0;   // bit u: the carry of the lane's scalar u into the next window
    for (uint32_t w = 0; w < W; ++w) {
        const uint32_t off = w * cw;
        const uint32_t j = off >> 5, shb = off & 31u;
        hist[tid] = 0;
        __syncthreads();
        uint32_t key[S3_T], rk[S3_T];   // key = low sh1 bits | bin << sh1 | sign << 31;  rk = rank in the bin, ~0 = no entry
#pragma unroll
        for (int u = 0; u < S3_T; ++u) {
            const uint32_t i = u * S3_THREADS + tid;
            const uint32_t lo = j < 8 ? sc[j * S3_PB + i] : 0u;
            const uint32_t hi = j + 1 < 8 ? sc[(j + 1) * S3_PB + i] : 0u;
            const uint32_t raw = (shb ? __builtin_amdgcn_alignbit(hi, lo, shb) : lo) & mask;
            const uint32_t v = raw + ((carry >> u) & 1u);
            int d;
            if (v > half) { d = (int)v - (int)(half << 1); carry |= 1u << u; }
            else { d = (int)v; carry &= ~(1u << u); }
            rk[u] = ~0u;
            key[u] = 0;
            if (d != 0 && base + i < npts) {
                const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
                const uint32_t bin = b >> sh1;
                key[u] = (b & lowmask) | (bin << sh1) | (d < 0 ? 0x80000000u : 0u);
                rk[u] = atomicAdd(&hist[bin], 1u);
            }
        }
        __syncthreads();
        {
            const uint32_t v = tid < nb ? hist[tid] : 0u;
            uint32_t total;
            const uint32_t excl = s3_block_excl_scan(v, wave_tot, &total);
            if (tid < nb) {
                lstart[tid] = excl;
                gbase[tid] = v ? atomicAdd(&cur1[tid], v) : 0u;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < S3_T; ++u) {
                if (rk[u] != ~0u) {
                    const uint32_t bin = (key[u] >> sh1) & 0xffu;
                    const uint32_t p = base + u * S3_THREADS + tid;
                    const uint32_t rem = key[u] & lowmask;
                    // index word: (base * W + window) | remainder bit 0 at bit 30 when the remainder has 17 bits | sign
                    const uint32_t iw = (p * W + w) | (g.xs ? (rem & 1u) << 30 : 0u) | (key[u] & 0x80000000u);
                    stage[lstart[bin] + rk[u]] = make_uint2(iw, (rem >> g.xs) | (bin << 16));
                }
            }
            __syncthreads();
            for (uint32_t slot = tid; slot < total; slot += S3_THREADS) {
                const uint2 e = stage[slot];
                const uint32_t bin = e.y >> 16;
                const uint32_t dst = gbase[bin] + (slot - lstart[bin]);
                o_idx[dst] = e.x;
                o_rem[dst] = (uint16_t)e.y;
            }
        }
        __syncthreads();
    }
}

// level 2: the stored u16 is the remainder >> xs; its level-2 digit is the top b2 bits
__global__ __launch_bounds__(S3_THREADS, 4) void k3t_l2_count(const uint16_t* __restrict__ rem, const uint32_t* __restrict__ off1,
                                                             const uint2* __restrict__ map, const uint32_t* __restrict__ nitems,
                                                             uint32_t* __restrict__ cnt2, S3TGeom g) {
    const int prio = g.prio;
    S3_PRIO();
    __shared__ uint32_t hist[256];
    uint32_t k, lo, hi;
    if (!s3_item(off1, map, nitems, blockIdx.x, S3_SLICE2, k, lo, hi)) return;
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t sh = g.b3 - g.xs;
    uint32_t r[S3_T2];
#pragma unroll
    for (int t = 0; t < S3_T2; ++t) {
        const uint32_t i = lo + t * S3_THREADS + threadIdx.x;
        r[t] = i < hi ? rem[i] : 0xffffffffu;
    }
#pragma unroll
    for (int t = 0; t < S3_T2; ++t)
        if (r[t] != 0xffffffffu) atomicAdd(&hist[r[t] >> sh], 1u);
    __syncthreads();
    if (threadIdx.x < (1u << g.b2)) {
        const uint32_t v = hist[threadIdx.x];
        if (v) atomicAdd(&cnt2[(k << g.b2) + threadIdx.x], v);
    }
}

__global__ __launch_bounds__(S3_THREADS, 4) void k3t_l2_scatter(const uint32_t* __restrict__ i_idx, const uint16_t* __restrict__ i_rem,
                                                               const uint32_t* __restrict__ off1, const uint2* __restrict__ map,
                                                               const uint32_t* __restrict__ nitems, uint32_t* __restrict__ cur2,
                                                               uint32_t* __restrict__ o_idx, uint16_t* __restrict__ o_lo, S3TGeom g) {
    const int prio = g.prio;
    S3_PRIO();
    __shared__ uint2 stage[S3_SLICE2];
    __shared__ uint32_t hist[256], lstart[256], gbase[256];
    __shared__ uint32_t wave_tot[4];
    uint32_t k, lo, hi;
    if (!s3_item(off1, map, nitems, blockIdx.x, S3_SLICE2, k, lo, hi)) return;
    const uint32_t tid = threadIdx.x;
    hist[tid] = 0;
    __syncthreads();
    const uint32_t sh = g.b3 - g.xs, lomask = (1u << sh) - 1u, nb2 = 1u << g.b2;
    uint32_t ex[S3_T2], ky[S3_T2];   // ky = final-level bucket (b3 <= 10 bits) | level-2 digit << 10 | rank << 18; ~0 = no entry
#pragma unroll
    for (int t = 0; t < S3_T2; ++t) {
        const uint32_t i = lo + t * S3_THREADS + tid;
        ky[t] = 0xffffffffu;
        ex[t] = 0;
        if (i < hi) {
            const uint32_t r = i_rem[i];
            const uint32_t iw = i_idx[i];
            const uint32_t h2 = r >> sh;
            const uint32_t low = g.xs ? ((r & lomask) << 1) | ((iw >> 30) & 1u) : (r & lomask);
            ex[t] = iw & 0xbfffffffu;
            const uint32_t rank = atomicAdd(&hist[h2], 1u);   // < 4096
            ky[t] = low | (h2 << 10) | (rank << 18);
        }
    }
    __syncthreads();
    const uint32_t v = tid < nb2 ? hist[tid] : 0u;
    uint32_t total;
    const uint32_t excl = s3_block_excl_scan(v, wave_tot, &total);
    if (tid < nb2) {
        lstart[tid] = excl;
        gbase[tid] = v ? atomicAdd(&cur2[(k << g.b2) + tid], v) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < S3_T2; ++t) {
        if (ky[t] != 0xffffffffu) {
            const uint32_t h2 = (ky[t] >> 10) & 0xffu;
            stage[lstart[h2] + (ky[t] >> 18)] = make_uint2(ex[t], ky[t] & 0x3ffffu);
        }
    }
    __syncthreads();
    for (uint32_t slot = tid; slot < total; slot += S3_THREADS) {
        const uint2 e = stage[slot];
        const uint32_t h2 = (e.y >> 10) & 0xffu;
        const uint32_t dst = gbase[h2] + (slot - lstart[h2]);
        o_idx[dst] = e.x;
        o_lo[dst] = (uint16_t)(e.y & 0x3ffu);
    }
}

// final level: one block per level-2 bin = 2^b3 consecutive buckets (b3 <= 10).  Two builds, chosen by the task's mean bin:
// REGS keeps a bin of up to S3_R3 entries in registers between the count and the placement (one pass over the bin, as
// k3_l3 above; 70 VGPRs); the other one reads the bin twice - count, then place into the LDS image - and holds nothing
// (bins of ~10 K entries: c = 26).  Both fall back to entry-by-entry placement for a bin beyond their capacity (hot buckets).
template <bool REGS>
__global__ __launch_bounds__(S3_THREADS, 4) void k3t_l3(const uint32_t* __restrict__ i_idx, const uint16_t* __restrict__ i_lo,
                                                       const uint32_t* __restrict__ off2, uint32_t* __restrict__ count,
                                                       uint32_t* __restrict__ entries, S3TGeom g, uint32_t img) {
    const int prio = g.prio;
    S3_PRIO();
    extern __shared__ __attribute__((aligned(16))) uint32_t out[];   // [img]
    __shared__ uint32_t hist[1024], cursor[1024];
    __shared__ uint32_t wave_tot[4];
    const uint32_t j = blockIdx.x, tid = threadIdx.x;
    const uint32_t a = off2[j], b = off2[j + 1], s = b - a;
    const uint32_t nb3 = 1u << g.b3;
#pragma unroll
    for (int q = 0; q < 4; ++q) hist[q * S3_THREADS + tid] = 0;
    __syncthreads();
    // exclusive scan of the nb3 <= 1024 bucket counts (four consecutive counters per thread) -> count[], cursor[]
    auto scan_counts = [&]() {
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t t = tid * 4u + q;
            v[q] = t < nb3 ? hist[t] : 0u;
            sum += v[q];
        }
        uint32_t total;
        uint32_t run = s3_block_excl_scan(sum, wave_tot, &total);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t t = tid * 4u + q;
            if (t < nb3) {
                count[((size_t)j << g.b3) + t] = v[q];
                cursor[t] = run;
            }
            run += v[q];
        }
    };
    if (REGS && s <= S3_R3 && s <= img) {
        uint32_t ex[S3_T3], ky[S3_T3];   // ky = bucket (10 bits) | rank in the bucket << 10; ~0 = no entry
#pragma unroll
        for (int t = 0; t < S3_T3; ++t) {
            const uint32_t i = a + t * S3_THREADS + tid;
            ky[t] = 0xffffffffu;
            ex[t] = 0;
            if (i < b) {
                ky[t] = i_lo[i];
                ex[t] = i_idx[i];
            }
        }
#pragma unroll
        for (int t = 0; t < S3_T3; ++t)
            if (ky[t] != 0xffffffffu) ky[t] |= atomicAdd(&hist[ky[t]], 1u) << 10;
        __syncthreads();
        scan_counts();
        __syncthreads();
#pragma unroll
        for (int t = 0; t < S3_T3; ++t)
            if (ky[t] != 0xffffffffu) out[cursor[ky[t] & 1023u] + (ky[t] >> 10)] = ex[t];
        __syncthreads();
        for (uint32_t t = tid; t < s; t += S3_THREADS) entries[a + t] = out[t];
        return;
    }
    for (uint32_t i = a + tid; i < b; i += S3_THREADS) atomicAdd(&hist[i_lo[i]], 1u);
    __syncthreads();
    scan_counts();
    __syncthreads();
    if (s <= img) {
        // second pass over the bin: place into the LDS image (the order inside a bucket is whatever the atomics make it)
        for (uint32_t i = a + tid; i < b; i += S3_THREADS) out[atomicAdd(&cursor[i_lo[i]], 1u)] = i_idx[i];
        __syncthreads();
        for (uint32_t t = tid; t < s; t += S3_THREADS) entries[a + t] = out[t];
    } else {
        for (uint32_t i = a + tid; i < b; i += S3_THREADS) entries[a + atomicAdd(&cursor[i_lo[i]], 1u)] = i_idx[i];
    }
}

static S3TGeom s3t_geometry(const MsmPlan& P, uint32_t npts) {
    S3TGeom g;
    g.W = P.W;
    g.c = P.c;
    g.prio = exp_knob("BLAZE_SORT_PRIO", 3);
    g.sh1 = (uint32_t)(P.c - 1 - 8);
    g.NB1 = (uint32_t)(P.G >> g.sh1);   // 256
    // final-level bins of about 6 K entries (uniform digits: entries per bucket = npts W / G)
    const double per_bucket = (double)npts * P.W / (double)P.G;
    int b3 = 4;
    while (b3 < 10 && (double)(2u << b3) * per_bucket <= 6144.0) ++b3;
    if ((uint32_t)b3 > g.sh1) b3 = (int)g.sh1;
    int b2 = (int)g.sh1 - b3;
    if (b2 > 8) { b2 = 8; b3 = (int)g.sh1 - 8; }
    g.b2 = (uint32_t)b2;
    g.b3 = (uint32_t)b3;
    g.xs = g.sh1 > 16 ? g.sh1 - 16 : 0;
    return g;
}

bool msm_sort3t_ok(const MsmPlan& P) {
    if (!P.table || P.sbits < 1 || P.sbits > 256 || P.c < 16 || P.c > 26 || P.W * P.c < P.sbits + 1) return false;
    if ((uint64_t)P.npts * P.W >= (1ull << 30)) return false;
    const S3TGeom g = s3t_geometry(P, P.npts);
    return g.NB1 <= 256 && g.b2 <= 8 && g.b3 >= 1 && g.b3 <= 10 && g.xs <= 1 && g.b3 > g.xs;
}

int msm_sort3t_max_vgprs() {
    static int cached = -1;
    if (cached >= 0) return cached;
    int mx = 0;
    const void* ks[] = {(const void*)k3t_l1_count, (const void*)k3t_l1_scatter, (const void*)k3t_l2_count, (const void*)k3t_l2_scatter,
                        (const void*)k3t_l3<true>, (const void*)k3t_l3<false>};
    for (const void* k : ks) {
        hipFuncAttributes a;
        if (hipFuncGetAttributes(&a, k) != hipSuccess) {
            (void)hipGetLastError();
            return cached = 0;
        }
        if (a.numRegs > mx) mx = a.numRegs;
    }
    return cached = mx;
}

int msm_sort3t(MsmEngine& E, const void* d_scalars, uint32_t npts) {
    const MsmPlan& P = E.last_plan;
    hipStream_t st = E.sort_st;
    MsmEngine::SortBufs& B = E.sb();
    const S3TGeom g = s3t_geometry(P, P.npts);
    const uint32_t NB1 = g.NB1, NB2 = (uint32_t)(P.G >> g.b3);
    const uint64_t max_entries = (uint64_t)npts * P.W;
    const uint32_t max_items = (uint32_t)(max_entries / S3_SLICE2) + NB1 + 1;
    const uint32_t nsb = (NB2 + 2047u) / 2048u;
    if (nsb > 256) return fail(BLZ_ERR_UNKNOWN, "window-table sort: %u final-level bins exceed the scan's range", NB2);
    const size_t tab_dw = (size_t)3 * (NB1 + 2) + (size_t)3 * (NB2 + 2) + 260 + 4 + 2 * ((size_t)max_items + 2);
    BLZ_TRY(E.sort3_tabs.reserve(tab_dw * 4));
    uint32_t* cnt1 = E.sort3_tabs.as<uint32_t>();
    uint32_t* off1 = cnt1 + NB1 + 2;
    uint32_t* cur1 = off1 + NB1 + 2;
    uint32_t* cnt2 = cur1 + NB1 + 2;
    uint32_t* off2 = cnt2 + NB2 + 2;
    uint32_t* cur2 = off2 + NB2 + 2;
    uint32_t* bsum = cur2 + NB2 + 2;
    uint32_t* nitems = bsum + 260;
    uint2* map = reinterpret_cast<uint2*>(nitems + 4);
    BLZ_TRY(E.inter.reserve(max_entries * 6 + 64));
    BLZ_TRY(E.inter2.reserve(max_entries * 6 + 64));
    uint32_t* i1_idx = E.inter.as<uint32_t>();
    uint16_t* i1_rem = reinterpret_cast<uint16_t*>(i1_idx + max_entries);
    uint32_t* i2_idx = E.inter2.as<uint32_t>();
    uint16_t* i2_lo = reinterpret_cast<uint16_t*>(i2_idx + max_entries);
    const uint32_t* sc = (const uint32_t*)d_scalars;
    const dim3 blk(S3_THREADS);
    // LDS image of a final-level bin: the register path's 7168 entries unless the mean bin is larger than that
    const double mean_bin = (double)max_entries / (double)(NB2 ? NB2 : 1);
    const bool big_bins = mean_bin * 1.1 > (double)S3_R3;
    const uint32_t img = big_bins ? S3T_IMG : S3_R3;

    BLZ_HIP(hipMemsetAsync(cnt1, 0, (size_t)(NB1 + 2) * 4, st), BLZ_ERR_UNKNOWN);
    BLZ_HIP(hipMemsetAsync(cnt2, 0, (size_t)(NB2 + 2) * 4, st), BLZ_ERR_UNKNOWN);
    hipLaunchKernelGGL(k3t_l1_count, dim3((npts + S3_CNT_PTS - 1) / S3_CNT_PTS), blk, 0, st, sc, npts, g, cnt1);
    hipLaunchKernelGGL(k3_scan_small, dim3(1), blk, 0, st, cnt1, NB1, off1, cur1, g.prio);
    const size_t lds1 = (size_t)(10 * S3_PB + 3 * 256) * 4;
    BLZ_TRY(ensure_dynamic_lds((const void*)k3t_l1_scatter, (int)lds1));
    hipLaunchKernelGGL(k3t_l1_scatter, dim3((npts + S3_PB - 1) / S3_PB), blk, lds1, st, sc, npts, g, cur1, i1_idx, i1_rem);
    hipLaunchKernelGGL(k3_slice_map, dim3(1), blk, 0, st, off1, NB1, S3_SLICE2, map, nitems, g.prio);
    hipLaunchKernelGGL(k3t_l2_count, dim3(max_items), blk, 0, st, i1_rem, off1, map, nitems, cnt2, g);
    hipLaunchKernelGGL(k3_scan_a, dim3(nsb), blk, 0, st, cnt2, NB2, bsum, g.prio);
    hipLaunchKernelGGL(k3_scan_b, dim3(1), blk, 0, st, bsum, nsb, g.prio);
    hipLaunchKernelGGL(k3_scan_c, dim3(nsb), blk, 0, st, cnt2, NB2, bsum, off2, cur2, g.prio);
    hipLaunchKernelGGL(k3t_l2_scatter, dim3(max_items), blk, 0, st, i1_idx, i1_rem, off1, map, nitems, cur2, i2_idx, i2_lo, g);
    if (big_bins) {
        BLZ_TRY(ensure_dynamic_lds((const void*)k3t_l3<false>, (int)(img * 4)));
        hipLaunchKernelGGL(k3t_l3<false>, dim3(NB2), blk, (size_t)img * 4, st, i2_idx, i2_lo, off2, B.count.as<uint32_t>(), B.entries.as<uint32_t>(), g, img);
    } else {
        BLZ_TRY(ensure_dynamic_lds((const void*)k3t_l3<true>, (int)(img * 4)));
        hipLaunchKernelGGL(k3t_l3<true>, dim3(NB2), blk, (size_t)img * 4, st, i2_idx, i2_lo, off2, B.count.as<uint32_t>(), B.entries.as<uint32_t>(), g, img);
    }
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

}  // namespace blz
