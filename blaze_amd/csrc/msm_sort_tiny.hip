// The whole sort stage of a SMALL task in one block.
//
// The reference's own tests run 8192 elements (tests/integration_msm.rs MSM_SIZE).  At that size the pipeline's sort stage -
// zero, two-level LDS sort (count / scan / scatter, twice), three scan kernels, the unit lists: fourteen launches of
// kernels built for 2^26 elements - took 0.64 ms of a 4 ms MSM, nearly all of it launch-to-launch latency.  Here one block of
// 1024 lanes does it all for tasks whose bucket space fits its LDS:
//   A  signed window digits of every scalar -> LDS histogram over the (flat) bucket space
//   B  exclusive scan of (entries, units) per bucket -> off[], unit_off[] (global), the histogram becomes the cursor
//   C  digits again -> entries[] (index | sign) at the cursor
//   D  unit -> bucket map and the units in order of descending run length (counting sort over the <= L + 1 lengths), the
//      length histogram k_combine_units reads, stats (units, largest bucket, entries)
// Same outputs, bit for bit the same buckets as the big path (the order of a bucket's entries differs; the group law does
// not care).  Chosen by msm.hip run() when msm_sort_tiny_ok().
#include "msm_engine.hpp"
#include "msm_digits.hip.hpp"

namespace blz {

constexpr int TINY_THREADS = 1024;
constexpr uint32_t TINY_MAX_G = 24576;        // 96 KiB of LDS for the histogram / cursor
constexpr uint32_t TINY_MAX_PTS = 20480;      // scalars (pf = 8: 32-bit chunks) one block walks twice: 0.25 ms at 2^13, 0.46 at 2^14, 0.87 at 2^15 - where the big path (0.6 ms flat) is the faster one again
constexpr uint32_t TINY_MAX_L = 1024;
constexpr uint32_t TINY_ONE_BLOCK_PTS = 2048;  // up to here the one-block kernel (one launch, histogram in the LDS) is the faster one

struct TinyGeom {
    int W;
    uint32_t G, L;
    uint8_t width[MSM_MAX_W];
    uint32_t boff[MSM_MAX_W + 1];
};

// exclusive scan of one u64 per thread over the block (1024 threads = 16 waves); returns the prefix, *total the sum
__device__ __forceinline__ uint64_t tiny_block_scan(uint64_t v, uint64_t* wave_tot, uint64_t* total) {
    uint64_t incl = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t t = __shfl_up(incl, o, 64);
        if ((threadIdx.x & 63u) >= (uint32_t)o) incl += t;
    }
    const uint32_t wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 63u) wave_tot[wv] = incl;
    __syncthreads();
    uint64_t base = 0, tot = 0;
    for (uint32_t q = 0; q < TINY_THREADS / 64; ++q) {
        const uint64_t t = wave_tot[q];
        if (q < wv) base += t;
        tot += t;
    }
    *total = tot;
    __syncthreads();
    return base + incl - v;
}

template <int SW>
__global__ __launch_bounds__(TINY_THREADS) void k_sort_tiny(const uint32_t* __restrict__ scalars, uint32_t npts, TinyGeom g,
                                                            uint32_t* __restrict__ count, uint32_t* __restrict__ off,
                                                            uint32_t* __restrict__ unit_off, uint32_t* __restrict__ entries,
                                                            uint32_t* __restrict__ unit_bucket, uint32_t* __restrict__ unit_order,
                                                            uint32_t* __restrict__ lenhist, uint32_t* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cnt[];   // [G]: histogram, then cursor
    __shared__ uint64_t wave_tot[TINY_THREADS / 64];
    __shared__ uint32_t lh[TINY_MAX_L + 1], lcur[TINY_MAX_L + 1];
    __shared__ uint32_t mx_sh;
    const uint32_t tid = threadIdx.x, G = g.G, L = g.L;
    for (uint32_t i = tid; i < G; i += TINY_THREADS) cnt[i] = 0;
    for (uint32_t i = tid; i <= L; i += TINY_THREADS) lh[i] = 0;
    if (tid == 0) mx_sh = 0;
    __syncthreads();
    // ---- A: histogram
    for (uint32_t p = tid; p < npts; p += TINY_THREADS) {
        ScalarWords<SW> sw;
        sw.load(scalars, p);
        uint32_t carry = 0;
        for (int w = 0; w < g.W; ++w) {
            const int cw = g.width[w];
            const int d = sw.next(cw, (1u << cw) - 1u, 1u << (cw - 1), carry);
            if (d != 0) atomicAdd(&cnt[g.boff[w] + (uint32_t)(d < 0 ? -d : d) - 1u], 1u);
        }
    }
    __syncthreads();
    // ---- B: scan.  Thread t owns the buckets [t chunk, (t + 1) chunk); (entries, units) packed in a u64 as in msm.hip
    const uint32_t chunk = (G + TINY_THREADS - 1) / TINY_THREADS;
    const uint32_t g0 = tid * chunk, g1 = g0 + chunk < G ? g0 + chunk : G;
    uint64_t mine = 0;
    uint32_t mx = 0;
    for (uint32_t b = g0; b < g1; ++b) {
        const uint32_t c = cnt[b];
        mine += (uint64_t)c | ((uint64_t)((c + L - 1) / L) << 32);
        mx = c > mx ? c : mx;
    }
    if (mx) atomicMax(&mx_sh, mx);
    uint64_t total;
    const uint64_t run0 = tiny_block_scan(mine, wave_tot, &total);
    uint64_t run = run0;
    for (uint32_t b = g0; b < g1; ++b) {
        const uint32_t c = cnt[b];
        const uint32_t o = (uint32_t)(run & 0xffffffffu), u = (uint32_t)(run >> 32);
        off[b] = o;
        unit_off[b] = u;
        count[b] = o;
        cnt[b] = o;   // cursor
        // unit lengths of this bucket: c / L full units and one of c % L entries
        const uint32_t nfull = c / L, rem = c - nfull * L;
        if (nfull) atomicAdd(&lh[L], nfull);
        if (rem) atomicAdd(&lh[rem], 1u);
        for (uint32_t k = 0; k < nfull + (rem ? 1u : 0u); ++k) unit_bucket[u + k] = b;
        run += (uint64_t)c | ((uint64_t)((c + L - 1) / L) << 32);
    }
    if (tid == 0) {
        off[G] = (uint32_t)(total & 0xffffffffu);
        unit_off[G] = (uint32_t)(total >> 32);
    }
    __syncthreads();
    // ---- C: entries
    for (uint32_t p = tid; p < npts; p += TINY_THREADS) {
        ScalarWords<SW> sw;
        sw.load(scalars, p);
        uint32_t carry = 0;
        for (int w = 0; w < g.W; ++w) {
            const int cw = g.width[w];
            const int d = sw.next(cw, (1u << cw) - 1u, 1u << (cw - 1), carry);
            if (d != 0) {
                const uint32_t pos = atomicAdd(&cnt[g.boff[w] + (uint32_t)(d < 0 ? -d : d) - 1u], 1u);
                entries[pos] = p | (d < 0 ? 0x80000000u : 0u);
            }
        }
    }
    // ---- D: units by descending length.  lcur[len] = number of units strictly longer than len
    if (tid == 0) {
        uint32_t r = 0;
        for (int len = (int)L; len >= 0; --len) {
            lcur[len] = r;
            r += lh[len];
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i <= L; i += TINY_THREADS) {
        lenhist[i] = lh[i];
        lenhist[TINY_MAX_L + 1 + i] = lcur[i] + lh[i];   // (the big path leaves its end-of-bin cursors there; nobody reads them)
    }
    run = run0;
    for (uint32_t b = g0; b < g1; ++b) {
        // (cnt[b] is the END of bucket b's run now; its start and first unit are the scan's running values again - not read back
        // from off[] / unit_off[]: a dependent global load per bucket, ~2 us each on this one-block kernel)
        const uint32_t o = (uint32_t)(run & 0xffffffffu), c = cnt[b] - o, u0 = (uint32_t)(run >> 32);
        const uint32_t nfull = c / L, rem = c - nfull * L;
        run += (uint64_t)c | ((uint64_t)((c + L - 1) / L) << 32);
        if (nfull) {
            const uint32_t pos = atomicAdd(&lcur[L], nfull);
            for (uint32_t k = 0; k < nfull; ++k) unit_order[pos + k] = u0 + k;
        }
        if (rem) unit_order[atomicAdd(&lcur[rem], 1u)] = u0 + nfull;
    }
    if (tid == 0) {
        stats[0] = (uint32_t)(total >> 32);
        stats[1] = mx_sh;
        stats[2] = (uint32_t)(total & 0xffffffffu);
    }
}

// The same stage for the upper half of the range (more than TINY_ONE_BLOCK_PTS scalars): the digit walks - 8 scalars x 26 windows
// per lane, twice, ~30 instructions per digit - were 0.2 of the 0.24 ms on the ONE CU the block runs on; here they spread
// over up to 16 blocks with the histogram and the cursors in global memory (k_tiny_hist, k_tiny_place), and one block does
// the scans and the unit lists in between (k_tiny_scan: phases B and D above, from the global histogram).  Three launches.
template <int SW>
__global__ __launch_bounds__(TINY_THREADS) void k_tiny_hist(const uint32_t* __restrict__ scalars, uint32_t npts, TinyGeom g, uint32_t* __restrict__ count) {
    for (uint32_t p = blockIdx.x * TINY_THREADS + threadIdx.x; p < npts; p += gridDim.x * TINY_THREADS) {
        ScalarWords<SW> sw;
        sw.load(scalars, p);
        uint32_t carry = 0;
        for (int w = 0; w < g.W; ++w) {
            const int cw = g.width[w];
            const int d = sw.next(cw, (1u << cw) - 1u, 1u << (cw - 1), carry);
            if (d != 0) atomicAdd(&count[g.boff[w] + (uint32_t)(d < 0 ? -d : d) - 1u], 1u);
        }
    }
}
template <int SW>
__global__ __launch_bounds__(TINY_THREADS) void k_tiny_place(const uint32_t* __restrict__ scalars, uint32_t npts, TinyGeom g, uint32_t* __restrict__ cursor,
                                                             uint32_t* __restrict__ entries) {
    for (uint32_t p = blockIdx.x * TINY_THREADS + threadIdx.x; p < npts; p += gridDim.x * TINY_THREADS) {
        ScalarWords<SW> sw;
        sw.load(scalars, p);
        uint32_t carry = 0;
        for (int w = 0; w < g.W; ++w) {
            const int cw = g.width[w];
            const int d = sw.next(cw, (1u << cw) - 1u, 1u << (cw - 1), carry);
            if (d != 0) {
                const uint32_t pos = atomicAdd(&cursor[g.boff[w] + (uint32_t)(d < 0 ? -d : d) - 1u], 1u);
                entries[pos] = p | (d < 0 ? 0x80000000u : 0u);
            }
        }
    }
}
// count[] holds the histogram on entry and the buckets' start offsets (the cursors of k_tiny_place) on exit
__global__ __launch_bounds__(TINY_THREADS) void k_tiny_scan(TinyGeom g, uint32_t* __restrict__ count, uint32_t* __restrict__ off,
                                                            uint32_t* __restrict__ unit_off, uint32_t* __restrict__ unit_bucket,
                                                            uint32_t* __restrict__ unit_order, uint32_t* __restrict__ lenhist, uint32_t* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cnt[];   // [G]: the histogram
    __shared__ uint64_t wave_tot[TINY_THREADS / 64];
    __shared__ uint32_t lh[TINY_MAX_L + 1], lcur[TINY_MAX_L + 1];
    __shared__ uint32_t mx_sh;
    const uint32_t tid = threadIdx.x, G = g.G, L = g.L;
    for (uint32_t i = tid; i < G; i += TINY_THREADS) cnt[i] = count[i];
    for (uint32_t i = tid; i <= L; i += TINY_THREADS) lh[i] = 0;
    if (tid == 0) mx_sh = 0;
    __syncthreads();
    const uint32_t chunk = (G + TINY_THREADS - 1) / TINY_THREADS;
    const uint32_t g0 = tid * chunk, g1 = g0 + chunk < G ? g0 + chunk : G;
    uint64_t mine = 0;
    uint32_t mx = 0;
    for (uint32_t b = g0; b < g1; ++b) {
        const uint32_t c = cnt[b];
        mine += (uint64_t)c | ((uint64_t)((c + L - 1) / L) << 32);
        mx = c > mx ? c : mx;
    }
    if (mx) atomicMax(&mx_sh, mx);
    uint64_t total;
    const uint64_t run0 = tiny_block_scan(mine, wave_tot, &total);
    uint64_t run = run0;
    for (uint32_t b = g0; b < g1; ++b) {
        const uint32_t c = cnt[b];
        const uint32_t o = (uint32_t)(run & 0xffffffffu), u = (uint32_t)(run >> 32);
        off[b] = o;
        unit_off[b] = u;
        count[b] = o;   // cursor
        const uint32_t nfull = c / L, rem = c - nfull * L;
        if (nfull) atomicAdd(&lh[L], nfull);
        if (rem) atomicAdd(&lh[rem], 1u);
        for (uint32_t k = 0; k < nfull + (rem ? 1u : 0u); ++k) unit_bucket[u + k] = b;
        run += (uint64_t)c | ((uint64_t)((c + L - 1) / L) << 32);
    }
    if (tid == 0) {
        off[G] = (uint32_t)(total & 0xffffffffu);
        unit_off[G] = (uint32_t)(total >> 32);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t r = 0;
        for (int len = (int)L; len >= 0; --len) {
            lcur[len] = r;
            r += lh[len];
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i <= L; i += TINY_THREADS) {
        lenhist[i] = lh[i];
        lenhist[TINY_MAX_L + 1 + i] = lcur[i] + lh[i];
    }
    run = run0;
    for (uint32_t b = g0; b < g1; ++b) {
        const uint32_t c = cnt[b], u0 = (uint32_t)(run >> 32);
        const uint32_t nfull = c / L, rem = c - nfull * L;
        if (nfull) {
            const uint32_t pos = atomicAdd(&lcur[L], nfull);
            for (uint32_t k = 0; k < nfull; ++k) unit_order[pos + k] = u0 + k;
        }
        if (rem) unit_order[atomicAdd(&lcur[rem], 1u)] = u0 + nfull;
        run += (uint64_t)c | ((uint64_t)((c + L - 1) / L) << 32);
    }
    if (tid == 0) {
        stats[0] = (uint32_t)(total >> 32);
        stats[1] = mx_sh;
        stats[2] = (uint32_t)(total & 0xffffffffu);
    }
}

bool msm_sort_tiny_ok(const MsmPlan& P, uint32_t npts, int sbits) {
    if (exp_knob("BLAZE_SORT_TINY", 1) == 0) return false;
    if (P.table || P.W < 1 || (sbits != 256 && sbits != 64 && sbits != 32)) return false;
    if (P.G == 0 || P.G > TINY_MAX_G || npts > TINY_MAX_PTS || P.L < 1 || P.L > TINY_MAX_L) return false;
    return true;
}

int msm_sort_tiny(MsmEngine& E, const void* d_scalars, uint32_t npts, int sbits, uint32_t max_units) {
    const MsmPlan& P = E.last_plan;
    hipStream_t st = E.sort_st;
    MsmEngine::SortBufs& B = E.sb();
    TinyGeom g;
    g.W = P.W;
    g.G = (uint32_t)P.G;
    g.L = P.L;
    for (int w = 0; w < P.W; ++w) {
        g.width[w] = P.width[w];
        g.boff[w] = P.boff[w];
    }
    g.boff[P.W] = P.boff[P.W];
    BLZ_TRY(B.unit_bucket.reserve(((size_t)max_units + 1) * 4));
    BLZ_TRY(B.unit_order.reserve(((size_t)max_units + 1) * 4));
    BLZ_TRY(B.lenhist.reserve(2 * (TINY_MAX_L + 1) * 4));
    const size_t lds = (size_t)g.G * 4;
    const uint32_t* sc = (const uint32_t*)d_scalars;
    if (npts > TINY_ONE_BLOCK_PTS && exp_knob("BLAZE_SORT_TINY_BLOCKS", 1) != 0) {
        uint32_t nb = (npts + TINY_THREADS - 1) / TINY_THREADS;
        if (nb > 16) nb = 16;
        uint32_t* count = B.count.as<uint32_t>();
        BLZ_HIP(hipMemsetAsync(count, 0, (size_t)g.G * 4, st), BLZ_ERR_UNKNOWN);
        BLZ_TRY(ensure_dynamic_lds((const void*)k_tiny_scan, (int)lds));
        BLZ_SW_DISPATCH(sbits, hipLaunchKernelGGL(k_tiny_hist<SW>, dim3(nb), dim3(TINY_THREADS), 0, st, sc, npts, g, count));
        hipLaunchKernelGGL(k_tiny_scan, dim3(1), dim3(TINY_THREADS), lds, st, g, count, B.off.as<uint32_t>(), B.unit_off.as<uint32_t>(),
                           B.unit_bucket.as<uint32_t>(), B.unit_order.as<uint32_t>(), B.lenhist.as<uint32_t>(), B.stats.as<uint32_t>());
        BLZ_SW_DISPATCH(sbits, hipLaunchKernelGGL(k_tiny_place<SW>, dim3(nb), dim3(TINY_THREADS), 0, st, sc, npts, g, count, B.entries.as<uint32_t>()));
    } else {
        BLZ_SW_DISPATCH(sbits, {
            BLZ_TRY(ensure_dynamic_lds((const void*)k_sort_tiny<SW>, (int)lds));
            hipLaunchKernelGGL(k_sort_tiny<SW>, dim3(1), dim3(TINY_THREADS), lds, st, sc, npts, g, B.count.as<uint32_t>(), B.off.as<uint32_t>(),
                               B.unit_off.as<uint32_t>(), B.entries.as<uint32_t>(), B.unit_bucket.as<uint32_t>(), B.unit_order.as<uint32_t>(),
                               B.lenhist.as<uint32_t>(), B.stats.as<uint32_t>());
        });
    }
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

}  // namespace blz
