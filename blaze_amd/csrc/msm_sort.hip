// Digit sort of the MSM pipeline: group (point index | sign) entries by (window, bucket).
//
// Two-level LDS-privatised counting sort (replaces one global atomic per entry, which ran at
// ~25 G atomics/s and cost 105 ms of a 2^26 MSM):
//   bucket index (c-1 bits) = coarse (ch bits) | fine (cl bits),  W * 2^ch <= 24576 coarse bins
//   k_coarse_count    per block: LDS histogram of its points' digits over all coarse bins, flushed
//                     with one global atomic per (block, non-empty bin)
//   k_coarse_scan     exclusive scan over the coarse bins (one block)
//   k_coarse_scatter  per block: LDS count, one global reservation per (block, bin), then every
//                     entry gets base + LDS rank; writes (entry, fine) pairs grouped by coarse bin
//   k_fine_count      per (coarse bin, slice): LDS histogram over the 2^cl fine buckets -> count[]
//   (k_scan_* of msm.hip: bucket offsets + unit offsets, unchanged)
//   k_fine_scatter    per (coarse bin, slice): LDS count, reservation per (block, bucket) on the
//                     scatter cursor, entries written to their final bucket slice
// HBM traffic: scalars read 3x (32 B each), 8 B/entry written + read twice, 4 B/entry written.
#include "msm_engine.hpp"
#include "msm_digits.cuh"

namespace blz {

constexpr int SORT_THREADS = 1024;
constexpr int FINE_THREADS = 512;

struct SortGeom {
    int c, W, ch, cl;      // window bits, windows, coarse / fine bits of the bucket index
    uint32_t Bw;           // buckets per window
    uint32_t NC;           // W << ch coarse bins
    uint32_t pts_per_block;
};

template <int SW>
__global__ __launch_bounds__(SORT_THREADS) void k_coarse_count(const uint32_t* __restrict__ scalars, uint32_t npts, SortGeom g,
                                                               uint32_t* __restrict__ coarse_count) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sh[];
    for (uint32_t i = threadIdx.x; i < g.NC; i += SORT_THREADS) sh[i] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * g.pts_per_block;
    uint32_t end = base + g.pts_per_block;
    if (end > npts) end = npts;
    const uint32_t mask = (1u << g.c) - 1u, half = 1u << (g.c - 1);
    for (uint32_t p = base + threadIdx.x; p < end; p += SORT_THREADS) {
        ScalarWords<SW> sw;
        sw.load(scalars, p);
        uint32_t carry = 0;
        for (int w = 0; w < g.W; ++w) {
            int d = sw.next(g.c, mask, half, carry);
            if (d != 0) {
                uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
                atomicAdd(&sh[((uint32_t)w << g.ch) + (b >> g.cl)], 1u);
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
                                                                 uint2* __restrict__ inter) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sh[];
    for (uint32_t i = threadIdx.x; i < g.NC; i += SORT_THREADS) sh[i] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * g.pts_per_block;
    uint32_t end = base + g.pts_per_block;
    if (end > npts) end = npts;
    const uint32_t mask = (1u << g.c) - 1u, half = 1u << (g.c - 1);
    for (uint32_t p = base + threadIdx.x; p < end; p += SORT_THREADS) {
        ScalarWords<SW> sw;
        sw.load(scalars, p);
        uint32_t carry = 0;
        for (int w = 0; w < g.W; ++w) {
            int d = sw.next(g.c, mask, half, carry);
            if (d != 0) {
                uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
                atomicAdd(&sh[((uint32_t)w << g.ch) + (b >> g.cl)], 1u);
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
    for (uint32_t p = base + threadIdx.x; p < end; p += SORT_THREADS) {
        ScalarWords<SW> sw;
        sw.load(scalars, p);
        uint32_t carry = 0;
        for (int w = 0; w < g.W; ++w) {
            int d = sw.next(g.c, mask, half, carry);
            if (d != 0) {
                uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
                uint32_t pos = atomicAdd(&sh[((uint32_t)w << g.ch) + (b >> g.cl)], 1u);
                inter[pos] = make_uint2(p | (d < 0 ? 0x80000000u : 0u), b & fmask);
            }
        }
    }
}

// slice s of S of coarse bin k: entries [lo, hi)
__device__ __forceinline__ void slice_range(const uint32_t* coarse_off, uint32_t k, uint32_t s, uint32_t S, uint32_t& lo,
                                            uint32_t& hi) {
    uint32_t a = coarse_off[k], b = coarse_off[k + 1];
    uint32_t per = (b - a + S - 1) / S;
    lo = a + s * per;
    hi = lo + per;
    if (lo > b) lo = b;
    if (hi > b) hi = b;
}

__global__ __launch_bounds__(FINE_THREADS) void k_fine_count(const uint2* __restrict__ inter, const uint32_t* __restrict__ coarse_off,
                                                             int cl, uint32_t S, uint32_t* __restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sh[];
    const uint32_t nf = 1u << cl;
    const uint32_t k = blockIdx.x / S, s = blockIdx.x % S;
    uint32_t lo, hi;
    slice_range(coarse_off, k, s, S, lo, hi);
    if (lo >= hi) return;
    for (uint32_t i = threadIdx.x; i < nf; i += FINE_THREADS) sh[i] = 0;
    __syncthreads();
    for (uint32_t j = lo + threadIdx.x; j < hi; j += FINE_THREADS) atomicAdd(&sh[inter[j].y], 1u);
    __syncthreads();
    uint32_t* dst = count + ((size_t)k << cl);
    for (uint32_t i = threadIdx.x; i < nf; i += FINE_THREADS) {
        uint32_t v = sh[i];
        if (v) {
            if (S == 1) dst[i] = v;
            else atomicAdd(&dst[i], v);
        }
    }
}

__global__ __launch_bounds__(FINE_THREADS) void k_fine_scatter(const uint2* __restrict__ inter, const uint32_t* __restrict__ coarse_off,
                                                               int cl, uint32_t S, uint32_t* __restrict__ cursor,
                                                               uint32_t* __restrict__ entries) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sh[];
    const uint32_t nf = 1u << cl;
    const uint32_t k = blockIdx.x / S, s = blockIdx.x % S;
    uint32_t lo, hi;
    slice_range(coarse_off, k, s, S, lo, hi);
    if (lo >= hi) return;
    for (uint32_t i = threadIdx.x; i < nf; i += FINE_THREADS) sh[i] = 0;
    __syncthreads();
    for (uint32_t j = lo + threadIdx.x; j < hi; j += FINE_THREADS) atomicAdd(&sh[inter[j].y], 1u);
    __syncthreads();
    uint32_t* cur = cursor + ((size_t)k << cl);
    for (uint32_t i = threadIdx.x; i < nf; i += FINE_THREADS) {
        uint32_t v = sh[i];
        sh[i] = v ? atomicAdd(&cur[i], v) : 0u;
    }
    __syncthreads();
    for (uint32_t j = lo + threadIdx.x; j < hi; j += FINE_THREADS) {
        uint2 e = inter[j];
        uint32_t pos = atomicAdd(&sh[e.y], 1u);
        entries[pos] = e.x;
    }
}

// coarse / fine split of the bucket index
static SortGeom make_geom(const MsmPlan& P) {
    SortGeom g;
    g.c = P.c;
    g.W = P.W;
    g.Bw = P.Bw;
    int cb = P.c - 1;
    int cl = cb < 10 ? cb : 10;
    int ch = cb - cl;
    while (((uint32_t)P.W << ch) > 24576u && ch > 0) { --ch; ++cl; }
    g.ch = ch;
    g.cl = cl;
    g.NC = (uint32_t)P.W << ch;
    g.pts_per_block = 0;
    return g;
}

int msm_sort_lds(MsmEngine& E, const void* d_scalars, uint32_t npts, int sbits) {
    const MsmPlan& P = E.last_plan;
    hipStream_t st = E.stream;
    SortGeom g = make_geom(P);
    if (g.cl > 12 || g.NC > 24576u) return fail(BLZ_ERR_UNKNOWN, "sort geometry out of range (c=%d W=%d)", P.c, P.W);
    // points per block: enough entries per block to amortise the NC-sized LDS sweeps, enough blocks to fill the chip
    uint32_t ppb = (uint32_t)(((uint64_t)g.NC * 16 + P.W - 1) / P.W);
    if (ppb < 4096) ppb = 4096;
    if (ppb > 65536) ppb = 65536;
    ppb = (ppb + SORT_THREADS - 1) / SORT_THREADS * SORT_THREADS;
    g.pts_per_block = ppb;
    const uint32_t nblk = (npts + ppb - 1) / ppb;
    const uint64_t max_entries = (uint64_t)npts * P.W;
    BLZ_TRY(E.coarse.reserve(((size_t)g.NC * 2 + 2) * 4));
    BLZ_TRY(E.inter.reserve(max_entries * 8));
    uint32_t* coarse_count = E.coarse.as<uint32_t>();
    uint32_t* coarse_off = coarse_count + g.NC;
    BLZ_HIP(hipMemsetAsync(coarse_count, 0, (size_t)g.NC * 4, st), BLZ_ERR_UNKNOWN);
    const size_t lds = (size_t)g.NC * 4;
    static bool attr_done = false;
    if (!attr_done) {
        BLZ_HIP(hipFuncSetAttribute((const void*)k_coarse_count<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024), BLZ_ERR_UNKNOWN);
        BLZ_HIP(hipFuncSetAttribute((const void*)k_coarse_count<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024), BLZ_ERR_UNKNOWN);
        BLZ_HIP(hipFuncSetAttribute((const void*)k_coarse_scatter<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024), BLZ_ERR_UNKNOWN);
        BLZ_HIP(hipFuncSetAttribute((const void*)k_coarse_scatter<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024), BLZ_ERR_UNKNOWN);
        attr_done = true;
    }
    const uint32_t* sc = (const uint32_t*)d_scalars;
    if (sbits == 256) hipLaunchKernelGGL(k_coarse_count<8>, dim3(nblk), dim3(SORT_THREADS), lds, st, sc, npts, g, coarse_count);
    else hipLaunchKernelGGL(k_coarse_count<1>, dim3(nblk), dim3(SORT_THREADS), lds, st, sc, npts, g, coarse_count);
    hipLaunchKernelGGL(k_coarse_scan, dim3(1), dim3(1024), 0, st, coarse_count, g.NC, coarse_off);
    if (sbits == 256)
        hipLaunchKernelGGL(k_coarse_scatter<8>, dim3(nblk), dim3(SORT_THREADS), lds, st, sc, npts, g, coarse_count, E.inter.as<uint2>());
    else
        hipLaunchKernelGGL(k_coarse_scatter<1>, dim3(nblk), dim3(SORT_THREADS), lds, st, sc, npts, g, coarse_count, E.inter.as<uint2>());
    // slices per coarse bin from the mean bin size (no host sync); a block loops over whatever its slice holds
    uint64_t mean = max_entries / g.NC + 1;
    uint32_t S = (uint32_t)((4 * mean + 65535) / 65536);
    if (S < 1) S = 1;
    if (S > 4096) S = 4096;
    E.sort_slices = S;
    E.sort_cl = g.cl;
    E.sort_nc = g.NC;
    hipLaunchKernelGGL(k_fine_count, dim3(g.NC * S), dim3(FINE_THREADS), (size_t)4 << g.cl, st, E.inter.as<uint2>(), coarse_off,
                       g.cl, S, E.count.as<uint32_t>());
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

int msm_sort_lds_scatter(MsmEngine& E) {
    hipStream_t st = E.stream;
    uint32_t* coarse_off = E.coarse.as<uint32_t>() + E.sort_nc;
    hipLaunchKernelGGL(k_fine_scatter, dim3(E.sort_nc * E.sort_slices), dim3(FINE_THREADS), (size_t)4 << E.sort_cl, st,
                       E.inter.as<uint2>(), coarse_off, E.sort_cl, E.sort_slices, E.count.as<uint32_t>(), E.entries.as<uint32_t>());
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

}  // namespace blz
