// VERDICT r02 item 6 (measure-or-kill): would a hybrid accumulation - a fraction f of the bucket additions on a
// batched-affine path (fewer multiply-adds, far more memory traffic), the rest on the XYZZ path (multiplier-bound) -
// balance the two pipes of the chip?  This probe gives the hybrid its best case: the affine share is represented by its
// MEMORY TRAFFIC ALONE (no arithmetic at all, a dozen VGPRs: it co-resides with k_accumulate freely), run on its own
// stream underneath the real XYZZ accumulation of the remaining (1 - f) n elements.  If even this lower bound does not
// beat the plain XYZZ kernel on all n elements by 5 ms, no real hybrid can.  tools/hybrid_probe.py drives it.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/hybrid_probe.hip -o build/libhybrid_probe.so
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(256) void k_fill_idx(uint32_t* idx, uint64_t n, uint32_t mask) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        uint64_t z = (i + 0x9e3779b97f4a7c15ull) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        idx[i] = (uint32_t)(z ^ (z >> 31)) & mask;
    }
}

// The first level of a batched-affine pairwise tree over runs of `len` entries (tools/gather_bw.hip, pattern B): forward
// pass gathers both operands of every pair and parks a 56-byte running product, backward pass gathers them again, reads
// the product back and writes the 112-byte sum.  (The upper tree levels - as many additions again - are not even charged.)
__global__ __launch_bounds__(128) void k_affine_traffic(const uint4* __restrict__ pts, const uint32_t* __restrict__ idx, uint64_t nruns,
                                                        uint32_t len, uint4* __restrict__ side, uint4* __restrict__ sink) {
    uint64_t t = (uint64_t)blockIdx.x * 128 + threadIdx.x;
    if (t >= nruns) return;
    const uint32_t* e = idx + t * len;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int pass = 0; pass < 2; ++pass) {
        uint4 nx[7];
        {
            const uint4* p = pts + (uint64_t)e[0] * 8;
#pragma unroll
            for (int k = 0; k < 7; ++k) nx[k] = p[k];
        }
        for (uint32_t j = 0; j < len; ++j) {
            uint4 cur[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) cur[k] = nx[k];
            if (j + 1 < len) {
                const uint4* p = pts + (uint64_t)e[j + 1] * 8;
#pragma unroll
                for (int k = 0; k < 7; ++k) nx[k] = p[k];
            }
#pragma unroll
            for (int k = 0; k < 7; ++k) { acc.x ^= cur[k].x; acc.y += cur[k].y; acc.z ^= cur[k].z; acc.w += cur[k].w; }
            if (j & 1) {
                uint4* s = side + (t * (len / 2) + j / 2) * 11;
                if (pass == 0) { s[0] = acc; s[1] = acc; s[2] = acc; s[3] = acc; }
                else {
                    acc.x ^= s[0].x ^ s[1].y ^ s[2].z ^ s[3].w;
#pragma unroll
                    for (int k = 4; k < 11; ++k) s[k] = acc;
                }
            }
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

static uint4 *g_pts, *g_side, *g_sink;
static uint32_t* g_idx;
static uint64_t g_entries;
static hipStream_t g_st;
static hipEvent_t g_e0, g_e1;
static const uint32_t kLen = 44;   // mean run length of the 2^26 plan

extern "C" int hp_setup(int log_table_points, uint64_t max_entries) {
    const uint64_t npts = 1ull << log_table_points;
    g_entries = max_entries;
    const uint64_t nruns = max_entries / kLen;
    if (hipMalloc(&g_pts, npts * 128) != hipSuccess) return 1;
    if (hipMalloc(&g_idx, max_entries * 4) != hipSuccess) return 2;
    if (hipMalloc(&g_side, nruns * (kLen / 2) * 11 * 16) != hipSuccess) return 3;
    if (hipMalloc(&g_sink, 64) != hipSuccess) return 4;
    hipMemset(g_pts, 1, npts * 128);
    hipLaunchKernelGGL(k_fill_idx, dim3(4096), dim3(256), 0, 0, g_idx, max_entries, (uint32_t)(npts - 1));
    hipStreamCreateWithFlags(&g_st, hipStreamNonBlocking);
    hipEventCreate(&g_e0);
    hipEventCreate(&g_e1);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 5;
}
// enqueue the affine-path traffic of `entries` bucket entries on the probe's own stream; returns at once
extern "C" int hp_launch(uint64_t entries) {
    if (entries > g_entries) return 1;
    const uint64_t nruns = entries / kLen;
    hipEventRecord(g_e0, g_st);
    hipLaunchKernelGGL(k_affine_traffic, dim3((unsigned)((nruns + 127) / 128)), dim3(128), 0, g_st, g_pts, g_idx, nruns, kLen, g_side, g_sink);
    hipEventRecord(g_e1, g_st);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
extern "C" float hp_wait() {
    hipStreamSynchronize(g_st);
    float ms = -1.f;
    hipEventElapsedTime(&ms, g_e0, g_e1);
    return ms;
}
