// Would LOCALITY in time lift the rate of the accumulation's gathers out of a table far beyond the TLB's reach (profiles/r05_tlb_probe.txt:
// 19.7 G lines/s of 64 bytes from 16 GiB on, 57 from 1 GiB)?  Every lane walks K gathers whose j-th address is j / K of the way through
// the table plus a random offset inside a window of W GiB - what a bucket's entries look like when they are ordered by point index -
// (a) with exactly one resident set of lanes (they sweep together), (b) with the grid oversubscribed 8 x (blocks start as others end:
// the sweeps drift apart, as a real kernel's would).  Build: hipcc --offload-arch=gfx950 -O3 tools/sweep_probe.hip -o build/sweep_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

template <int LINE16>
__global__ __launch_bounds__(128, 2) void k_sweep(const uint4* __restrict__ tab, uint64_t nlines, uint64_t window_lines, uint32_t K, uint4* __restrict__ sink) {
    const uint64_t t = (uint64_t)blockIdx.x * 128 + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint64_t z = (t + 1) * 0x9e3779b97f4a7c15ull;
    for (uint32_t j = 0; j < K; ++j) {
        z ^= z >> 29; z *= 0xbf58476d1ce4e5b9ull; z ^= z >> 32;
        uint64_t line = (uint64_t)j * (nlines - window_lines) / K + z % window_lines;
        const uint4* p = tab + line * LINE16;
#pragma unroll
        for (int k = 0; k < LINE16; ++k) { uint4 v = p[k]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
    }
    if (acc.x == 0x12345678u && acc.y == 7u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const uint64_t bytes = 16ull << 30;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { printf("allocation failed\n"); return 1; }
    hipMemset(p, 1, bytes);
    uint4* sink; hipMalloc((void**)&sink, 64);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t resident = 256 * 4 * 2 * 64;   // lanes at two waves per SIMD (the accumulation's occupancy)
    for (int over : {1, 8})
        for (double w : {16.0, 4.0, 1.4, 0.5, 0.125}) {
            const uint32_t K = over == 1 ? 2048 : 256;
            const uint64_t lanes = (uint64_t)resident * over;
            const uint64_t nlines = bytes / 64, wl = (uint64_t)(w * (1ull << 30)) / 64 >= nlines ? nlines - 1 : (uint64_t)(w * (1ull << 30)) / 64;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k_sweep<4>, dim3((unsigned)(lanes / 128)), dim3(128), 0, 0, (const uint4*)p, nlines, wl, K, sink);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double g = (double)lanes * K;
            printf("16 GiB, 64-byte lines, %d x resident lanes, %4u gathers per lane, window %6.3f GiB: %8.2f ms = %6.2f G lines/s\n", over, K, w, best, g / best * 1e-6);
        }
    return 0;
}
