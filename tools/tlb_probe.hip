// Does the reach of the TLB depend on HOW the table was allocated?  Random 64-byte (BN254 point) / 128-byte (BLS point) line gathers
// out of a table of G GiB from hipMalloc and from hipExtMallocWithFlags(hipDeviceMallocContiguous) - physically contiguous memory
// can be mapped with larger page-table fragments.  Build: hipcc --offload-arch=gfx950 -O3 tools/tlb_probe.hip -o build/tlb_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

template <int LINE16>   // 16-byte pieces per gather: 4 (64 B) or 8 (128 B)
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ tab, uint64_t nlines_mask, uint32_t per_lane, uint4* __restrict__ sink) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint64_t z = (t + 1) * 0x9e3779b97f4a7c15ull;
    for (uint32_t j = 0; j < per_lane; ++j) {
        z ^= z >> 29; z *= 0xbf58476d1ce4e5b9ull; z ^= z >> 32;
        const uint4* p = tab + (z & nlines_mask) * LINE16;
#pragma unroll
        for (int k = 0; k < LINE16; ++k) { uint4 v = p[k]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
    }
    if (acc.x == 0x12345678u && acc.y == 7u) sink[0] = acc;
}

static float run(const uint4* tab, uint64_t bytes, int line, uint64_t gathers, uint4* sink) {
    const uint32_t per_lane = 256;
    const uint64_t lanes = gathers / per_lane;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        if (line == 64) hipLaunchKernelGGL(k_gather<4>, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, tab, bytes / 64 - 1, per_lane, sink);
        else hipLaunchKernelGGL(k_gather<8>, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, tab, bytes / 128 - 1, per_lane, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const uint64_t gathers = 1ull << 29;
    uint4* sink; hipMalloc((void**)&sink, 64);
    for (int a = 1; a < argc; ++a) {
        const uint64_t bytes = (uint64_t)atoi(argv[a]) << 30;
        for (int mode = 0; mode < 2; ++mode) {
            void* p = nullptr;
            hipError_t e = mode ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous) : hipMalloc(&p, bytes);
            if (e != hipSuccess) { printf("%llu GiB %s: allocation failed: %s\n", (unsigned long long)(bytes >> 30), mode ? "contiguous" : "hipMalloc", hipGetErrorString(e)); (void)hipGetLastError(); continue; }
            hipMemset(p, 1, bytes);
            hipDeviceSynchronize();
            for (int line : {64, 128}) {
                const float ms = run((const uint4*)p, bytes, line, gathers, sink);
                printf("%3llu GiB %-10s %3d-byte lines: %8.2f ms for 2^29 gathers = %6.2f G lines/s, %6.1f GB/s\n", (unsigned long long)(bytes >> 30),
                       mode ? "contiguous" : "hipMalloc", line, ms, gathers / ms * 1e-6, gathers * (double)line / ms * 1e-6);
            }
            hipFree(p);
        }
    }
    return 0;
}
