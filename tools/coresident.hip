// Dev tool: does a small-footprint streaming kernel get scheduled UNDER a register-heavy ALU kernel that leaves
// 112 VGPRs per SIMD free (2 waves x 200 VGPRs, like k_accumulate), when both sit on different HIP streams?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/coresident tools/coresident.hip && /tmp/coresident
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(128) void k_alu(uint64_t* out, int iters) {
    asm volatile("v_mov_b32 v199, 0" ::: "v199");   // claim 200 VGPRs: 2 waves per SIMD
    uint64_t a = threadIdx.x + 1, b = blockIdx.x * 7 + 3;
    uint32_t x = (uint32_t)a * 2654435761u, y = (uint32_t)b | 1u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { a = (uint64_t)x * y + a; b = (uint64_t)y * (uint32_t)a + b; x += (uint32_t)b; }
    }
    if (a + b == 12345) out[0] = a;
}
template <int NV>
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16) {
    if (NV > 0) asm volatile("v_mov_b32 v%0, 0" :: "n"(NV - 1) : );
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

int main(int argc, char** argv) {
    const int prio = argc > 1 ? atoi(argv[1]) : 0;
    hipStream_t sa, sb;
    if (prio) {
        int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        printf("priority range: least %d greatest %d\n", lo, hi);
        CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, prio == 1 ? hi : lo));
        CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, prio == 1 ? lo : hi));
    } else { CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); }
    const size_t bytes = (size_t)1 << 30;
    uint4 *in, *out; uint64_t* o;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMalloc(&o, 64));
    CK(hipMemset(in, 1, bytes));
    hipEvent_t e0, e1, f0, f1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    const int blocksA = 256 * 4 * 64, iters = 3000;
    float ta, tb;
    // alone
    hipLaunchKernelGGL(k_alu, dim3(blocksA), dim3(128), 0, sa, o, 10); CK(hipStreamSynchronize(sa));
    CK(hipEventRecord(e0, sa)); hipLaunchKernelGGL(k_alu, dim3(blocksA), dim3(128), 0, sa, o, iters); CK(hipEventRecord(e1, sa));
    CK(hipStreamSynchronize(sa)); CK(hipEventElapsedTime(&ta, e0, e1));
    printf("ALU kernel alone: %.2f ms\n", ta);
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(f0, sb));
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k_copy<0>, dim3(4096), dim3(256), 0, sb, in, out, bytes / 16);
        CK(hipEventRecord(f1, sb)); CK(hipStreamSynchronize(sb)); CK(hipEventElapsedTime(&tb, f0, f1));
    }
    printf("copy alone: %.2f ms for 8 GiB -> %.0f GB/s (r+w)\n", tb, 16.0 * bytes / tb / 1e6);
    // together: copies are enqueued to last about as long as the ALU kernel
    const int ncopy = (int)(ta / (tb / 8)) + 1;
    CK(hipEventRecord(e0, sa)); hipLaunchKernelGGL(k_alu, dim3(blocksA), dim3(128), 0, sa, o, iters); CK(hipEventRecord(e1, sa));
    CK(hipEventRecord(f0, sb));
    for (int i = 0; i < ncopy; ++i) hipLaunchKernelGGL(k_copy<0>, dim3(4096), dim3(256), 0, sb, in, out, bytes / 16);
    CK(hipEventRecord(f1, sb));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ta, e0, e1)); CK(hipEventElapsedTime(&tb, f0, f1));
    float gap; CK(hipEventElapsedTime(&gap, e0, f1));
    printf("together: ALU %.2f ms; %d copies %.2f ms (%.0f GB/s); both done %.2f ms after the ALU kernel started\n", ta, ncopy, tb,
           2.0 * ncopy * bytes / tb / 1e6, gap);
    return 0;
}
