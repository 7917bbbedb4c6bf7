// What batched-affine bucket accumulation would have to move, measured: Montgomery's trick
// trades ~3.5 field products per addition for memory traffic - every addition reads its two operands twice
// (forward pass: denominators and running products; backward pass: slopes and sums) and parks a running product and a
// result in memory in between.  This tool times the bare memory side on the 2^26 working set:
//   A  805 M random 128-byte line gathers out of an 8 GiB table (what k_accumulate does once per MSM)
//   B  the same gathers done twice (operands of a level re-read on the way back), plus the 56-byte running product
//      written and read back and the 112-byte affine result written, per addition
// No arithmetic at all: these are floors for the memory side, to be set against the multiply-adds saved.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_bw.hip -o build/gather_bw
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_fill_idx(uint32_t* idx, uint64_t n, uint32_t mask) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        uint64_t z = (i + 0x9e3779b97f4a7c15ull) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        idx[i] = (uint32_t)(z ^ (z >> 31)) & mask;
    }
}

// one lane per run of `len` entries (a bucket's run), 2 gathers in flight per lane (the next point prefetched),
// exactly the access pattern of the accumulation kernel; MODE 1 adds the batched-affine side traffic
template <int MODE>
__global__ __launch_bounds__(128) void k_gather(const uint4* __restrict__ pts, const uint32_t* __restrict__ idx, uint64_t nruns,
                                                uint32_t len, uint4* __restrict__ side, uint4* __restrict__ sink) {
    uint64_t t = (uint64_t)blockIdx.x * 128 + threadIdx.x;
    if (t >= nruns) return;
    const uint32_t* e = idx + t * len;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const int passes = MODE ? 2 : 1;
    for (int pass = 0; pass < passes; ++pass) {
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
            if (MODE) {
                // per addition (= per pair of points, so every other entry): 56 B running product out (pass 0) and
                // back in (pass 1), 112 B result out (pass 1)
                if (j & 1) {
                    uint4* s = side + (t * (len / 2) + j / 2) * 11;   // 3.5 + 7 sixteen-byte pieces, rounded to 11
                    if (pass == 0) { s[0] = acc; s[1] = acc; s[2] = acc; s[3] = acc; }
                    else {
                        acc.x ^= s[0].x ^ s[1].y ^ s[2].z ^ s[3].w;
#pragma unroll
                        for (int k = 4; k < 11; ++k) s[k] = acc;
                    }
                }
            }
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

int main(int argc, char** argv) {
    // optional argument: log2 of the table size in points (26 = the 8 GiB table of the 2^26 MSM); the entry count stays
    const int logp = argc > 1 ? atoi(argv[1]) : 26;
    const int nmodes = argc > 2 ? atoi(argv[2]) : 2;   // 1: pattern A only (big tables: B's side buffer is 97 GB)
    const uint64_t npts = 1ull << logp, entries = 12ull << 26;
    printf("table: 2^%d points x 128 B = %.2f GiB\n", logp, (double)npts * 128 / (1ull << 30));
    const uint32_t len = 44;   // mean run length of the 2^26 plan (805 M entries over ~18 M buckets)
    const uint64_t nruns = entries / len;
    uint4 *pts, *side, *sink;
    uint32_t* idx;
    hipMalloc(&pts, npts * 128);
    hipMalloc(&idx, entries * 4);
    hipMalloc(&side, nmodes > 1 ? nruns * (len / 2) * 11 * 16 : 64);
    hipMalloc(&sink, 64);
    hipMemset(pts, 1, npts * 128);
    hipLaunchKernelGGL(k_fill_idx, dim3(4096), dim3(256), 0, 0, idx, entries, (uint32_t)(npts - 1));
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < nmodes; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3((unsigned)((nruns + 127) / 128)), dim3(128), 0, 0, pts, idx, nruns, len, side, sink);
            else hipLaunchKernelGGL(k_gather<1>, dim3((unsigned)((nruns + 127) / 128)), dim3(128), 0, 0, pts, idx, nruns, len, side, sink);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double gathered = (double)nruns * len * 112 * (mode ? 2 : 1);
        const double sidebytes = mode ? (double)nruns * (len / 2) * (64.0 * 2 + 112.0) : 0.0;
        printf("%s: %.2f ms for %.1f GB gathered (128-B lines, 112 B used) + %.1f GB side traffic -> %.2f TB/s of useful bytes\n",
               mode ? "B batched-affine traffic pattern (no arithmetic)" : "A one gather per entry (k_accumulate's pattern, no arithmetic)",
               best, gathered / 1e9, sidebytes / 1e9, (gathered + sidebytes) / best / 1e9);
    }
    return 0;
}
