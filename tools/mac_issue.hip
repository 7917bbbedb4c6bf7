// Issue cost of the product-scanning MAC step on gfx950, in shader cycles (s_memtime), measured the
// way the multiplier uses it: batches of 4 v_mad_u64_u32 (carry -> rotating SGPR pairs) followed by
// their 4 v_addc_co_u32, all on one 96-bit accumulator.  Run with 1, 2, 3 waves per SIMD: the
// per-wave time divided by the waves per SIMD is the throughput cost per instruction pair.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mac_issue.hip -o mac_issue
#include <hip/hip_runtime.h>
#include "field.hip.hpp"   // -I blaze_amd/csrc: the real multiplier, timed the same way
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

constexpr int REPS = 32768;  // batches per measurement (each: 4 MAD + 4 ADDC)
constexpr int MUL_REPS = 2048;

template <int MODE>  // 0: MAD+ADDC batches, 1: MADs only, 2: ADDCs only (v_addc on vcc), 3: v_add_u32 only
__global__ __launch_bounds__(64) void k_issue(uint64_t* out, uint32_t seed) {
    uint32_t x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u;
    uint64_t alo = x, b1 = y, b2 = x + 1, b3 = y + 1;
    uint32_t ahi = 0, h1 = 1, h2 = 2, h3 = 3;
    uint64_t c0, c1, c2, c3;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
        if constexpr (MODE == 0) {
            asm volatile(
                "v_mad_u64_u32 %[alo], %[c0], %[x], %[y], %[alo]\n\t"
                "v_mad_u64_u32 %[alo], %[c1], %[y], %[x], %[alo]\n\t"
                "v_mad_u64_u32 %[alo], %[c2], %[x], %[x], %[alo]\n\t"
                "v_mad_u64_u32 %[alo], %[c3], %[y], %[y], %[alo]\n\t"
                "v_addc_co_u32 %[ahi], %[c0], 0, %[ahi], %[c0]\n\t"
                "v_addc_co_u32 %[ahi], %[c1], 0, %[ahi], %[c1]\n\t"
                "v_addc_co_u32 %[ahi], %[c2], 0, %[ahi], %[c2]\n\t"
                "v_addc_co_u32 %[ahi], %[c3], 0, %[ahi], %[c3]\n\t"
                : [alo] "+&v"(alo), [ahi] "+&v"(ahi), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3)
                : [x] "v"(x), [y] "v"(y));
        } else if constexpr (MODE == 1) {
            asm volatile(
                "v_mad_u64_u32 %[alo], %[c0], %[x], %[y], %[alo]\n\t"
                "v_mad_u64_u32 %[alo], %[c1], %[y], %[x], %[alo]\n\t"
                "v_mad_u64_u32 %[alo], %[c2], %[x], %[x], %[alo]\n\t"
                "v_mad_u64_u32 %[alo], %[c3], %[y], %[y], %[alo]\n\t"
                : [alo] "+&v"(alo), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3)
                : [x] "v"(x), [y] "v"(y));
        } else if constexpr (MODE == 2) {
            asm volatile(
                "v_addc_co_u32 %[ahi], vcc, 0, %[ahi], vcc\n\t"
                "v_addc_co_u32 %[ahi], vcc, 0, %[ahi], vcc\n\t"
                "v_addc_co_u32 %[ahi], vcc, 0, %[ahi], vcc\n\t"
                "v_addc_co_u32 %[ahi], vcc, 0, %[ahi], vcc\n\t"
                : [ahi] "+&v"(ahi) : : "vcc");
        } else if constexpr (MODE == 5) {   // 4 independent 64-bit accumulators: the pipe's MAD issue rate
            asm volatile(
                "v_mad_u64_u32 %[a0], %[c0], %[x], %[y], %[a0]\n\t"
                "v_mad_u64_u32 %[a1], %[c1], %[y], %[x], %[a1]\n\t"
                "v_mad_u64_u32 %[a2], %[c2], %[x], %[x], %[a2]\n\t"
                "v_mad_u64_u32 %[a3], %[c3], %[y], %[y], %[a3]\n\t"
                : [a0] "+&v"(alo), [a1] "+&v"(b1), [a2] "+&v"(b2), [a3] "+&v"(b3), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3)
                : [x] "v"(x), [y] "v"(y));
        } else if constexpr (MODE == 6) {   // two independent 96-bit accumulators, interleaved MAD / ADDC
            asm volatile(
                "v_mad_u64_u32 %[a0], %[c0], %[x], %[y], %[a0]\n\t"
                "v_mad_u64_u32 %[a1], %[c1], %[y], %[x], %[a1]\n\t"
                "v_mad_u64_u32 %[a0], %[c2], %[x], %[x], %[a0]\n\t"
                "v_mad_u64_u32 %[a1], %[c3], %[y], %[y], %[a1]\n\t"
                "v_addc_co_u32 %[h0], %[c0], 0, %[h0], %[c0]\n\t"
                "v_addc_co_u32 %[h1], %[c1], 0, %[h1], %[c1]\n\t"
                "v_addc_co_u32 %[h0], %[c2], 0, %[h0], %[c2]\n\t"
                "v_addc_co_u32 %[h1], %[c3], 0, %[h1], %[c3]\n\t"
                : [a0] "+&v"(alo), [a1] "+&v"(b1), [h0] "+&v"(ahi), [h1] "+&v"(h1), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3)
                : [x] "v"(x), [y] "v"(y));
        } else if constexpr (MODE == 7) {   // 4 independent v_add_u32
            asm volatile(
                "v_add_u32 %[h0], %[h0], %[x]\n\t"
                "v_add_u32 %[h1], %[h1], %[y]\n\t"
                "v_add_u32 %[h2], %[h2], %[x]\n\t"
                "v_add_u32 %[h3], %[h3], %[y]\n\t"
                : [h0] "+&v"(ahi), [h1] "+&v"(h1), [h2] "+&v"(h2), [h3] "+&v"(h3) : [x] "v"(x), [y] "v"(y));
        } else {
            asm volatile(
                "v_add_u32 %[ahi], %[ahi], %[x]\n\t"
                "v_add_u32 %[ahi], %[ahi], %[y]\n\t"
                "v_add_u32 %[ahi], %[ahi], %[x]\n\t"
                "v_add_u32 %[ahi], %[ahi], %[y]\n\t"
                : [ahi] "+&v"(ahi) : [x] "v"(x), [y] "v"(y));
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (alo + b1 + b2 + b3 == 0x123456789ull && ahi + h1 + h2 + h3 == 77) out[0] = 0;  // keep the chains alive
}

// chain of Montgomery products r = r * b (Fq_BLS381: 12 limbs) and of the fused r = r*b + c*d
template <int FUSED>
__global__ __launch_bounds__(64) void k_fpmul(uint64_t* out, uint32_t seed) {
    using namespace blz;
    Fp<Fq_BLS381> r, b, c, d;
    for (int i = 0; i < 12; ++i) {
        r.v[i] = (threadIdx.x + 1) * 2654435761u + seed * i;
        b.v[i] = r.v[i] ^ 0x9e3779b9u;
        c.v[i] = r.v[i] + 12345u * i;
        d.v[i] = b.v[i] ^ 0x55aa55aau;
    }
    r.v[11] &= 0x0fffffffu; b.v[11] &= 0x0fffffffu; c.v[11] &= 0x0fffffffu; d.v[11] &= 0x0fffffffu;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < MUL_REPS; ++i) {
        if constexpr (FUSED) fp_mul2(r, r, b, c, d);
        else fp_mul(r, r, b);
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    uint32_t o = 0;
    for (int i = 0; i < 12; ++i) o |= r.v[i];
    if (o == 0x12345u) out[0] = 0;
}

template <int FUSED>
void run_mul(const char* name, uint64_t* d_out, int cus) {
    for (int wps = 1; wps <= 4; ++wps) {
        int blocks = cus * 4 * wps;
        hipLaunchKernelGGL(k_fpmul<FUSED>, dim3(blocks), dim3(64), 0, 0, d_out, 1u);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_fpmul<FUSED>, dim3(blocks), dim3(64), 0, 0, d_out, 2u);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> h(blocks);
        hipMemcpy(h.data(), d_out, blocks * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        double med = (double)h[blocks / 2];
        double prod_per_s = (double)blocks * 64 * MUL_REPS / (ms * 1e-3);
        printf("%-28s waves/SIMD %d: %8.0f ticks per product per wave, %7.0f per SIMD; kernel %.3f ms -> %.3e lane-products/s, %.2f GHz tick rate\n",
               name, wps, med / MUL_REPS, med / MUL_REPS / wps, ms, prod_per_s, med / (ms * 1e6));
    }
}

template <int MODE>
void run(const char* name, int instr_per_batch, uint64_t* d_out, int cus) {
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = cus * 4 * wps;
        hipLaunchKernelGGL(k_issue<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, 1u);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_issue<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, 2u);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> h(blocks);
        hipMemcpy(h.data(), d_out, blocks * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        double med = (double)h[blocks / 2];
        double per_instr_wave = med / REPS / instr_per_batch;          // elapsed cycles per instruction seen by one wave
        printf("%-28s waves/SIMD %d: %6.2f ticks/instr per wave, %6.2f per SIMD (kernel %.3f ms, %.2f GHz tick rate)\n",
               name, wps, per_instr_wave, per_instr_wave / wps, ms, med / (ms * 1e6));
    }
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    printf("CUs %d, reported clock %d kHz; s_memtime ticks per instruction below\n", cus, khz);
    uint64_t* d_out;
    hipMalloc(&d_out, 8 * 8192);
    run<0>("4 MAD64 + 4 ADDC (batch)", 8, d_out, cus);
    run<1>("4 MAD64", 4, d_out, cus);
    run<2>("4 ADDC (vcc chain)", 4, d_out, cus);
    run<3>("4 v_add_u32", 4, d_out, cus);
    run<7>("4 v_add_u32, independent", 4, d_out, cus);
    run<5>("4 MAD64, 4 accumulators", 4, d_out, cus);
    run<6>("4 MAD+4 ADDC, 2 accumulators", 8, d_out, cus);
    run_mul<0>("fp_mul Fq_BLS381", d_out, cus);
    run_mul<1>("fp_mul2 (ab+cd) Fq_BLS381", d_out, cus);
    return 0;
}
