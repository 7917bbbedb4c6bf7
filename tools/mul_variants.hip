// Field-multiplier alternatives on gfx950, timed the same way:
//   A  32-bit limbs, product scanning, v_mad_u64_u32 + v_addc_co_u32 per multiply-add (field.hip.hpp, shipped r01)
//   B  reduced radix 14 x 28 bits, v_mad_u64_u32 only, one 64-bit column accumulator (field_rr.hip.hpp)
//   C  reduced radix 13 x 30 bits, two-phase (a*b columns normalised, then the reduction columns)
//   D  instruction-mix model of the DFMA hi/lo multiplier (8 x 52-bit limbs: 2 v_fma_f64 + 1 v_add_f64 +
//      2 v_lshl_add_u64 per limb product) - NOT a functional multiplier, only its issue cost
//   E  instruction-mix model of 32-bit limbs with 64-bit column sums kept by v_lshl_add_u64
//      (v_mad_u64_u32 into a fresh register, then two 64-bit adds of the halves)
// plus the raw issue rates of the instructions those variants are made of.
// Every kernel is a dependent chain r = r * b on registers; throughput = lane-products / wall time of a
// >= 5 ms kernel; the shader clock is derived from s_memtime / s_memrealtime (100 MHz) in the same run.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I blaze_amd/csrc tools/mul_variants.hip -o build/mul_variants
#include <hip/hip_runtime.h>
#include "field_rr.hip.hpp"
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

using namespace blz;

// ---------------------------------------------------------------- variant C: 13 x 30 bits, two-phase
struct Q30 {
    static constexpr int B = 30, NL = 13;
    static constexpr uint32_t MASK = (1u << 30) - 1;
    uint32_t mod[13];
    uint32_t n0;
};
constexpr Q30 make_q30() {
    Q30 q{};
    for (int i = 0; i < 13; ++i) {
        int bit = 30 * i, j = bit >> 5, s = bit & 31;
        uint64_t lo = j < 12 ? Fq_BLS381::MOD[j] : 0, hi = j + 1 < 12 ? Fq_BLS381::MOD[j + 1] : 0;
        q.mod[i] = (uint32_t)(((lo | (hi << 32)) >> s) & Q30::MASK);
    }
    uint32_t inv = 1;
    for (int it = 0; it < 6; ++it) inv *= 2u - q.mod[0] * inv;
    q.n0 = (0u - inv) & Q30::MASK;
    return q;
}
__device__ constexpr Q30 kQ30 = make_q30();

__device__ __forceinline__ uint64_t mad64(uint32_t x, uint32_t y, uint64_t acc) { return (uint64_t)x * y + acc; }
template <int B> __device__ __forceinline__ uint64_t shr64(uint64_t acc) { return acc >> B; }
struct F30 { uint32_t v[13]; };
__device__ __forceinline__ void mul30(F30& r, const F30& a, const F30& b) {
    constexpr int NL = 13, B = 30;
    uint32_t t[2 * NL];
    uint64_t acc = 0;
    // phase 1: a*b, columns normalised (<= 13 products < 13 * 2^60 < 2^64)
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; ++k) {
        const int ilo = k < NL ? 0 : k - NL + 1, ihi = k < NL ? k : NL - 1;
#pragma unroll
        for (int i = ilo; i <= ihi; ++i) acc = mad64(a.v[i], b.v[k - i], acc);
        t[k] = (uint32_t)acc & Q30::MASK;
        acc = shr64<B>(acc);
    }
    t[2 * NL - 1] = (uint32_t)acc;
    // phase 2: reduction columns (t_k + <= 13 products)
    uint32_t q[NL];
    acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; ++k) {
        acc = mad64(t[k], 1u, acc);
        if (k < NL) {
#pragma unroll
            for (int i = 0; i < k; ++i) acc = mad64(q[i], kQ30.mod[k - i], acc);
            q[k] = ((uint32_t)acc * kQ30.n0) & Q30::MASK;
            acc = mad64(q[k], kQ30.mod[0], acc);
        } else {
#pragma unroll
            for (int i = k - NL + 1; i < NL; ++i) acc = mad64(q[i], kQ30.mod[k - i], acc);
            r.v[k - NL] = (uint32_t)acc & Q30::MASK;
        }
        acc = shr64<B>(acc);
    }
    r.v[NL - 1] = (uint32_t)acc + t[2 * NL - 1];
}

constexpr int MUL_REPS_DEFAULT = 8192;

template <int V>
__global__ __launch_bounds__(256) void k_mul(uint64_t* out, uint32_t seed, int MUL_REPS) {
    uint64_t t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    uint32_t sink = 0;
    if constexpr (V == 0 || V == 10) {
        Fp<Fq_BLS381> r, b, c, d;
        for (int i = 0; i < 12; ++i) {
            r.v[i] = (threadIdx.x + 1) * 2654435761u + seed * i;
            b.v[i] = r.v[i] ^ 0x9e3779b9u; c.v[i] = r.v[i] + 12345u * i; d.v[i] = b.v[i] ^ 0x55aa55aau;
        }
        r.v[11] &= 0x0fffffffu; b.v[11] &= 0x0fffffffu; c.v[11] &= 0x0fffffffu; d.v[11] &= 0x0fffffffu;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < MUL_REPS; ++i) {
            if constexpr (V == 10) fp_mul2(r, r, b, c, d); else fp_mul(r, r, b);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 12; ++i) sink |= r.v[i];
    } else if constexpr (V == 5 || V == 6 || V == 12) {
        using Q = Fq_BLS381_RR;
        Frr<Q, 1, 2> r, b, c, d;
        for (int i = 0; i < Q::NL; ++i) {
            r.v[i] = ((threadIdx.x + 1) * 2654435761u + seed * i) & Q::MASK;
            b.v[i] = (r.v[i] ^ 0x9e3779b9u) & Q::MASK; c.v[i] = (r.v[i] + 12345u * i) & Q::MASK; d.v[i] = (b.v[i] ^ 0x55aa55aau) & Q::MASK;
        }
        r.v[Q::NL - 1] &= 0xffffu; b.v[Q::NL - 1] &= 0xffffu; c.v[Q::NL - 1] &= 0xffffu; d.v[Q::NL - 1] &= 0xffffu;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < MUL_REPS; ++i) {
            if constexpr (V == 12) rr_mul2(r, r, b, c, d);
            else if constexpr (V == 6) rr_sqr(r, r);
            else rr_mul(r, r, b);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < Q::NL; ++i) sink |= r.v[i];
    } else if constexpr (V == 1) {
        using Q = Fq_BLS381_RR;
        Frr<Q, 1, 2> r, b, c, d;
        for (int i = 0; i < Q::NL; ++i) {
            r.v[i] = ((threadIdx.x + 1) * 2654435761u + seed * i) & Q::MASK;
            b.v[i] = (r.v[i] ^ 0x9e3779b9u) & Q::MASK; c.v[i] = (r.v[i] + 12345u * i) & Q::MASK; d.v[i] = (b.v[i] ^ 0x55aa55aau) & Q::MASK;
        }
        r.v[Q::NL - 1] &= 0xffffu; b.v[Q::NL - 1] &= 0xffffu; c.v[Q::NL - 1] &= 0xffffu; d.v[Q::NL - 1] &= 0xffffu;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < MUL_REPS; ++i) {
            rr_mul_ref(r, r, b);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < Q::NL; ++i) sink |= r.v[i];
    } else if constexpr (V == 2) {
        F30 r, b;
        for (int i = 0; i < 13; ++i) {
            r.v[i] = ((threadIdx.x + 1) * 2654435761u + seed * i) & Q30::MASK;
            b.v[i] = (r.v[i] ^ 0x9e3779b9u) & Q30::MASK;
        }
        r.v[12] &= 0xffffu; b.v[12] &= 0xffffu;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < MUL_REPS; ++i) mul30(r, r, b);
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 13; ++i) sink |= r.v[i];
    } else if constexpr (V == 3) {
        // D: DFMA model.  128 limb products per Montgomery product (8 x 8 for a*b, 8 x 8 for q*m), each
        // 2 FMA + 1 ADD (f64) + 2 x 64-bit integer add; 16 columns x ~6 ops of carry resolution / int<->double.
        double a = 1.0 + threadIdx.x, b = 3.0 + seed, hi = 0, lo = 0, sub = 0;
        const double c1 = 0x1p104, c2 = 0x1p104 + 0x1p52;
        uint64_t s0 = threadIdx.x, s1 = seed;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < MUL_REPS; ++i) {
#pragma unroll
            for (int p = 0; p < 128; ++p) {
                asm volatile(
                    "v_fma_f64 %[hi], %[a], %[b], %[c1]\n\t"
                    "v_add_f64 %[sub], %[c2], -%[hi]\n\t"
                    "v_fma_f64 %[lo], %[a], %[b], %[sub]\n\t"
                    "v_lshl_add_u64 %[s0], %[hi], 0, %[s0]\n\t"
                    "v_lshl_add_u64 %[s1], %[lo], 0, %[s1]\n\t"
                    : [hi] "=&v"(hi), [lo] "=&v"(lo), [sub] "=&v"(sub), [s0] "+v"(s0), [s1] "+v"(s1)
                    : [a] "v"(a), [b] "v"(b), [c1] "v"(c1), [c2] "v"(c2));
            }
#pragma unroll
            for (int p = 0; p < 16; ++p) {  // per column: subtract the bias, split at 52 bits, carry on, convert
                asm volatile(
                    "v_lshl_add_u64 %[s0], %[s1], 0, %[s0]\n\t"
                    "v_lshrrev_b64 %[s1], 52, %[s0]\n\t"
                    "v_and_b32 %[x], 0xfffff, %[x]\n\t"
                    "v_lshl_add_u64 %[s1], %[s0], 0, %[s1]\n\t"
                    "v_cvt_f64_u32 %[hi], %[x]\n\t"
                    "v_fma_f64 %[lo], %[hi], %[c1], %[lo]\n\t"
                    : [s0] "+v"(s0), [s1] "+v"(s1), [x] "+v"(sink), [hi] "=&v"(hi), [lo] "+v"(lo)
                    : [c1] "v"(c1));
            }
            a = lo;
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        sink |= (uint32_t)s0 ^ (uint32_t)s1 ^ (uint32_t)__double_as_longlong(a);
    } else if constexpr (V == 4) {
        // E: 32-bit limbs, 64-bit column sums by v_lshl_add_u64: per multiply-add one v_mad_u64_u32 (zero addend) +
        // v_lshl_add_u64 of the zero-extended low half + v_lshl_add_u64 of the zero-extended high half
        // (the zero-extension moves are NOT counted: lower bound), 288 per product + 24 column closes of 4 ops
        uint32_t x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u;
        uint64_t p = 0, lo = x, hi = y, z = 0;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < MUL_REPS; ++i) {
#pragma unroll
            for (int k = 0; k < 288; ++k) {
                asm volatile(
                    "v_mad_u64_u32 %[p], vcc, %[x], %[y], 0\n\t"
                    "v_lshl_add_u64 %[lo], %[p], 0, %[lo]\n\t"
                    "v_lshl_add_u64 %[hi], %[z], 0, %[hi]\n\t"
                    : [p] "=&v"(p), [lo] "+v"(lo), [hi] "+v"(hi) : [x] "v"(x), [y] "v"(y), [z] "v"(z) : "vcc");
            }
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                asm volatile(
                    "v_mul_lo_u32 %[x], %[x], %[y]\n\t"
                    "v_lshrrev_b64 %[lo], 32, %[lo]\n\t"
                    "v_lshl_add_u64 %[lo], %[hi], 0, %[lo]\n\t"
                    "v_mov_b32 %[y], %[x]\n\t"
                    : [x] "+v"(x), [y] "+v"(y), [lo] "+v"(lo) : [hi] "v"(hi));
            }
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        sink |= (uint32_t)lo ^ (uint32_t)hi ^ x;
    }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (sink == 0x12345u) out[0] = 0;
}

// ---------------------------------------------------------------- raw issue rates
constexpr int IREPS = 1 << 16;
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(uint64_t* out, uint32_t seed) {
    uint32_t x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u, a0 = 1, a1 = 2, a2 = 3, a3 = 4;
    uint64_t w0 = x, w1 = y, w2 = x + 1, w3 = y + 1;
    double d0 = 1.0 + x, d1 = 2.0, d2 = 3.0, d3 = 4.0, dx = 1.0000001, dy = 0.5;
    uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < IREPS; ++r) {
#define I4(fmt)                                                                                                      \
    asm volatile(fmt(a0, w0, d0) fmt(a1, w1, d1) fmt(a2, w2, d2) fmt(a3, w3, d3)                                      \
                 : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2), \
                   [w3] "+v"(w3), [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3)                                \
                 : [x] "v"(x), [y] "v"(y), [dx] "v"(dx), [dy] "v"(dy)                                                      \
                 : "vcc")
#define F_ADD(a, w, d) "v_add_u32 %[" #a "], %[" #a "], %[x]\n\t"
#define F_MAD(a, w, d) "v_mad_u64_u32 %[" #w "], vcc, %[x], %[y], %[" #w "]\n\t"
#define F_MADI(a, w, d) "v_mad_i64_i32 %[" #w "], vcc, %[x], %[y], %[" #w "]\n\t"
#define F_LSHLADD(a, w, d) "v_lshl_add_u64 %[" #w "], %[" #w "], 0, %[" #w "]\n\t"
#define F_SHR64(a, w, d) "v_lshrrev_b64 %[" #w "], 3, %[" #w "]\n\t"
#define F_ALIGN(a, w, d) "v_alignbit_b32 %[" #a "], %[" #a "], %[x], 28\n\t"
#define F_AND(a, w, d) "v_and_b32 %[" #a "], %[" #a "], %[x]\n\t"
#define F_MULLO(a, w, d) "v_mul_lo_u32 %[" #a "], %[" #a "], %[x]\n\t"
#define F_BFE(a, w, d) "v_bfe_u32 %[" #a "], %[" #a "], 3, 28\n\t"
#define F_ADD3(a, w, d) "v_add3_u32 %[" #a "], %[" #a "], %[x], %[y]\n\t"
#define F_FMA64(a, w, d) "v_fma_f64 %[" #d "], %[" #d "], %[dx], %[dy]\n\t"
#define F_ADD64(a, w, d) "v_add_f64 %[" #d "], %[" #d "], %[dy]\n\t"
#define F_SUB(a, w, d) "v_sub_u32 %[" #a "], %[x], %[" #a "]\n\t"
#define F_LSHLADD32(a, w, d) "v_lshl_add_u32 %[" #a "], %[" #a "], 1, %[x]\n\t"
#define F_ADDC(a, w, d) "v_addc_co_u32 %[" #a "], vcc, 0, %[" #a "], vcc\n\t"
        if constexpr (MODE == 0) I4(F_ADD);
        else if constexpr (MODE == 1) I4(F_MAD);
        else if constexpr (MODE == 2) I4(F_MADI);
        else if constexpr (MODE == 3) I4(F_LSHLADD);
        else if constexpr (MODE == 4) I4(F_SHR64);
        else if constexpr (MODE == 5) I4(F_ALIGN);
        else if constexpr (MODE == 6) I4(F_AND);
        else if constexpr (MODE == 7) I4(F_MULLO);
        else if constexpr (MODE == 8) I4(F_BFE);
        else if constexpr (MODE == 9) I4(F_ADD3);
        else if constexpr (MODE == 10) I4(F_FMA64);
        else if constexpr (MODE == 11) I4(F_ADD64);
        else if constexpr (MODE == 12) I4(F_SUB);
        else if constexpr (MODE == 13) I4(F_LSHLADD32);
        else I4(F_ADDC);
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (a0 + a1 + a2 + a3 + (uint32_t)(w0 + w1 + w2 + w3) + (uint32_t)(d0 + d1 + d2 + d3) == 0x1234567u) out[0] = 0;
}

// ---------------------------------------------------------------- correctness of B against A
__global__ void k_check(uint32_t* bad, uint32_t seed) {
    using Q = Fq_BLS381_RR;
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 747796405u + seed;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s ^ (s >> 15); };
    uint32_t xw[12], yw[12];
    for (int i = 0; i < 12; ++i) { xw[i] = rnd(); yw[i] = rnd(); }
    xw[11] &= 0x0fffffffu; yw[11] &= 0x0fffffffu;  // < 2^380 < m
    Fp<Fq_BLS381> x, y, r32;
    for (int i = 0; i < 12; ++i) { x.v[i] = xw[i]; y.v[i] = yw[i]; }
    fp_to_mont(x, x); fp_to_mont(y, y);
    Fp<Fq_BLS381> s32, d32;
    fp_add(s32, x, y); fp_sub(d32, x, y);
    fp_mul2(r32, x, y, s32, d32);      // x y + (x + y)(x - y)
    fp_from_mont(r32, r32);
    Frr<Q, 1, 2> a, b, r, rref, s1, s2;
    rr_to_mont_from_words<Q>(a, xw);
    rr_to_mont_from_words<Q>(b, yw);
    const auto sm = rr_add(a, b);      // limbs < 2 * 2^B, value < 4m
    const auto df = rr_sub<2>(a, b);   // limbs < 3 * 2^B, value < 6m
    rr_mul2(r, a, b, sm, df);
    {
        // asm columns against the plain C++ scan, and the squaring against the product
        rr_mul(s1, sm, df); rr_mul_ref(s2, sm, df);
        Frr<Q, 1, 2> q1, q2;
        rr_sqr(q1, df); rr_mul_ref(q2, df, df);
        uint32_t dd = 0;
        for (int i = 0; i < Q::NL; ++i) dd |= (s1.v[i] ^ s2.v[i]) | (q1.v[i] ^ q2.v[i]);
        if (dd) atomicAdd(bad, 1u);
    }
    uint32_t w[12];
    rr_to_mont32_words<Q>(w, r);
    Fp<Fq_BLS381> back;
    for (int i = 0; i < 12; ++i) back.v[i] = w[i];
    fp_from_mont(back, back);
    uint32_t diff = 0;
    for (int i = 0; i < 12; ++i) diff |= back.v[i] ^ r32.v[i];
    // and the zero test
    const auto z = rr_sub<2>(a, a);
    if (!rr_is_zero(z) || !rr_maybe_equal(a, a) || rr_is_zero(a)) diff |= 1;
    if (diff) atomicAdd(bad, 1u);
}

struct Res { double prod_per_s, ticks, ghz; };
template <class K, class... A>
Res time_kernel(K kern, int blocks, uint64_t* d_out, double work_per_block, A... extra) {
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1u, extra...);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 2u, extra...);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(2 * blocks);
    hipMemcpy(h.data(), d_out, 16 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> tk(blocks), rt(blocks);
    for (int i = 0; i < blocks; ++i) { tk[i] = (double)h[2 * i]; rt[i] = (double)h[2 * i + 1]; }
    std::sort(tk.begin(), tk.end()); std::sort(rt.begin(), rt.end());
    Res r;
    r.prod_per_s = work_per_block * blocks / (ms * 1e-3);
    r.ticks = tk[blocks / 2];
    r.ghz = tk[blocks / 2] / (rt[blocks / 2] / 100e6) / 1e9;
    return r;
}

template <int V>
void run_mul(const char* name, uint64_t* d_out, int cus, double baseline[5]) {
    for (int wps = 1; wps <= 4; ++wps) {
        int occ = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_mul<V>, 256, 0);
        if (occ < wps) { printf("%-44s waves/SIMD %d: not resident (max %d)\n", name, wps, occ); continue; }
        Res r = time_kernel(k_mul<V>, cus * wps, d_out, 256.0 * MUL_REPS_DEFAULT, MUL_REPS_DEFAULT);
        if (baseline[wps] == 0) baseline[wps] = r.prod_per_s;
        printf("%-44s waves/SIMD %d: %.3e lane-products/s (x%.3f vs A), clock %.2f GHz\n",
               name, wps, r.prod_per_s, r.prod_per_s / baseline[wps], r.ghz);
    }
}
template <int MODE>
void run_rate(const char* name, uint64_t* d_out, int cus) {
    for (int wps = 1; wps <= 8; wps *= 2) {
        int occ = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_rate<MODE>, 256, 0);
        if (occ < wps) continue;
        Res r = time_kernel(k_rate<MODE>, cus * wps, d_out, 256.0 * IREPS * 4);
        printf("%-22s waves/SIMD %d: %.3e lane-ops/s, %5.2f cycles per wave-instruction per SIMD, clock %.2f GHz\n", name, wps,
               r.prod_per_s, r.ticks / IREPS / 4 / wps, r.ghz);
    }
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("CUs %d\n", cus);
    uint64_t* d_out;
    hipMalloc(&d_out, 16 * 8192);
    uint32_t* d_bad;
    hipMalloc(&d_bad, 4);
    hipMemset(d_bad, 0, 4);
    hipLaunchKernelGGL(k_check, dim3(256), dim3(256), 0, 0, d_bad, 12345u);
    uint32_t bad = 1;
    hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
    printf("check B (14x28 reduced radix, lazy add/sub, fused ab+cd, conversions) against A on 65536 random pairs: %u mismatches\n", bad);
    double base[5] = {0, 0, 0, 0, 0}, base2[5] = {0, 0, 0, 0, 0};
    run_mul<0>("A  12x32, mad+addc product scan (shipped r01)", d_out, cus, base);
    run_mul<1>("B  14x28, mad only, plain C++ (hipcc's schedule)", d_out, cus, base);
    run_mul<5>("B' 14x28, asm columns", d_out, cus, base);
    run_mul<6>("B' 14x28, asm columns, squaring", d_out, cus, base);
    run_mul<2>("C  13x30, mad only, two-phase", d_out, cus, base);
    run_mul<3>("D  DFMA 8x52 instruction-mix model", d_out, cus, base);
    run_mul<4>("E  12x32 + v_lshl_add_u64 column sums (model)", d_out, cus, base);
    run_mul<10>("A2 12x32 fused ab+cd", d_out, cus, base2);
    // (the plain C++ fused form needs 454 VGPRs after hipcc's re-association: 3.7e10 at 1 wave per SIMD, not resident at 2)
    run_mul<12>("B2' 14x28 fused ab+cd, asm columns", d_out, cus, base2);
    // sustained: the accumulation kernel runs for ~100 ms; does the multiplier keep its rate and clock that long?
    for (int reps : {8192, 65536, 262144}) {
        Res r = time_kernel(k_mul<5>, cus * 2, d_out, 256.0 * reps, reps);
        printf("B' sustained, 2 waves/SIMD, %6d products per lane (%.1f ms): %.3e lane-products/s, clock %.2f GHz\n", reps,
               256.0 * reps * cus * 2 / r.prod_per_s * 1e3, r.prod_per_s, r.ghz);
    }
    run_rate<0>("v_add_u32", d_out, cus);
    run_rate<1>("v_mad_u64_u32", d_out, cus);
    run_rate<2>("v_mad_i64_i32", d_out, cus);
    run_rate<3>("v_lshl_add_u64", d_out, cus);
    run_rate<4>("v_lshrrev_b64", d_out, cus);
    run_rate<5>("v_alignbit_b32", d_out, cus);
    run_rate<6>("v_and_b32", d_out, cus);
    run_rate<7>("v_mul_lo_u32", d_out, cus);
    run_rate<8>("v_bfe_u32", d_out, cus);
    run_rate<9>("v_add3_u32", d_out, cus);
    run_rate<10>("v_fma_f64", d_out, cus);
    run_rate<11>("v_add_f64", d_out, cus);
    run_rate<12>("v_sub_u32", d_out, cus);
    run_rate<13>("v_lshl_add_u32", d_out, cus);
    run_rate<14>("v_addc_co_u32 (vcc)", d_out, cus);
    return bad != 0;
}
