// Prime-field arithmetic for gfx950: 32-bit limbs, Montgomery form, R = 2^(32 N).
//
// Representation.  For fields with R > 4m (every Fq here, and Fr of BLS12-377 / BN254) values
// live in the redundant range [0, 2m] ("lazy"): Montgomery multiplication then needs no final
// subtraction (a,b <= 2m  =>  (ab + qm)/R < 2m), add/sub renormalise against 2m.  For Fr of
// BLS12-381 (255 bits in 256) values are kept canonical in [0, m).  `fp_reduce` gives the
// canonical representative either way.
//
// The multiplier is v_mad_u64_u32 (32x32+64 -> 64).  On MI355X it issues at ~1/2 the rate of a
// plain VALU add (profiles/r01_microbench_int_rates.txt), so a 12-limb Montgomery product
// (288 MACs) is ~900 issue slots: the kernels built on this are integer-issue bound, not HBM bound.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <utility>
#include "curve_constants.h"

namespace blz {

#define BLZ_DEV __device__ __forceinline__

BLZ_DEV uint32_t add_cc(uint32_t a, uint32_t b, uint32_t& carry) {
    uint32_t co;
    uint32_t r = __builtin_addc(a, b, carry, &co);
    carry = co;
    return r;
}
BLZ_DEV uint32_t sub_bb(uint32_t a, uint32_t b, uint32_t& borrow) {
    uint32_t bo;
    uint32_t r = __builtin_subc(a, b, borrow, &bo);
    borrow = bo;
    return r;
}

template <class P>
struct Fp {
    uint32_t v[P::N];
};

template <class P>
BLZ_DEV void fp_zero(Fp<P>& r) {
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = 0;
}
template <class P>
BLZ_DEV void fp_one(Fp<P>& r) {  // Montgomery one
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = P::R1[i];
}

// r = a - K if a >= K else a   (K a compile-time constant array)
template <class P, const uint32_t (&K)[P::N]>
BLZ_DEV void fp_csub_const(Fp<P>& a) {
    uint32_t t[P::N];
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) t[i] = sub_bb(a.v[i], K[i], br);
#pragma unroll
    for (int i = 0; i < P::N; ++i) a.v[i] = br ? a.v[i] : t[i];
}

// canonical representative in [0, m)
template <class P>
BLZ_DEV void fp_reduce(Fp<P>& a) {
    if constexpr (P::LAZY) {
        fp_csub_const<P, P::MOD>(a);  // [0,2m] -> [0,m]
        fp_csub_const<P, P::MOD>(a);  // m -> 0
    }
}

template <class P>
BLZ_DEV void fp_add(Fp<P>& r, const Fp<P>& a, const Fp<P>& b) {
    uint32_t c = 0;
    uint32_t t[P::N];
#pragma unroll
    for (int i = 0; i < P::N; ++i) t[i] = add_cc(a.v[i], b.v[i], c);
    // lazy: a+b <= 4m < R, no carry out.  strict: a+b < 2m < R as well (m < 2^(32N-1)).
    uint32_t u[P::N];
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) u[i] = sub_bb(t[i], P::LAZY ? P::MOD2[i] : P::MOD[i], br);
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = br ? t[i] : u[i];
}

template <class P>
BLZ_DEV void fp_sub(Fp<P>& r, const Fp<P>& a, const Fp<P>& b) {
    uint32_t br = 0;
    uint32_t t[P::N];
#pragma unroll
    for (int i = 0; i < P::N; ++i) t[i] = sub_bb(a.v[i], b.v[i], br);
    uint32_t c = 0;
    uint32_t mask = 0u - br;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = add_cc(t[i], (P::LAZY ? P::MOD2[i] : P::MOD[i]) & mask, c);
}

// r = 2m - a (lazy) / m - a (strict, a != 0 -> handled)   : additive inverse
template <class P>
BLZ_DEV void fp_neg(Fp<P>& r, const Fp<P>& a) {
    if constexpr (P::LAZY) {
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < P::N; ++i) r.v[i] = sub_bb(P::MOD2[i], a.v[i], br);
    } else {
        uint32_t nz = 0;
#pragma unroll
        for (int i = 0; i < P::N; ++i) nz |= a.v[i];
        uint32_t mask = nz ? 0xffffffffu : 0u;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < P::N; ++i) r.v[i] = sub_bb(P::MOD[i] & mask, a.v[i], br);
    }
}

template <class P>
BLZ_DEV void fp_dbl(Fp<P>& r, const Fp<P>& a) { fp_add(r, a, a); }

// ------------------------------------------------------------------------------------------
// Montgomery multiplication.
//
// fp_mul_cios: plain C++ coarsely-integrated operand scanning (what hipcc schedules by itself: per
//   MAC one v_mad_u64_u32 plus ~2 v_mov and a 64-bit add - kept as the readable reference variant).
// fp_mul_ps:   finely-integrated PRODUCT scanning with a 96-bit column accumulator (lo64, hi32):
//   every MAC is exactly  v_mad_u64_u32 lo64 += x*y (carry -> SGPR pair) ; v_addc_co_u32 hi32 += carry
//   i.e. 2 issue slots per 32x32 MAC and no register shuffling; modulus limbs ride the constant bus
//   as SGPRs.  a*b and q*m products go to two independent accumulators so one wave has two
//   dependency chains in flight.
// ------------------------------------------------------------------------------------------
template <class P>
BLZ_DEV void fp_mul_cios(Fp<P>& r, const Fp<P>& a, const Fp<P>& b) {
    constexpr int N = P::N;
    uint32_t t[N + 1];
#pragma unroll
    for (int j = 0; j <= N; ++j) t[j] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t c = 0;
        const uint32_t bi = b.v[i];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            c += (uint64_t)a.v[j] * bi + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[N];
        t[N] = (uint32_t)c;
        uint32_t hi = (uint32_t)(c >> 32);
        const uint32_t m = t[0] * P::N0;
        c = (uint64_t)m * P::MOD[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < N; ++j) {
            c += (uint64_t)m * P::MOD[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[N];
        t[N - 1] = (uint32_t)c;
        t[N] = hi + (uint32_t)(c >> 32);
    }
    if constexpr (P::LAZY) {
#pragma unroll
        for (int j = 0; j < N; ++j) r.v[j] = t[j];
    } else {
        // t < 2m, possibly with t[N] = 1
        uint32_t u[N];
        uint32_t br = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) u[j] = sub_bb(t[j], P::MOD[j], br);
        uint32_t keep = (t[N] == 0) & br;  // t < m
#pragma unroll
        for (int j = 0; j < N; ++j) r.v[j] = keep ? t[j] : u[j];
    }
}

// gfx950 hazard: a VALU instruction that reads an SGPR (here: the carry) written by a previous VALU
// instruction needs 2 wait states in between (LLVM GCNHazardRecognizer, VALUWriteSGPRVALURead; hipcc
// pads its own v_add_co/v_addc pairs with s_nop 1).  Inside an asm statement nothing is padded, so
// every carry below has >= 2 instructions (or an s_nop) between its producer and its consumer.
//
// (hi:lo) += x * y      x, y in VGPRs
BLZ_DEV void mac_vv(uint64_t& lo, uint32_t& hi, uint32_t x, uint32_t y) {
    uint64_t cr;
    asm("v_mad_u64_u32 %[lo], %[cr], %[x], %[y], %[lo]\n\ts_nop 1\n\tv_addc_co_u32 %[hi], %[cr], 0, %[hi], %[cr]"
        : [lo] "+&v"(lo), [hi] "+&v"(hi), [cr] "=&s"(cr)
        : [x] "v"(x), [y] "v"(y));
}
// (hi:lo) += x * k      k wave-uniform (modulus limb) in an SGPR
BLZ_DEV void mac_vs(uint64_t& lo, uint32_t& hi, uint32_t x, uint32_t k) {
    uint64_t cr;
    asm("v_mad_u64_u32 %[lo], %[cr], %[x], %[k], %[lo]\n\ts_nop 1\n\tv_addc_co_u32 %[hi], %[cr], 0, %[hi], %[cr]"
        : [lo] "+&v"(lo), [hi] "+&v"(hi), [cr] "=&s"(cr)
        : [x] "v"(x), [k] "s"(k));
}
// One statement = n MACs into the a*b accumulator interleaved with n MACs into the q*m accumulator
// (two independent dependency chains, four carry SGPR pairs in rotation).  hipcc pads every asm
// statement boundary with an s_nop, so MACs are batched per statement.
#define BLZ_MAC_AB(n, c) "v_mad_u64_u32 %[alo], %[" #c "], %[x" #n "], %[y" #n "], %[alo]\n\t"
#define BLZ_MAC_QM(n, c) "v_mad_u64_u32 %[mlo], %[" #c "], %[q" #n "], %[k" #n "], %[mlo]\n\t"
#define BLZ_CARRY_A(c) "v_addc_co_u32 %[ahi], %[" #c "], 0, %[ahi], %[" #c "]\n\t"
#define BLZ_CARRY_M(c) "v_addc_co_u32 %[mhi], %[" #c "], 0, %[mhi], %[" #c "]\n\t"
#define BLZ_PAIR2(n0, n1) \
    BLZ_MAC_AB(n0, c0) BLZ_MAC_QM(n0, c1) BLZ_MAC_AB(n1, c2) BLZ_MAC_QM(n1, c3) BLZ_CARRY_A(c0) BLZ_CARRY_M(c1) BLZ_CARRY_A(c2) BLZ_CARRY_M(c3)
BLZ_DEV void mac_pair1(uint64_t& alo, uint32_t& ahi, uint64_t& mlo, uint32_t& mhi, uint32_t x0, uint32_t y0, uint32_t q0,
                       uint32_t k0) {
    uint64_t c0, c1;
    asm(BLZ_MAC_AB(0, c0) BLZ_MAC_QM(0, c1) "s_nop 0\n\t" BLZ_CARRY_A(c0) BLZ_CARRY_M(c1)
        : [alo] "+&v"(alo), [ahi] "+&v"(ahi), [mlo] "+&v"(mlo), [mhi] "+&v"(mhi), [c0] "=&s"(c0), [c1] "=&s"(c1)
        : [x0] "v"(x0), [y0] "v"(y0), [q0] "v"(q0), [k0] "s"(k0));
}
BLZ_DEV void mac_pair2(uint64_t& alo, uint32_t& ahi, uint64_t& mlo, uint32_t& mhi, uint32_t x0, uint32_t y0, uint32_t q0,
                       uint32_t k0, uint32_t x1, uint32_t y1, uint32_t q1, uint32_t k1) {
    uint64_t c0, c1, c2, c3;
    asm(BLZ_PAIR2(0, 1)
        : [alo] "+&v"(alo), [ahi] "+&v"(ahi), [mlo] "+&v"(mlo), [mhi] "+&v"(mhi), [c0] "=&s"(c0), [c1] "=&s"(c1),
          [c2] "=&s"(c2), [c3] "=&s"(c3)
        : [x0] "v"(x0), [y0] "v"(y0), [q0] "v"(q0), [k0] "s"(k0), [x1] "v"(x1), [y1] "v"(y1), [q1] "v"(q1), [k1] "s"(k1));
}
BLZ_DEV void mac_pair4(uint64_t& alo, uint32_t& ahi, uint64_t& mlo, uint32_t& mhi, uint32_t x0, uint32_t y0, uint32_t q0,
                       uint32_t k0, uint32_t x1, uint32_t y1, uint32_t q1, uint32_t k1, uint32_t x2, uint32_t y2,
                       uint32_t q2, uint32_t k2, uint32_t x3, uint32_t y3, uint32_t q3, uint32_t k3) {
    uint64_t c0, c1, c2, c3;
    asm(BLZ_PAIR2(0, 1) BLZ_PAIR2(2, 3)
        : [alo] "+&v"(alo), [ahi] "+&v"(ahi), [mlo] "+&v"(mlo), [mhi] "+&v"(mhi), [c0] "=&s"(c0), [c1] "=&s"(c1),
          [c2] "=&s"(c2), [c3] "=&s"(c3)
        : [x0] "v"(x0), [y0] "v"(y0), [q0] "v"(q0), [k0] "s"(k0), [x1] "v"(x1), [y1] "v"(y1), [q1] "v"(q1), [k1] "s"(k1),
          [x2] "v"(x2), [y2] "v"(y2), [q2] "v"(q2), [k2] "s"(k2), [x3] "v"(x3), [y3] "v"(y3), [q3] "v"(q3), [k3] "s"(k3));
}

#ifndef BLZ_PS_SINGLE
#define BLZ_PS_SINGLE 1
#endif
// Single-accumulator forms: the same MACs, all into (ahi:alo).  No per-column merge of two
// accumulators (3 adds + hazard pads + 3 zeroing moves per column); the dependent v_mad chain is
// covered by the other waves on the SIMD.
#define BLZ_MAC_QS(n, c) "v_mad_u64_u32 %[alo], %[" #c "], %[q" #n "], %[k" #n "], %[alo]\n\t"
#define BLZ_QUAD1(n0, n1) \
    BLZ_MAC_AB(n0, c0) BLZ_MAC_QS(n0, c1) BLZ_MAC_AB(n1, c2) BLZ_MAC_QS(n1, c3) BLZ_CARRY_A(c0) BLZ_CARRY_A(c1) BLZ_CARRY_A(c2) BLZ_CARRY_A(c3)
BLZ_DEV void mac1_pair1(uint64_t& alo, uint32_t& ahi, uint32_t x0, uint32_t y0, uint32_t q0, uint32_t k0) {
    uint64_t c0, c1;
    asm(BLZ_MAC_AB(0, c0) BLZ_MAC_QS(0, c1) "s_nop 0\n\t" BLZ_CARRY_A(c0) BLZ_CARRY_A(c1)
        : [alo] "+&v"(alo), [ahi] "+&v"(ahi), [c0] "=&s"(c0), [c1] "=&s"(c1)
        : [x0] "v"(x0), [y0] "v"(y0), [q0] "v"(q0), [k0] "s"(k0));
}
BLZ_DEV void mac1_pair2(uint64_t& alo, uint32_t& ahi, uint32_t x0, uint32_t y0, uint32_t q0, uint32_t k0, uint32_t x1,
                        uint32_t y1, uint32_t q1, uint32_t k1) {
    uint64_t c0, c1, c2, c3;
    asm(BLZ_QUAD1(0, 1)
        : [alo] "+&v"(alo), [ahi] "+&v"(ahi), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3)
        : [x0] "v"(x0), [y0] "v"(y0), [q0] "v"(q0), [k0] "s"(k0), [x1] "v"(x1), [y1] "v"(y1), [q1] "v"(q1), [k1] "s"(k1));
}
BLZ_DEV void mac1_pair4(uint64_t& alo, uint32_t& ahi, uint32_t x0, uint32_t y0, uint32_t q0, uint32_t k0, uint32_t x1,
                        uint32_t y1, uint32_t q1, uint32_t k1, uint32_t x2, uint32_t y2, uint32_t q2, uint32_t k2,
                        uint32_t x3, uint32_t y3, uint32_t q3, uint32_t k3) {
    uint64_t c0, c1, c2, c3;
    asm(BLZ_QUAD1(0, 1) BLZ_QUAD1(2, 3)
        : [alo] "+&v"(alo), [ahi] "+&v"(ahi), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3)
        : [x0] "v"(x0), [y0] "v"(y0), [q0] "v"(q0), [k0] "s"(k0), [x1] "v"(x1), [y1] "v"(y1), [q1] "v"(q1), [k1] "s"(k1),
          [x2] "v"(x2), [y2] "v"(y2), [q2] "v"(q2), [k2] "s"(k2), [x3] "v"(x3), [y3] "v"(y3), [q3] "v"(q3), [k3] "s"(k3));
}
template <class P, int K>
BLZ_DEV void ps1_column(const Fp<P>& a, const Fp<P>& b, uint32_t (&q)[P::N], uint32_t (&t)[P::N], uint64_t& alo,
                        uint32_t& ahi) {
    constexpr int N = P::N;
    constexpr int ilo = K < N ? 0 : K - N + 1;
    constexpr int ihq = K < N ? K - 1 : N - 1;
    constexpr int npair = ihq - ilo + 1;
    constexpr int n4 = npair > 0 ? npair / 4 : 0;
    constexpr int rem = npair > 0 ? npair % 4 : 0;
#pragma unroll
    for (int g = 0; g < n4; ++g) {
        const int i = ilo + 4 * g;
        mac1_pair4(alo, ahi, a.v[i], b.v[K - i], q[i], P::MOD[K - i], a.v[i + 1], b.v[K - i - 1], q[i + 1], P::MOD[K - i - 1],
                   a.v[i + 2], b.v[K - i - 2], q[i + 2], P::MOD[K - i - 2], a.v[i + 3], b.v[K - i - 3], q[i + 3],
                   P::MOD[K - i - 3]);
    }
    {
        constexpr int i = ilo + 4 * n4;
        if constexpr (rem >= 2)
            mac1_pair2(alo, ahi, a.v[i], b.v[K - i], q[i], P::MOD[K - i], a.v[i + 1], b.v[K - i - 1], q[i + 1], P::MOD[K - i - 1]);
        if constexpr (rem == 1 || rem == 3) {
            constexpr int i2 = i + (rem == 3 ? 2 : 0);
            mac1_pair1(alo, ahi, a.v[i2], b.v[K - i2], q[i2], P::MOD[K - i2]);
        }
    }
    if constexpr (K < N) {
        mac_vv(alo, ahi, a.v[K], b.v[0]);  // the a*b product that has no q*m partner yet
        q[K] = (uint32_t)alo * P::N0;
        mac_vs(alo, ahi, q[K], P::MOD[0]);  // low word becomes zero
        alo = (alo >> 32) | ((uint64_t)ahi << 32);
    } else {
        t[K - N] = (uint32_t)alo;
        alo = (alo >> 32) | ((uint64_t)ahi << 32);
    }
    ahi = 0;
}

template <class P, int... Ks>
BLZ_DEV void ps1_columns(const Fp<P>& a, const Fp<P>& b, uint32_t (&q)[P::N], uint32_t (&t)[P::N], uint64_t& alo,
                         uint32_t& ahi, std::integer_sequence<int, Ks...>) {
    (ps1_column<P, Ks>(a, b, q, t, alo, ahi), ...);
}

// column K of the product scan: all a_i b_j and q_i m_j with i + j = K
template <class P, int K>
BLZ_DEV void ps_column(const Fp<P>& a, const Fp<P>& b, uint32_t (&q)[P::N], uint32_t (&t)[P::N], uint64_t& alo,
                       uint32_t& ahi) {
    constexpr int N = P::N;
    constexpr int ilo = K < N ? 0 : K - N + 1;
    constexpr int ihi = K < N ? K : N - 1;           // a*b products: i in [ilo, ihi]
    constexpr int ihq = K < N ? K - 1 : N - 1;       // q*m products: i in [ilo, ihq] (q_K m_0 comes after q_K exists)
    uint64_t mlo = 0;
    uint32_t mhi = 0;
    constexpr int npair = ihq - ilo + 1;
    constexpr int n4 = npair > 0 ? npair / 4 : 0;
    constexpr int rem = npair > 0 ? npair % 4 : 0;
#pragma unroll
    for (int g = 0; g < n4; ++g) {
        const int i = ilo + 4 * g;
        mac_pair4(alo, ahi, mlo, mhi, a.v[i], b.v[K - i], q[i], P::MOD[K - i], a.v[i + 1], b.v[K - i - 1], q[i + 1],
                  P::MOD[K - i - 1], a.v[i + 2], b.v[K - i - 2], q[i + 2], P::MOD[K - i - 2], a.v[i + 3], b.v[K - i - 3],
                  q[i + 3], P::MOD[K - i - 3]);
    }
    {
        constexpr int i = ilo + 4 * n4;
        if constexpr (rem >= 2)
            mac_pair2(alo, ahi, mlo, mhi, a.v[i], b.v[K - i], q[i], P::MOD[K - i], a.v[i + 1], b.v[K - i - 1], q[i + 1],
                      P::MOD[K - i - 1]);
        if constexpr (rem == 1 || rem == 3) {
            constexpr int i2 = i + (rem == 3 ? 2 : 0);
            mac_pair1(alo, ahi, mlo, mhi, a.v[i2], b.v[K - i2], q[i2], P::MOD[K - i2]);
        }
    }
    if constexpr (K < N) mac_vv(alo, ahi, a.v[K], b.v[0]);  // the a*b product that has no q*m partner yet
    // ---- merge the two accumulators: (ahi:alo) += (mhi:mlo)
    uint32_t l0 = (uint32_t)alo, l1 = (uint32_t)(alo >> 32);
    if constexpr (npair > 0) {
        uint32_t c = 0;
        l0 = add_cc(l0, (uint32_t)mlo, c);
        l1 = add_cc(l1, (uint32_t)(mlo >> 32), c);
        ahi = ahi + mhi + c;
    }
    if constexpr (K < N) {
        q[K] = l0 * P::N0;
        uint64_t lo2 = ((uint64_t)l1 << 32) | l0;
        mac_vs(lo2, ahi, q[K], P::MOD[0]);  // low word becomes zero
        alo = (lo2 >> 32) | ((uint64_t)ahi << 32);
    } else {
        t[K - N] = l0;
        alo = (uint64_t)l1 | ((uint64_t)ahi << 32);
    }
    ahi = 0;
}

template <class P, int... Ks>
BLZ_DEV void ps_columns(const Fp<P>& a, const Fp<P>& b, uint32_t (&q)[P::N], uint32_t (&t)[P::N], uint64_t& alo,
                        uint32_t& ahi, std::integer_sequence<int, Ks...>) {
    (ps_column<P, Ks>(a, b, q, t, alo, ahi), ...);
}

template <class P>
BLZ_DEV void fp_mul_ps(Fp<P>& r, const Fp<P>& a, const Fp<P>& b) {
    constexpr int N = P::N;
    uint32_t q[N];
    uint32_t t[N];
    uint64_t alo = 0;
    uint32_t ahi = 0;
#if BLZ_PS_SINGLE
    ps1_columns<P>(a, b, q, t, alo, ahi, std::make_integer_sequence<int, 2 * N>{});
#else
    ps_columns<P>(a, b, q, t, alo, ahi, std::make_integer_sequence<int, 2 * N>{});
#endif
    // alo now holds the word above the result (0 in the lazy representation)
    if constexpr (P::LAZY) {
#pragma unroll
        for (int j = 0; j < N; ++j) r.v[j] = t[j];
    } else {
        uint32_t u[N];
        uint32_t br = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) u[j] = sub_bb(t[j], P::MOD[j], br);
        uint32_t keep = ((uint32_t)alo == 0) & br;  // t < m
#pragma unroll
        for (int j = 0; j < N; ++j) r.v[j] = keep ? t[j] : u[j];
    }
}

#ifndef BLZ_MUL_VARIANT
#define BLZ_MUL_VARIANT 1
#endif
template <class P>
BLZ_DEV void fp_mul(Fp<P>& r, const Fp<P>& a, const Fp<P>& b) {
#if BLZ_MUL_VARIANT == 0
    fp_mul_cios(r, a, b);
#else
    fp_mul_ps(r, a, b);
#endif
}

template <class P>
BLZ_DEV void fp_sqr(Fp<P>& r, const Fp<P>& a) { fp_mul(r, a, a); }

template <class P>
BLZ_DEV void fp_to_mont(Fp<P>& r, const Fp<P>& a) {
    Fp<P> r2;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r2.v[i] = P::R2[i];
    fp_mul(r, a, r2);
}
// Montgomery -> canonical plain integer
template <class P>
BLZ_DEV void fp_from_mont(Fp<P>& r, const Fp<P>& a) {
    Fp<P> one;
#pragma unroll
    for (int i = 0; i < P::N; ++i) one.v[i] = (i == 0) ? 1u : 0u;
    fp_mul(r, a, one);
    fp_reduce(r);
}

template <class P>
BLZ_DEV bool fp_is_zero(const Fp<P>& a) {  // a == 0 (mod m)
    Fp<P> t = a;
    fp_reduce(t);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) o |= t.v[i];
    return o == 0;
}
template <class P>
BLZ_DEV bool fp_eq(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> d;
    fp_sub(d, a, b);
    return fp_is_zero(d);
}

// a^(m-2) by square-and-multiply (not unrolled: used once per MSM / batch)
template <class P>
__device__ __noinline__ void fp_inv(Fp<P>& r, const Fp<P>& a) {
    Fp<P> acc;
    fp_one(acc);
    Fp<P> base = a;
    bool started = false;
    for (int i = P::N * 32 - 1; i >= 0; --i) {
        uint32_t bit = (P::MODM2[i >> 5] >> (i & 31)) & 1u;
        if (started) fp_sqr(acc, acc);
        if (bit) {
            if (started) fp_mul(acc, acc, base);
            else { acc = base; started = true; }
        }
    }
    r = acc;
}

// ------------------------------------------------------------------------------------------
// global-memory I/O: limbs are contiguous little-endian dwords; 16-byte vector accesses.
// ------------------------------------------------------------------------------------------
template <class P>
BLZ_DEV void fp_load(Fp<P>& r, const void* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < P::N / 4; ++i) {
        uint4 x = q[i];
        r.v[4 * i] = x.x; r.v[4 * i + 1] = x.y; r.v[4 * i + 2] = x.z; r.v[4 * i + 3] = x.w;
    }
}
template <class P>
BLZ_DEV void fp_store(void* p, const Fp<P>& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < P::N / 4; ++i) q[i] = make_uint4(a.v[4 * i], a.v[4 * i + 1], a.v[4 * i + 2], a.v[4 * i + 3]);
}

}  // namespace blz
