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
// fp_mul_ps:   finely-integrated PRODUCT scanning with a 96-bit column accumulator (lo64, hi32):
//   every MAC is exactly  v_mad_u64_u32 lo64 += x*y (carry -> SGPR pair) ; v_addc_co_u32 hi32 += carry
//   i.e. 2 issue slots per 32x32 MAC and no register shuffling; modulus limbs ride the constant bus
//   as SGPRs.  One accumulator for a*b and q*m products (a second one costs a 3-add merge per
//   column and measured slower); the dependent v_mad chain is covered by the other waves.
// ------------------------------------------------------------------------------------------
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
// MAC batches (one asm statement each; hipcc pads every statement boundary with an s_nop, so MACs
// are batched): generated by tools/gen_mac_asm.py.
#include "mac_gen.inc"

// One chunk of column K: NP (a_i b_(K-i), q_i m_(K-i)) pairs starting at i = I0, optionally opening
// the column (FIRST: the carry word is written, not accumulated) and optionally closing its a*b
// products with a_K b_0 (EXTRA, columns K < N).
template <class P, int K, int I0, int NP, bool FIRST, bool EXTRA>
BLZ_DEV void ps_chunk(const Fp<P>& a, const Fp<P>& b, const uint32_t (&q)[P::N], uint64_t& alo, uint32_t& ahi) {
#define BLZ_PA(d) a.v[I0 + d], b.v[K - I0 - d], q[I0 + d], P::MOD[K - I0 - d]
#define BLZ_DISPATCH(NPV, ...)                                                                        \
    if constexpr (NP == NPV) {                                                                        \
        if constexpr (FIRST && EXTRA) mac1_p##NPV##fx(alo, ahi, __VA_ARGS__, a.v[K], b.v[0]);         \
        else if constexpr (FIRST) mac1_p##NPV##f(alo, ahi, __VA_ARGS__);                              \
        else if constexpr (EXTRA) mac1_p##NPV##x(alo, ahi, __VA_ARGS__, a.v[K], b.v[0]);              \
        else mac1_p##NPV(alo, ahi, __VA_ARGS__);                                                      \
    }
    if constexpr (NP == 0) {
        static_assert(NP != 0 || EXTRA, "empty chunk");
        if constexpr (FIRST) mac1_p0fx(alo, ahi, a.v[K], b.v[0]);
        else mac1_p0x(alo, ahi, a.v[K], b.v[0]);
    }
    BLZ_DISPATCH(1, BLZ_PA(0))
    BLZ_DISPATCH(2, BLZ_PA(0), BLZ_PA(1))
    BLZ_DISPATCH(3, BLZ_PA(0), BLZ_PA(1), BLZ_PA(2))
    BLZ_DISPATCH(4, BLZ_PA(0), BLZ_PA(1), BLZ_PA(2), BLZ_PA(3))
#undef BLZ_DISPATCH
#undef BLZ_PA
}

template <class P, int K, int I0, int LEFT, bool FIRST, bool EXTRA>
BLZ_DEV void ps_chunks(const Fp<P>& a, const Fp<P>& b, const uint32_t (&q)[P::N], uint64_t& alo, uint32_t& ahi) {
    if constexpr (LEFT > 4) {
        ps_chunk<P, K, I0, 4, FIRST, false>(a, b, q, alo, ahi);
        ps_chunks<P, K, I0 + 4, LEFT - 4, false, EXTRA>(a, b, q, alo, ahi);
    } else if constexpr (LEFT > 0 || EXTRA) {
        ps_chunk<P, K, I0, LEFT, FIRST, EXTRA>(a, b, q, alo, ahi);
    }
}

// column K of the product scan: all a_i b_j and q_i m_j with i + j = K, single 96-bit accumulator
template <class P, int K>
BLZ_DEV void ps_column(const Fp<P>& a, const Fp<P>& b, uint32_t (&q)[P::N], uint32_t (&t)[P::N], uint64_t& alo) {
    constexpr int N = P::N;
    constexpr int ilo = K < N ? 0 : K - N + 1;
    constexpr int ihq = K < N ? K - 1 : N - 1;  // q*m products: i in [ilo, ihq] (q_K m_0 comes after q_K exists)
    constexpr int npair = ihq - ilo + 1 > 0 ? ihq - ilo + 1 : 0;
    constexpr bool extra = K < N;                // a_K b_0
    if constexpr (npair == 0 && !extra) {        // K = 2N-1: nothing left to add
        t[K - N] = (uint32_t)alo;
        alo >>= 32;
    } else {
        uint32_t ahi;  // written by the column's first batch
        ps_chunks<P, K, ilo, npair, true, extra>(a, b, q, alo, ahi);
        if constexpr (K < N) {
            q[K] = (uint32_t)alo * P::N0;
            mac_vs(alo, ahi, q[K], P::MOD[0]);  // low word becomes zero
        } else {
            t[K - N] = (uint32_t)alo;
        }
        alo = (alo >> 32) | ((uint64_t)ahi << 32);
    }
}

template <class P, int... Ks>
BLZ_DEV void ps_columns(const Fp<P>& a, const Fp<P>& b, uint32_t (&q)[P::N], uint32_t (&t)[P::N], uint64_t& alo,
                        std::integer_sequence<int, Ks...>) {
    (ps_column<P, Ks>(a, b, q, t, alo), ...);
}

template <class P>
BLZ_DEV void fp_mul_ps(Fp<P>& r, const Fp<P>& a, const Fp<P>& b) {
    constexpr int N = P::N;
    uint32_t q[N];
    uint32_t t[N];
    uint64_t alo = 0;
    ps_columns<P>(a, b, q, t, alo, std::make_integer_sequence<int, 2 * N>{});
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

// ------------------------------------------------------------------------------------------------
// Sum of two products with ONE Montgomery reduction:  r = (a b + c d) R^-1  (mod m).
// Column K accumulates a_i b_j + c_i d_j + q_i m_j; everything else is the scan above.  The group law
// ends in such a sum (Y3 = R (Q - X3) - Y1 PPP), so each point addition saves one reduction: N^2 of the
// 2 N^2 multiply-adds of a field multiplication.
// Range: inputs in [0, 2m] give (a b + c d + q m) / R < 8 m^2 / R + m, which is < 2m when 8 m <= R
// (both BLS fields); BN254's q is 254 of 256 bits (8 m^2 / R < 1.52 m), so one conditional subtraction
// of 2m brings the result back into the lazy range.
// ------------------------------------------------------------------------------------------------
template <class P, int K, int I0, int NP, bool FIRST, bool EXTRA>
BLZ_DEV void ps2_chunk(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d, const uint32_t (&q)[P::N],
                       uint64_t& alo, uint32_t& ahi) {
#define BLZ_PA(e) a.v[I0 + e], b.v[K - I0 - e], c.v[I0 + e], d.v[K - I0 - e], q[I0 + e], P::MOD[K - I0 - e]
#define BLZ_DISPATCH(NPV, ...)                                                                               \
    if constexpr (NP == NPV) {                                                                               \
        if constexpr (FIRST && EXTRA) mac2_p##NPV##fx(alo, ahi, __VA_ARGS__, a.v[K], b.v[0], c.v[K], d.v[0]); \
        else if constexpr (FIRST) mac2_p##NPV##f(alo, ahi, __VA_ARGS__);                                     \
        else if constexpr (EXTRA) mac2_p##NPV##x(alo, ahi, __VA_ARGS__, a.v[K], b.v[0], c.v[K], d.v[0]);      \
        else mac2_p##NPV(alo, ahi, __VA_ARGS__);                                                             \
    }
    if constexpr (NP == 0) {
        static_assert(NP != 0 || EXTRA, "empty chunk");
        if constexpr (FIRST) mac2_p0fx(alo, ahi, a.v[K], b.v[0], c.v[K], d.v[0]);
        else mac2_p0x(alo, ahi, a.v[K], b.v[0], c.v[K], d.v[0]);
    }
    BLZ_DISPATCH(1, BLZ_PA(0))
    BLZ_DISPATCH(2, BLZ_PA(0), BLZ_PA(1))
    BLZ_DISPATCH(3, BLZ_PA(0), BLZ_PA(1), BLZ_PA(2))
#undef BLZ_DISPATCH
#undef BLZ_PA
}

template <class P, int K, int I0, int LEFT, bool FIRST, bool EXTRA>
BLZ_DEV void ps2_chunks(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d, const uint32_t (&q)[P::N],
                        uint64_t& alo, uint32_t& ahi) {
    if constexpr (LEFT > 3) {
        ps2_chunk<P, K, I0, 3, FIRST, false>(a, b, c, d, q, alo, ahi);
        ps2_chunks<P, K, I0 + 3, LEFT - 3, false, EXTRA>(a, b, c, d, q, alo, ahi);
    } else if constexpr (LEFT > 0 || EXTRA) {
        ps2_chunk<P, K, I0, LEFT, FIRST, EXTRA>(a, b, c, d, q, alo, ahi);
    }
}

template <class P, int K>
BLZ_DEV void ps2_column(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d, uint32_t (&q)[P::N],
                        uint32_t (&t)[P::N], uint64_t& alo) {
    constexpr int N = P::N;
    constexpr int ilo = K < N ? 0 : K - N + 1;
    constexpr int ihq = K < N ? K - 1 : N - 1;
    constexpr int npair = ihq - ilo + 1 > 0 ? ihq - ilo + 1 : 0;
    constexpr bool extra = K < N;
    if constexpr (npair == 0 && !extra) {
        t[K - N] = (uint32_t)alo;
        alo >>= 32;
    } else {
        uint32_t ahi;
        ps2_chunks<P, K, ilo, npair, true, extra>(a, b, c, d, q, alo, ahi);
        if constexpr (K < N) {
            q[K] = (uint32_t)alo * P::N0;
            mac_vs(alo, ahi, q[K], P::MOD[0]);
        } else {
            t[K - N] = (uint32_t)alo;
        }
        alo = (alo >> 32) | ((uint64_t)ahi << 32);
    }
}

template <class P, int... Ks>
BLZ_DEV void ps2_columns(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d, uint32_t (&q)[P::N],
                         uint32_t (&t)[P::N], uint64_t& alo, std::integer_sequence<int, Ks...>) {
    (ps2_column<P, Ks>(a, b, c, d, q, t, alo), ...);
}

// r = a b + c d   (lazy fields only: the group law's base fields)
template <class P>
BLZ_DEV void fp_mul2(Fp<P>& r, const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d) {
    static_assert(P::LAZY, "fp_mul2 needs the lazy [0, 2m] representation");
    constexpr int N = P::N;
    uint32_t q[N];
    uint32_t t[N];
    uint64_t alo = 0;
    ps2_columns<P>(a, b, c, d, q, t, alo, std::make_integer_sequence<int, 2 * N>{});
#pragma unroll
    for (int j = 0; j < N; ++j) r.v[j] = t[j];
    if constexpr (P::MOD[N - 1] >= (1u << 29)) {  // 8 m > R: the sum can reach 2.52 m (BN254)
        uint32_t u[N];
        uint32_t br = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) u[j] = sub_bb(t[j], P::MOD2[j], br);
#pragma unroll
        for (int j = 0; j < N; ++j) r.v[j] = br ? t[j] : u[j];
    }
}
// r = a b - c d
template <class P>
BLZ_DEV void fp_mulsub2(Fp<P>& r, const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d) {
    Fp<P> nc;
    uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < P::N; ++j) nc.v[j] = sub_bb(P::MOD2[j], c.v[j], br);  // 2m - c in [0, 2m]
    fp_mul2(r, a, b, nc, d);
}

// ------------------------------------------------------------------------------------------------
// "Wide lazy" arithmetic for a field whose modulus is too large for the [0, 2m] representation to leave
// head-room (BLS12-381 Fr: 4m > R = 2^256, so P::LAZY is false), used by the NTT where every product is
// data x canonical twiddle:  x < 2^(32N) arbitrary, w < m  =>  (x w + q m) / R < m (1 + x/R) < 2m, so the
// product needs NO final subtraction if the data are allowed to live in [0, 2m).  Sums of two such values
// can exceed 2^(32N) (2m is most of the word), so add / sub carry one extra bit through the comparison.
// ------------------------------------------------------------------------------------------------
template <class P>
BLZ_DEV void fp_mul_nr(Fp<P>& r, const Fp<P>& x, const Fp<P>& w_canonical) {
    constexpr int N = P::N;
    uint32_t q[N];
    uint32_t t[N];
    uint64_t alo = 0;
    ps_columns<P>(x, w_canonical, q, t, alo, std::make_integer_sequence<int, 2 * N>{});
#pragma unroll
    for (int j = 0; j < N; ++j) r.v[j] = t[j];   // < 2m < 2^(32N): the word above (alo) is zero
}
// r = a + b mod-ish: a, b in [0, 2m) -> r in [0, 2m)
template <class P>
BLZ_DEV void fp_add_wide(Fp<P>& r, const Fp<P>& a, const Fp<P>& b) {
    uint32_t c = 0;
    uint32_t t[P::N];
#pragma unroll
    for (int i = 0; i < P::N; ++i) t[i] = add_cc(a.v[i], b.v[i], c);
    uint32_t u[P::N];
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) u[i] = sub_bb(t[i], P::MOD2[i], br);
    const bool keep = (c == 0) & (br != 0);   // a + b < 2m: no carry out and the subtraction borrowed
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = keep ? t[i] : u[i];
}
// r = a - b: a, b in [0, 2m) -> r in [0, 2m)
template <class P>
BLZ_DEV void fp_sub_wide(Fp<P>& r, const Fp<P>& a, const Fp<P>& b) {
    uint32_t br = 0;
    uint32_t t[P::N];
#pragma unroll
    for (int i = 0; i < P::N; ++i) t[i] = sub_bb(a.v[i], b.v[i], br);
    uint32_t c = 0;
    uint32_t mask = 0u - br;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = add_cc(t[i], P::MOD2[i] & mask, c);   // + 2m if it went negative
}
// [0, 2m) -> [0, m)
template <class P>
BLZ_DEV void fp_canon_wide(Fp<P>& a) {
    uint32_t u[P::N];
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) u[i] = sub_bb(a.v[i], P::MOD[i], br);
#pragma unroll
    for (int i = 0; i < P::N; ++i) a.v[i] = br ? a.v[i] : u[i];
}

template <class P>
BLZ_DEV void fp_mul(Fp<P>& r, const Fp<P>& a, const Fp<P>& b) { fp_mul_ps(r, a, b); }

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

// ------------------------------------------------------------------------------------------
// Inversion: binary extended Euclid on plain integers (shifts, adds, compares only).  On one lane it
// is ~25 k instructions (Kaliski's form, below) against ~510 k for Fermat's a^(m-2) (570 Montgomery products), and the
// inversions of this library all sit on single-lane latency paths (final normalisation of an MSM,
// combine of multi-GPU partials, table set-up).  Data-dependent trip count; not constant time (the
// inputs are public).  Input and output in Montgomery form; inverse of 0 is 0.
// ------------------------------------------------------------------------------------------
template <int N>
BLZ_DEV bool mp_is_one(const uint32_t (&a)[N]) {
    uint32_t o = a[0] ^ 1u;
#pragma unroll
    for (int i = 1; i < N; ++i) o |= a[i];
    return o == 0;
}
template <int N>
BLZ_DEV void mp_shr1(uint32_t (&a)[N], uint32_t top) {  // a = (top : a) >> 1
#pragma unroll
    for (int i = 0; i + 1 < N; ++i) a[i] = (a[i] >> 1) | (a[i + 1] << 31);
    a[N - 1] = (a[N - 1] >> 1) | (top << 31);
}
template <int N>
BLZ_DEV bool mp_geq(const uint32_t (&a)[N], const uint32_t (&b)[N]) {
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) (void)sub_bb(a[i], b[i], br);
    return br == 0;
}
// x = x / 2 mod m  (m odd)
template <class P>
BLZ_DEV void mp_half_mod(uint32_t (&x)[P::N]) {
    uint32_t c = 0;
    if (x[0] & 1u) {
#pragma unroll
        for (int i = 0; i < P::N; ++i) x[i] = add_cc(x[i], P::MOD[i], c);
    }
    mp_shr1<P::N>(x, c);
}
// x = (x - y) mod m,  x, y in [0, m)
template <class P>
BLZ_DEV void mp_sub_mod(uint32_t (&x)[P::N], const uint32_t (&y)[P::N]) {
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) x[i] = sub_bb(x[i], y[i], br);
    uint32_t c = 0, mask = 0u - br;
#pragma unroll
    for (int i = 0; i < P::N; ++i) x[i] = add_cc(x[i], P::MOD[i] & mask, c);
}

// bit length of the modulus
template <class P>
constexpr int mp_mod_bits() {
    int top = P::N - 1;
    while (top > 0 && P::MOD[top] == 0) --top;
    int b = 0;
    for (uint32_t x = P::MOD[top]; x; x >>= 1) ++b;
    return 32 * top + b;
}
template <int N>
BLZ_DEV void mp_shl1(uint32_t (&a)[N]) {
#pragma unroll
    for (int i = N - 1; i > 0; --i) a[i] = (a[i] << 1) | (a[i - 1] >> 31);
    a[0] <<= 1;
}
// the plain integer 2^d (d < 32 N) as a field operand
template <class P>
BLZ_DEV void fp_pow2(Fp<P>& r, int d) {
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = (d >> 5) == i ? 1u << (d & 31) : 0u;
}

// Kaliski's almost inverse (round 4; the extended Euclid with a modular halving per step that stood here was ~60 k instructions,
// 0.16 ms of every lone task's 0.68 ms Horner kernel): the cofactors r, s are only ever added and doubled - no reduction
// inside the loop, they stay below 2m - and the loop leaves A^-1 2^k mod m with n <= k <= 2n (n = bit length of m); the power of
// two comes off in the Montgomery products that put the result in Montgomery form anyway.  ~25 k instructions.
template <class P>
__device__ __noinline__ void fp_inv(Fp<P>& r_out, const Fp<P>& a_in) {
    constexpr int N = P::N, W = 32 * N, BITS = mp_mod_bits<P>();
    static_assert(BITS + 1 <= W, "the cofactors reach 2m");
    Fp<P> a = a_in;
    fp_reduce(a);
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) nz |= a.v[i];
    if (nz == 0) { fp_zero(r_out); return; }
    uint32_t u[N], v[N], r[N], s[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { u[i] = P::MOD[i]; v[i] = a.v[i]; r[i] = 0u; s[i] = i == 0 ? 1u : 0u; }
    int k = 0;
    // invariants (A = the input as a plain integer): A r == -u 2^k, A s == v 2^k (mod m); u, v > 0 until v reaches 0 with u = 1
    for (;;) {
        uint32_t vz = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) vz |= v[i];
        if (vz == 0) break;
        if ((u[0] & 1u) == 0) {
            mp_shr1<N>(u, 0);
            mp_shl1<N>(s);
        } else if ((v[0] & 1u) == 0) {
            mp_shr1<N>(v, 0);
            mp_shl1<N>(r);
        } else if (!mp_geq<N>(v, u)) {   // u > v
            uint32_t br = 0, c = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) u[i] = sub_bb(u[i], v[i], br);
            mp_shr1<N>(u, 0);
#pragma unroll
            for (int i = 0; i < N; ++i) r[i] = add_cc(r[i], s[i], c);
            mp_shl1<N>(s);
        } else {
            uint32_t br = 0, c = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = sub_bb(v[i], u[i], br);
            mp_shr1<N>(v, 0);
#pragma unroll
            for (int i = 0; i < N; ++i) s[i] = add_cc(s[i], r[i], c);
            mp_shl1<N>(r);
        }
        ++k;
    }
    // r < 2m holds -A^-1 2^k: bring it below m and negate
    {
        uint32_t d[N], br = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = sub_bb(r[i], P::MOD[i], br);
        if (br == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) r[i] = d[i];
        }
        br = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) r[i] = sub_bb(P::MOD[i], r[i], br);
    }
    // x = A^-1 2^k with A = a R: the Montgomery form of a^-1 is a^-1 R = A^-1 R^2 = x 2^(2W - k), and 6 <= 2W - k <= 2W - n.
    // A Montgomery product by R^2 multiplies by R, one by the plain integer 2^d by 2^d / R (2^d <= 2m: d <= BITS).
    Fp<P> t, r2, pw;
#pragma unroll
    for (int i = 0; i < N; ++i) { t.v[i] = r[i]; r2.v[i] = P::R2[i]; }
    int d = 2 * W - k;
    fp_mul(t, t, r2);
    if (d > BITS) {          // (k within a few bits of n: an input that is nearly a power of two; two more products)
        const int d1 = d / 2;
        fp_pow2(pw, d1);
        fp_mul(t, t, pw);
        fp_mul(t, t, r2);
        d -= d1;
    }
    fp_pow2(pw, d);
    fp_mul(r_out, t, pw);
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
// the same with the non-temporal hint (streamed once: do not keep the lines in the caches)
typedef uint32_t blz_u32x4 __attribute__((ext_vector_type(4)));
template <class P>
BLZ_DEV void fp_load_nt(Fp<P>& r, const void* p) {
    const blz_u32x4* q = reinterpret_cast<const blz_u32x4*>(p);
#pragma unroll
    for (int i = 0; i < P::N / 4; ++i) {
        blz_u32x4 x = __builtin_nontemporal_load(q + i);
        r.v[4 * i] = x.x; r.v[4 * i + 1] = x.y; r.v[4 * i + 2] = x.z; r.v[4 * i + 3] = x.w;
    }
}
template <class P>
BLZ_DEV void fp_store_nt(void* p, const Fp<P>& a) {
    blz_u32x4* q = reinterpret_cast<blz_u32x4*>(p);
#pragma unroll
    for (int i = 0; i < P::N / 4; ++i) {
        blz_u32x4 x = {a.v[4 * i], a.v[4 * i + 1], a.v[4 * i + 2], a.v[4 * i + 3]};
        __builtin_nontemporal_store(x, q + i);
    }
}
template <class P>
BLZ_DEV void fp_store(void* p, const Fp<P>& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < P::N / 4; ++i) q[i] = make_uint4(a.v[4 * i], a.v[4 * i + 1], a.v[4 * i + 2], a.v[4 * i + 3]);
}

}  // namespace blz
