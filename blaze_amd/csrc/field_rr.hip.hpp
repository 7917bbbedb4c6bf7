// Reduced-radix prime-field arithmetic for gfx950: NL limbs of B < 32 bits in 32-bit registers,
// Montgomery radix Rrr = 2^(B NL)  (BLS12-377/381 Fq: 14 x 28 bits, bucket accumulation and first reduce level;
// the three scalar fields: 9 x 29 bits, the 2^27 NTT).
//
// Why a second representation.  With full 32-bit limbs (field.hip.hpp) a column sum of 32x32 products
// needs 64 + log2(count) bits, so every v_mad_u64_u32 is followed by a v_addc_co_u32 that folds its
// carry-out into a third accumulator word: 2 issue slots per multiply-add, and both are "long"
// (VOP3-class) instructions that issue at half the rate of a plain 32-bit add on this chip.  With B-bit
// limbs a product is < 2^(2B + slack) and a whole column (2 NL products) fits the 64-bit accumulator of
// v_mad_u64_u32 itself: ONE instruction per multiply-add, no carry word, and the per-column
// bookkeeping (mask, shift) is 3 more.  14 x 14 x 2 = 392 multiply-adds instead of 12 x 12 x 2 x 2 = 576
// instructions for a BLS base-field product: 7.8e10 instead of 6.2e10 products/s on the chip, which is
// the bare v_mad_u64_u32 rate (3.1e13 lane-ops/s / 392); a squaring does its off-diagonal products once
// against the doubled operand (301 multiply-adds: 9.65e10 /s).  profiles/r02_mul_variants.txt holds
// these and the alternatives that lost (13 x 30 bits two-phase, DFMA, v_lshl_add_u64 column sums).
// The spare bits also make add / sub carry-free: limbs are allowed to grow ("limb slack") and values
// are allowed to grow ("value slack", Rrr / m = 2^HEAD) between products; only a product renormalises.
//
// Ranges are part of the TYPE, so every bound below is checked at compile time:
//   Frr<Q, F, V>:  every limb < F 2^B   and   the integer < V m.
//   F = 1: "normalised" (the top limb holds whatever is left; < 2^B because V m < Rrr).
// rr_mul(a, b):    needs (Fa Fb + 1) NL + 1 <= 2^(64 - 2B)  (64-bit column sums) and Va Vb <= 2^HEAD;
//                  result Frr<Q, 1, 2>  (< (Va Vb / 2^HEAD + 1) m).
// rr_sub<J>(a, b): a - b + 2^J m, b normalised with Vb <= 2^(J-1): Frr<Q, Fa + 2, Va + 2^J>.
// Values are never canonical in flight; zero tests are done modulo m on demand (rr_is_zero).
#pragma once
#include "field.hip.hpp"

namespace blz {

template <class Q>
constexpr int rr_head() { return Q::B * Q::NL - Q::BITS; }  // log2(Rrr / 2^BITS) <= log2(Rrr / m)

template <class Q, int F = 1, int V = 2>
struct Frr {
    static_assert(F >= 1 && F < (1 << (32 - Q::B)) && V >= 1 && V <= (1 << rr_head<Q>()), "bound out of range");
    uint32_t v[Q::NL];
};

// 64-bit column sums: fsum = sum over the product kinds of Fa Fb (1 kind for a b, 2 for a b + c d)
template <class Q>
constexpr bool rr_cols_ok(unsigned fsum) {
    return (unsigned long long)(fsum + 1) * Q::NL + 1 <= (1ull << (64 - 2 * Q::B));
}
template <class Q>
constexpr bool rr_vals_ok(unsigned vsum) { return vsum <= (1u << rr_head<Q>()); }

// widen the bounds (free)
template <int F2, int V2, class Q, int F, int V>
BLZ_DEV const Frr<Q, F2, V2>& rr_as(const Frr<Q, F, V>& a) {
    static_assert(F2 >= F && V2 >= V, "rr_as only widens");
    return reinterpret_cast<const Frr<Q, F2, V2>&>(a);
}

template <class Q, int F, int V>
BLZ_DEV void rr_zero(Frr<Q, F, V>& r) {
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) r.v[i] = 0;
}
template <class Q, int F, int V>
BLZ_DEV void rr_one(Frr<Q, F, V>& r) {  // Montgomery one
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) r.v[i] = Q::ONE[i];
}
template <class Q, int F, int V>
BLZ_DEV bool rr_all_zero(const Frr<Q, F, V>& a) {  // literal zero limbs
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) o |= a.v[i];
    return o == 0;
}

#include "rr_gen.inc"

// 1: plain C++ products, scheduled by hipcc (A/B experiment: k_accumulate 119 ms against 106 ms with the asm
// columns at 2^26; tools/gen_rr_asm.py's header says why)
#ifndef BLZ_RR_PLAIN
#define BLZ_RR_PLAIN 0
#endif

// ---- product scanning over asm columns ----------------------------------------------------------------
// column K: the caller's products (AB), the reduction products q_i m_j of this column, then either the next
// quotient digit q_K (K < NL) or the next result limb
template <class Q, int K, class AB>
BLZ_DEV void rr_column(uint64_t& acc, uint32_t (&q)[Q::NL], uint32_t (&t)[Q::NL], AB&& ab) {
    constexpr int NL = Q::NL;
    ab(std::integral_constant<int, K>{}, acc);
#if BLZ_RR_PLAIN
    {
        constexpr int ilo = K < NL ? 0 : K - NL + 1, ihi = K < NL ? K - 1 : NL - 1;
#pragma unroll
        for (int i = ilo; i <= ihi; ++i) acc += (uint64_t)q[i] * Q::MOD[K - i];
    }
#else
    rr_qm<NL, K>(acc, q, Q::MOD);
#endif
    if constexpr (K < NL) {
        q[K] = ((uint32_t)acc * Q::N0) & Q::MASK;
        acc = (uint64_t)q[K] * Q::MOD[0] + acc;  // low B bits become zero
    } else {
        t[K - NL] = (uint32_t)acc & Q::MASK;
    }
    acc >>= Q::B;
}
template <class Q, class AB, int... Ks>
BLZ_DEV void rr_columns(Frr<Q, 1, 2>& r, AB&& ab, std::integer_sequence<int, Ks...>) {
    uint32_t q[Q::NL], t[Q::NL];
    uint64_t acc = 0;
    (rr_column<Q, Ks>(acc, q, t, ab), ...);
    t[Q::NL - 1] = (uint32_t)acc;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) r.v[i] = t[i];
}

#if BLZ_RR_PLAIN
template <int NL, int K> BLZ_DEV void rr_ab_c(uint64_t& acc, const uint32_t (&a)[NL], const uint32_t (&b)[NL]) {
    constexpr int ilo = K < NL ? 0 : K - NL + 1, ihi = K < NL ? K : NL - 1;
#pragma unroll
    for (int i = ilo; i <= ihi; ++i) acc += (uint64_t)a[i] * b[K - i];
}
template <int NL, int K> BLZ_DEV void rr_sq_c(uint64_t& acc, const uint32_t (&a)[NL], const uint32_t (&a2)[NL]) {
    constexpr int ilo = K < NL ? 0 : K - NL + 1, ihi = K < NL ? K : NL - 1;
#pragma unroll
    for (int i = ilo; i <= ihi; ++i) if (i < K - i) acc += (uint64_t)a[i] * a2[K - i];
    if constexpr (K % 2 == 0) acc += (uint64_t)a[K / 2] * a[K / 2];
}
#define BLZ_RR_AB rr_ab_c
#define BLZ_RR_SQ rr_sq_c
#else
#define BLZ_RR_AB rr_ab
#define BLZ_RR_SQ rr_sq
#endif
// the plain product's columns: caller's products and reduction products in one asm statement where the operand count
// allows (rr_gen.inc rr_fused_ok), then the column's tail as in rr_column
template <class Q, int K>
BLZ_DEV void rr_column_mul(uint64_t& acc, uint32_t (&q)[Q::NL], uint32_t (&t)[Q::NL], const uint32_t (&a)[Q::NL],
                           const uint32_t (&b)[Q::NL]) {
    constexpr int NL = Q::NL;
#if BLZ_RR_PLAIN || defined(BLZ_RR_NO_FUSE)
    rr_column<Q, K>(acc, q, t, [&](auto k, uint64_t& c) { BLZ_RR_AB<NL, decltype(k)::value>(c, a, b); });
#else
    if constexpr (rr_fused_ok(NL, K)) {
        rr_abqm<NL, K>(acc, a, b, q, Q::MOD);
        if constexpr (K < NL) {
            q[K] = ((uint32_t)acc * Q::N0) & Q::MASK;
            acc = (uint64_t)q[K] * Q::MOD[0] + acc;
        } else {
            t[K - NL] = (uint32_t)acc & Q::MASK;
        }
        acc >>= Q::B;
    } else {
        rr_column<Q, K>(acc, q, t, [&](auto k, uint64_t& c) { BLZ_RR_AB<NL, decltype(k)::value>(c, a, b); });
    }
#endif
}
template <class Q, int... Ks>
BLZ_DEV void rr_columns_mul(Frr<Q, 1, 2>& r, const uint32_t (&a)[Q::NL], const uint32_t (&b)[Q::NL], std::integer_sequence<int, Ks...>) {
    uint32_t q[Q::NL], t[Q::NL];
    uint64_t acc = 0;
    (rr_column_mul<Q, Ks>(acc, q, t, a, b), ...);
    t[Q::NL - 1] = (uint32_t)acc;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) r.v[i] = t[i];
}
// r = a b / Rrr  (mod m)
template <class Q, int Fa, int Va, int Fb, int Vb>
BLZ_DEV void rr_mul(Frr<Q, 1, 2>& r, const Frr<Q, Fa, Va>& a, const Frr<Q, Fb, Vb>& b) {
    static_assert(rr_cols_ok<Q>(Fa * Fb), "column sum would overflow 64 bits: normalise an operand");
    static_assert(rr_vals_ok<Q>(Va * Vb), "product would leave the lazy value range");
    rr_columns_mul<Q>(r, a.v, b.v, std::make_integer_sequence<int, 2 * Q::NL - 1>{});
}
// r = a^2 / Rrr: the off-diagonal products once, against the doubled operand
template <class Q, int Fa, int Va>
BLZ_DEV void rr_sqr(Frr<Q, 1, 2>& r, const Frr<Q, Fa, Va>& a) {
    static_assert(rr_cols_ok<Q>(Fa * Fa) && 2 * Fa < (1 << (32 - Q::B)), "column sum would overflow 64 bits");
    static_assert(rr_vals_ok<Q>(Va * Va), "product would leave the lazy value range");
    uint32_t a2[Q::NL];
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) a2[i] = a.v[i] << 1;
    rr_columns<Q>(r, [&](auto k, uint64_t& acc) { BLZ_RR_SQ<Q::NL, decltype(k)::value>(acc, a.v, a2); },
                  std::make_integer_sequence<int, 2 * Q::NL - 1>{});
}
// r = (a b + c d) / Rrr with ONE reduction (the group law's Y3 = R (Q - X3) - Y1 PPP)
template <class Q, int Fa, int Va, int Fb, int Vb, int Fc, int Vc, int Fd, int Vd>
BLZ_DEV void rr_mul2(Frr<Q, 1, 2>& r, const Frr<Q, Fa, Va>& a, const Frr<Q, Fb, Vb>& b, const Frr<Q, Fc, Vc>& c,
                     const Frr<Q, Fd, Vd>& d) {
    static_assert(rr_cols_ok<Q>(Fa * Fb + Fc * Fd), "column sum would overflow 64 bits");
    static_assert(rr_vals_ok<Q>(Va * Vb + Vc * Vd), "sum of products would leave the lazy value range");
    rr_columns<Q>(r, [&](auto k, uint64_t& acc) {
        BLZ_RR_AB<Q::NL, decltype(k)::value>(acc, a.v, b.v);
        BLZ_RR_AB<Q::NL, decltype(k)::value>(acc, c.v, d.v);
    }, std::make_integer_sequence<int, 2 * Q::NL - 1>{});
}

// ---- Shoup product: x times a CONSTANT known in advance (the NTT's table twiddles) -----------------------------------
// For a canonical w the table also holds wq = floor(w Rrr / m).  Then q = floor(x wq / Rrr) is floor(x w / m) or one
// below it, and x w - q m is the product in [0, 2m).  Here: the high half of x wq is summed from two guard columns up
// (NL^2 - 28 = 53 multiply-adds at NL = 9: the columns left out are worth < 2^-26 of q's unit, so q is that floor or one
// below), then x w + q (Rrr - m), i.e. x w - q m modulo Rrr, in NL columns (90 multiply-adds): 143 against the Montgomery
// product's 153, no quotient-digit chain between the columns (+14.6 % products per second at two waves per SIMD,
// profiles/r03_shoup_probe.txt), the result the plain product x w - no Montgomery factor.  The truncation can leave the
// result in [2m, 3m): a top-limb compare (T2M) and a rarely taken subtraction of m bring every result below 2m.
// x: any limbs the column bound admits (F <= 6 at 9 x 29 bits), value < Rrr (guaranteed by the type: V m < Rrr).
template <class Q>
struct RRShoup {
    uint32_t w[Q::NL];   // the twiddle, canonical
    uint32_t wq[Q::NL];  // floor(w Rrr / m)
};
template <class Q, int K>
BLZ_DEV void rr_shoup_hi(uint64_t& acc, uint32_t (&q)[Q::NL], const uint32_t (&x)[Q::NL], const uint32_t (&wq)[Q::NL]) {
    rr_ab<Q::NL, K>(acc, x, wq);
    if constexpr (K >= Q::NL) q[K - Q::NL] = (uint32_t)acc & Q::MASK;
    acc >>= Q::B;
}
template <class Q, int K>
BLZ_DEV void rr_shoup_lo(uint64_t& acc, uint32_t (&r)[Q::NL], const uint32_t (&x)[Q::NL], const uint32_t (&w)[Q::NL],
                         const uint32_t (&q)[Q::NL]) {
    if constexpr (rr_lo2_ok(K)) {
        rr_abqs<Q::NL, K>(acc, x, w, q, Q::MBAR);
    } else {
        rr_ab<Q::NL, K>(acc, x, w);
        rr_as<Q::NL, K>(acc, q, Q::MBAR);
    }
    r[K] = (uint32_t)acc & Q::MASK;
    acc >>= Q::B;
}
template <class Q, int... Hs, int... Ls>
BLZ_DEV void rr_shoup_columns(uint32_t (&r)[Q::NL], const uint32_t (&x)[Q::NL], const RRShoup<Q>& t, std::integer_sequence<int, Hs...>,
                              std::integer_sequence<int, Ls...>) {
    uint32_t q[Q::NL];
    uint64_t acc = 0;
    (rr_shoup_hi<Q, Q::NL - 2 + Hs>(acc, q, x, t.wq), ...);   // columns NL - 2 .. 2 NL - 2
    q[Q::NL - 1] = (uint32_t)acc;
    acc = 0;
    (rr_shoup_lo<Q, Ls>(acc, r, x, t.w, q), ...);             // columns 0 .. NL - 1
}
// The same entry held in SGPRs: a twiddle every lane of the wave shares (the w8 powers inside an 8-point DFT).  Its
// multiply-adds take the constant as their scalar operand (rr_as); as a VGPR operand each use cost a v_mov from the SGPR the
// compiler keeps the uniform value in: 18 per product, 15 products per lane and pass.
template <class Q>
struct RRShoupU {
    uint32_t w[Q::NL];
    uint32_t wq[Q::NL];
};
template <class Q>
BLZ_DEV void rr_shoup_uniform(RRShoupU<Q>& u, const RRShoup<Q>& t) {
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) {
        u.w[i] = __builtin_amdgcn_readfirstlane(t.w[i]);
        u.wq[i] = __builtin_amdgcn_readfirstlane(t.wq[i]);
    }
}
template <class Q, int K>
BLZ_DEV void rr_shoup_hi_u(uint64_t& acc, uint32_t (&q)[Q::NL], const uint32_t (&x)[Q::NL], const uint32_t (&wq)[Q::NL]) {
    rr_as<Q::NL, K>(acc, x, wq);
    if constexpr (K >= Q::NL) q[K - Q::NL] = (uint32_t)acc & Q::MASK;
    acc >>= Q::B;
}
template <class Q, int K>
BLZ_DEV void rr_shoup_lo_u(uint64_t& acc, uint32_t (&r)[Q::NL], const uint32_t (&x)[Q::NL], const uint32_t (&w)[Q::NL],
                           const uint32_t (&q)[Q::NL]) {
    if constexpr (rr_lo2_ok(K)) {
        rr_asqs<Q::NL, K>(acc, x, w, q, Q::MBAR);
    } else {
        rr_as<Q::NL, K>(acc, x, w);
        rr_as<Q::NL, K>(acc, q, Q::MBAR);
    }
    r[K] = (uint32_t)acc & Q::MASK;
    acc >>= Q::B;
}
template <class Q, int... Hs, int... Ls>
BLZ_DEV void rr_shoup_columns_u(uint32_t (&r)[Q::NL], const uint32_t (&x)[Q::NL], const RRShoupU<Q>& t, std::integer_sequence<int, Hs...>,
                                std::integer_sequence<int, Ls...>) {
    uint32_t q[Q::NL];
    uint64_t acc = 0;
    (rr_shoup_hi_u<Q, Q::NL - 2 + Hs>(acc, q, x, t.wq), ...);
    q[Q::NL - 1] = (uint32_t)acc;
    acc = 0;
    (rr_shoup_lo_u<Q, Ls>(acc, r, x, t.w, q), ...);
}
template <class Q>
BLZ_DEV void rr_shoup_fix(Frr<Q, 1, 2>& r, uint32_t (&o)[Q::NL]) {
    if (__builtin_expect(o[Q::NL - 1] >= Q::T2M, 0)) {   // in [2m - eps, 3m): take m off (top limb < T2M means < 2m)
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < Q::NL - 1; ++i) {
            const uint32_t d = o[i] - Q::MOD[i] - borrow;
            borrow = d >> 31;
            o[i] = d & Q::MASK;
        }
        o[Q::NL - 1] = o[Q::NL - 1] - Q::MOD[Q::NL - 1] - borrow;
    }
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) r.v[i] = o[i];
}
template <class Q, int F, int V>
BLZ_DEV void rr_mul_shoup(Frr<Q, 1, 2>& r, const Frr<Q, F, V>& x, const RRShoupU<Q>& t) {
    static_assert(Q::NL == 9, "the constant-operand columns (rr_as) are generated for 9 limbs");
    static_assert(rr_cols_ok<Q>(F), "column sum would overflow 64 bits: normalise x");
    uint32_t o[Q::NL];
    rr_shoup_columns_u<Q>(o, x.v, t, std::make_integer_sequence<int, Q::NL + 1>{}, std::make_integer_sequence<int, Q::NL>{});
    rr_shoup_fix<Q>(r, o);
}
template <class Q, int F, int V>
BLZ_DEV void rr_mul_shoup(Frr<Q, 1, 2>& r, const Frr<Q, F, V>& x, const RRShoup<Q>& t) {
    static_assert(Q::NL == 9, "the constant-operand columns (rr_as) are generated for 9 limbs");
    static_assert(rr_cols_ok<Q>(F), "column sum would overflow 64 bits: normalise x");
    uint32_t o[Q::NL];
    rr_shoup_columns<Q>(o, x.v, t, std::make_integer_sequence<int, Q::NL + 1>{}, std::make_integer_sequence<int, Q::NL>{});
    if (__builtin_expect(o[Q::NL - 1] >= Q::T2M, 0)) {   // in [2m - eps, 3m): take m off (top limb < T2M means < 2m)
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < Q::NL - 1; ++i) {
            const uint32_t d = o[i] - Q::MOD[i] - borrow;
            borrow = d >> 31;
            o[i] = d & Q::MASK;
        }
        o[Q::NL - 1] = o[Q::NL - 1] - Q::MOD[Q::NL - 1] - borrow;
    }
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) r.v[i] = o[i];
}
// wq = floor(w Rrr / m) for a canonical w, by B NL shift-and-subtract steps (table set-up only)
template <class Q>
BLZ_DEV void rr_shoup_quot(RRShoup<Q>& t, const Frr<Q, 1, 1>& w) {
    constexpr int NL = Q::NL, B = Q::B;
    uint32_t rem[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        t.w[i] = w.v[i];
        rem[i] = w.v[i];
        t.wq[i] = 0;
    }
    for (int step = 0; step < B * NL; ++step) {
        uint32_t ct = 0, cq = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const uint32_t nt = (rem[i] << 1) | ct, nq = (t.wq[i] << 1) | cq;
            ct = nt >> B;
            cq = nq >> B;
            rem[i] = i == NL - 1 ? nt : (nt & Q::MASK);   // rem < 2m: the top limb stays inside its register
            t.wq[i] = nq & Q::MASK;
        }
        uint32_t d[NL], borrow = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const uint32_t x = rem[i] - Q::MOD[i] - borrow;
            borrow = x >> 31;
            d[i] = i == NL - 1 ? x : (x & Q::MASK);
        }
        if (!borrow) {
#pragma unroll
            for (int i = 0; i < NL; ++i) rem[i] = d[i];
            t.wq[0] |= 1u;
        }
    }
}

// ---- two independent products, column by column in one instruction stream ------------------------------------------
// A product is ONE dependent chain through its 64-bit column accumulator (every multiply-add waits for the one before, the
// quotient digit for the column's sum); with two waves per SIMD the second wave covers most of that latency, not all.  Where
// the group law has two products that do not depend on each other, their columns CAN alternate: two chains per wave
// (-DBLZ_RR_PAIR).  Measured (round 3, 2^26 BLS12-381, same box): k_accumulate alone 104.0 -> 103.4 ms, but the second
// chain's registers take the kernel from 198 to 211 VGPRs, and with 2 x 216 of a SIMD's 512 allocated the next task's
// hidden digit sort no longer runs beside it (its level-2 scatter waits for the accumulation to end): 118.9 -> 122.7 ms
// per step.  Off by default: the products of a pair run one after the other.
template <class Q, class AB1, class AB2, int... Ks>
BLZ_DEV void rr_columns_pair(Frr<Q, 1, 2>& r1, AB1&& ab1, Frr<Q, 1, 2>& r2, AB2&& ab2, std::integer_sequence<int, Ks...>) {
    uint32_t q1[Q::NL], t1[Q::NL], q2[Q::NL], t2[Q::NL];
    uint64_t acc1 = 0, acc2 = 0;
    ((rr_column<Q, Ks>(acc1, q1, t1, ab1), rr_column<Q, Ks>(acc2, q2, t2, ab2)), ...);
    t1[Q::NL - 1] = (uint32_t)acc1;
    t2[Q::NL - 1] = (uint32_t)acc2;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) {
        r1.v[i] = t1[i];
        r2.v[i] = t2[i];
    }
}
// r1 = a1 b1, r2 = a2 b2
template <class Q, int Fa1, int Va1, int Fb1, int Vb1, int Fa2, int Va2, int Fb2, int Vb2>
BLZ_DEV void rr_mul_pair(Frr<Q, 1, 2>& r1, const Frr<Q, Fa1, Va1>& a1, const Frr<Q, Fb1, Vb1>& b1, Frr<Q, 1, 2>& r2,
                         const Frr<Q, Fa2, Va2>& a2, const Frr<Q, Fb2, Vb2>& b2) {
#ifndef BLZ_RR_PAIR
    rr_mul(r1, a1, b1);
    rr_mul(r2, a2, b2);
#else
    static_assert(rr_cols_ok<Q>(Fa1 * Fb1) && rr_cols_ok<Q>(Fa2 * Fb2), "column sum would overflow 64 bits: normalise an operand");
    static_assert(rr_vals_ok<Q>(Va1 * Vb1) && rr_vals_ok<Q>(Va2 * Vb2), "product would leave the lazy value range");
    rr_columns_pair<Q>(r1, [&](auto k, uint64_t& c) { BLZ_RR_AB<Q::NL, decltype(k)::value>(c, a1.v, b1.v); },
                       r2, [&](auto k, uint64_t& c) { BLZ_RR_AB<Q::NL, decltype(k)::value>(c, a2.v, b2.v); },
                       std::make_integer_sequence<int, 2 * Q::NL - 1>{});
#endif
}
// r1 = a1^2, r2 = a2^2
template <class Q, int Fa1, int Va1, int Fa2, int Va2>
BLZ_DEV void rr_sqr_pair(Frr<Q, 1, 2>& r1, const Frr<Q, Fa1, Va1>& a1, Frr<Q, 1, 2>& r2, const Frr<Q, Fa2, Va2>& a2) {
#ifndef BLZ_RR_PAIR
    rr_sqr(r1, a1);
    rr_sqr(r2, a2);
#else
    static_assert(rr_cols_ok<Q>(Fa1 * Fa1) && 2 * Fa1 < (1 << (32 - Q::B)) && rr_cols_ok<Q>(Fa2 * Fa2) && 2 * Fa2 < (1 << (32 - Q::B)),
                  "column sum would overflow 64 bits");
    static_assert(rr_vals_ok<Q>(Va1 * Va1) && rr_vals_ok<Q>(Va2 * Va2), "product would leave the lazy value range");
    uint32_t d1[Q::NL], d2[Q::NL];
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) {
        d1[i] = a1.v[i] << 1;
        d2[i] = a2.v[i] << 1;
    }
    rr_columns_pair<Q>(r1, [&](auto k, uint64_t& c) { BLZ_RR_SQ<Q::NL, decltype(k)::value>(c, a1.v, d1); },
                       r2, [&](auto k, uint64_t& c) { BLZ_RR_SQ<Q::NL, decltype(k)::value>(c, a2.v, d2); },
                       std::make_integer_sequence<int, 2 * Q::NL - 1>{});
#endif
}

// the same product in plain C++ (what hipcc schedules by itself: kept as the readable reference and for
// tools/mul_variants.hip; it re-associates every column into "products first, carry-in last")
template <class Q, int Fa, int Va, int Fb, int Vb>
BLZ_DEV void rr_mul_ref(Frr<Q, 1, 2>& r, const Frr<Q, Fa, Va>& a, const Frr<Q, Fb, Vb>& b) {
    constexpr int NL = Q::NL, B = Q::B;
    uint32_t q[NL], t[NL];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; ++k) {
        const int ilo = k < NL ? 0 : k - NL + 1;
        const int ihi = k < NL ? k : NL - 1;
#pragma unroll
        for (int i = ilo; i <= ihi; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
        if (k < NL) {
#pragma unroll
            for (int i = 0; i < k; ++i) acc += (uint64_t)q[i] * Q::MOD[k - i];
            q[k] = ((uint32_t)acc * Q::N0) & Q::MASK;
            acc += (uint64_t)q[k] * Q::MOD[0];
        } else {
#pragma unroll
            for (int i = k - NL + 1; i < NL; ++i) acc += (uint64_t)q[i] * Q::MOD[k - i];
            t[k - NL] = (uint32_t)acc & Q::MASK;
        }
        acc >>= B;
    }
    t[NL - 1] = (uint32_t)acc;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.v[i] = t[i];
}

// ---- carry-free add / sub ---------------------------------------------------------------------------
template <class Q, int Fa, int Va, int Fb, int Vb>
BLZ_DEV Frr<Q, Fa + Fb, Va + Vb> rr_add(const Frr<Q, Fa, Va>& a, const Frr<Q, Fb, Vb>& b) {
    Frr<Q, Fa + Fb, Va + Vb> r;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) r.v[i] = a.v[i] + b.v[i];
    return r;
}
// a - b + 2^J m.  KM[J-1] is 2^J m in borrow form: every limb but the top is >= 2^B - 1 >= b_i and the top limb
// is >= the top limb of any b < 2^(J-1) m (tools/gen_constants.py), so no limb goes negative.
template <int J, class Q, int Fa, int Va, int Vb>
BLZ_DEV Frr<Q, Fa + 2, Va + (1 << J)> rr_sub(const Frr<Q, Fa, Va>& a, const Frr<Q, 1, Vb>& b) {
    static_assert(J >= 1 && J <= Q::NKM && Vb <= (1 << (J - 1)), "multiple of m too small for this subtrahend");
    Frr<Q, Fa + 2, Va + (1 << J)> r;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) r.v[i] = a.v[i] + (Q::KM[J - 1][i] - b.v[i]);
    return r;
}
// a - 2 b + 2^(J+1) m
template <int J, class Q, int Fa, int Va, int Vb>
BLZ_DEV Frr<Q, Fa + 4, Va + (2 << J)> rr_sub_twice(const Frr<Q, Fa, Va>& a, const Frr<Q, 1, Vb>& b) {
    static_assert(J >= 1 && J <= Q::NKM && Vb <= (1 << (J - 1)), "multiple of m too small for this subtrahend");
    Frr<Q, Fa + 4, Va + (2 << J)> r;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) r.v[i] = a.v[i] + 2u * (Q::KM[J - 1][i] - b.v[i]);
    return r;
}
// 2^J m - b
template <int J, class Q, int Vb>
BLZ_DEV Frr<Q, 2, (1 << J)> rr_neg(const Frr<Q, 1, Vb>& b) {
    static_assert(J >= 1 && J <= Q::NKM && Vb <= (1 << (J - 1)), "multiple of m too small");
    Frr<Q, 2, (1 << J)> r;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) r.v[i] = Q::KM[J - 1][i] - b.v[i];
    return r;
}
// sign ? 2^J m - b : b   (per lane)
template <int J, class Q, int Vb>
BLZ_DEV Frr<Q, 2, (1 << J)> rr_cneg(const Frr<Q, 1, Vb>& b, bool sign) {
    static_assert(J >= 1 && J <= Q::NKM && Vb <= (1 << (J - 1)), "multiple of m too small");
    Frr<Q, 2, (1 << J)> r;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) r.v[i] = sign ? Q::KM[J - 1][i] - b.v[i] : b.v[i];
    return r;
}
// carry propagation: limbs < 2^B again (the top limb keeps the rest; V m < Rrr by the type's own bound)
template <class Q, int F, int V>
BLZ_DEV Frr<Q, 1, V> rr_norm(const Frr<Q, F, V>& a) {
    Frr<Q, 1, V> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < Q::NL - 1; ++i) {
        uint32_t s = a.v[i] + c;
        r.v[i] = s & Q::MASK;
        c = s >> Q::B;
    }
    r.v[Q::NL - 1] = a.v[Q::NL - 1] + c;
    return r;
}

// x < V m  ->  the same residue below 2m, normalised, WITHOUT a product: one quotient digit estimated from the top limb,
// q = floor(x_top MU / 2^32) with MU = floor(2^32 / (m_top + 1)), then x - q m limb by limb.  q never exceeds
// floor(x / m) and falls short of it by less than (V + 1) / m_top + x_top / 2^32 <= 1/2, i.e. by at most 1.
// NL multiply-adds + ~8 NL plain instructions, against 2 NL^2 multiply-adds for a product by one: the way an 8-point
// DFT's un-twiddled output (and a forward transform's last output) gets back into the lazy range.
template <class Q, int F, int V>
BLZ_DEV Frr<Q, 1, 2> rr_reduce2m(const Frr<Q, F, V>& x) {
    constexpr int NL = Q::NL, B = Q::B;
    // the two error terms of the estimate, each held below 1/4
    static_assert(4ull * (V + 1) <= Q::MOD[NL - 1], "top limb of m too short for the quotient estimate");
    static_assert((V + 1ull) * (Q::MOD[NL - 1] + 1ull) <= (1ull << 30), "x_top too large for the 32-bit reciprocal");
    const Frr<Q, 1, V> a = rr_norm(x);
    constexpr uint32_t MU = (uint32_t)((1ull << 32) / (Q::MOD[NL - 1] + 1ull));
    const uint32_t q = __umulhi(a.v[NL - 1], MU);
    Frr<Q, 1, 2> r;
    uint64_t p = 0;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) {
        p = (uint64_t)q * Q::MOD[i] + (p >> B);           // limb i of q m, in the low B bits
        const uint32_t d = a.v[i] - ((uint32_t)p & Q::MASK) - borrow;
        borrow = d >> 31;                                 // both terms are < 2^B <= 2^31: negative <=> bit 31
        r.v[i] = d & Q::MASK;
    }
    p = (uint64_t)q * Q::MOD[NL - 1] + (p >> B);
    r.v[NL - 1] = a.v[NL - 1] - (uint32_t)p - borrow;
    return r;
}

// ---- conversions -----------------------------------------------------------------------------------
// plain integer in 32-bit words (any value < 2^(32 N32)) -> B-bit limbs, normalised
template <class Q, int V>
BLZ_DEV void rr_from_words(Frr<Q, 1, V>& r, const uint32_t (&w)[Q::N32]) {
    static_assert(32 * Q::N32 <= Q::B * Q::NL, "words do not fit the limbs");
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) {
        const int bit = i * Q::B, j = bit >> 5, s = bit & 31;
        uint32_t lo = j < Q::N32 ? w[j] : 0u;
        uint32_t hi = j + 1 < Q::N32 ? w[j + 1] : 0u;
        uint32_t x = s == 0 ? lo : __builtin_amdgcn_alignbit(hi, lo, s);
        r.v[i] = x & Q::MASK;
    }
}
// normalised B-bit limbs of a value < 2^(32 N32) -> 32-bit words
template <class Q, int V>
BLZ_DEV void rr_to_words(uint32_t (&w)[Q::N32], const Frr<Q, 1, V>& a) {
    static_assert(Q::BITS + 1 <= 32 * Q::N32 && V <= 2, "value does not fit the words");
#pragma unroll
    for (int j = 0; j < Q::N32; ++j) {
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < Q::NL; ++i) {
            const int sh = i * Q::B - 32 * j;  // position of limb i inside word j
            if (sh > -Q::B && sh < 32) x |= sh >= 0 ? (a.v[i] << sh) : (a.v[i] >> (-sh));
        }
        w[j] = x;
    }
}

// wire integer (32-bit words; any 32 N32-bit value, canonical or not) -> Montgomery form x Rrr mod m
template <class Q>
BLZ_DEV void rr_to_mont_from_words(Frr<Q, 1, 2>& r, const uint32_t (&w)[Q::N32]) {
    // x < 2^(32 N32) = 2^(32 N32 - BITS) 2^BITS < 2^(32 N32 - BITS + 1) m
    constexpr int VRAW = 1 << (32 * Q::N32 - Q::BITS + 1);
    Frr<Q, 1, VRAW> x;
    rr_from_words<Q>(x, w);
    Frr<Q, 1, 1> k;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) k.v[i] = Q::RR2[i];
    rr_mul(r, x, k);
}
// x Rrr (normalised) -> x R32 as 32-bit words in the lazy range [0, 2m) of field.hip.hpp's twin field
template <class Q, int V>
BLZ_DEV void rr_to_mont32_words(uint32_t (&w)[Q::N32], const Frr<Q, 1, V>& a) {
    Frr<Q, 1, 1> k;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) k.v[i] = Q::TO32[i];
    Frr<Q, 1, 2> t;
    rr_mul(t, a, k);
    rr_to_words<Q>(w, t);
}
// x R32 in 32-bit words (lazy [0, 2m]) -> x Rrr
template <class Q>
BLZ_DEV void rr_from_mont32_words(Frr<Q, 1, 2>& r, const uint32_t (&w)[Q::N32]) {
    Frr<Q, 1, 3> x;
    rr_from_words<Q>(x, w);
    Frr<Q, 1, 1> k;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) k.v[i] = Q::FROM32[i];
    rr_mul(r, x, k);
}

// exact test a == 0 (mod m): a / Rrr (mod m) is < 2m after one reduction, so it is 0 or m exactly
template <class Q, int F, int V>
BLZ_DEV bool rr_is_zero(const Frr<Q, F, V>& a) {
    Frr<Q, 1, 1> one;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) one.v[i] = i == 0 ? 1u : 0u;
    Frr<Q, 1, 2> t;
    rr_mul(t, a, one);
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) {
        z |= t.v[i];
        e |= t.v[i] ^ Q::MOD[i];
    }
    return z == 0 || e == 0;
}
// cheap filter: can a - b be 0 (mod m)?  a - b = c m for an integer -Vb < c < Va, and c = (a_0 - b_0) m^-1 mod 2^B
// whatever the limb slack (only the low B bits of the low limbs matter).  False positives ~ (Va + Vb) / 2^B.
template <class Q, int Fa, int Va, int Fb, int Vb>
BLZ_DEV bool rr_maybe_equal(const Frr<Q, Fa, Va>& a, const Frr<Q, Fb, Vb>& b) {
    const uint32_t c = ((a.v[0] - b.v[0]) * Q::MINV) & Q::MASK;
    return c < (uint32_t)Va || c > Q::MASK - (uint32_t)Vb;
}

// global-memory I/O: NL dwords, 16-byte vector accesses where the count allows.  Arrays of elements use a stride of
// rr_stride<Q>() dwords (NL rounded up to even: 8-byte alignment for every element).
template <class Q>
constexpr int rr_stride() { return (Q::NL + 1) & ~1; }
template <class Q, int F, int V>
BLZ_DEV void rr_load(Frr<Q, F, V>& r, const uint32_t* p) {
    constexpr int NL = Q::NL;
    const uint4* q4 = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < NL / 4; ++i) {
        uint4 x = q4[i];
        r.v[4 * i] = x.x; r.v[4 * i + 1] = x.y; r.v[4 * i + 2] = x.z; r.v[4 * i + 3] = x.w;
    }
    if constexpr (NL % 4 >= 2) {
        uint2 x = *reinterpret_cast<const uint2*>(p + (NL & ~3));
        r.v[NL & ~3] = x.x; r.v[(NL & ~3) + 1] = x.y;
    }
    if constexpr (NL % 2 == 1) r.v[NL - 1] = p[NL - 1];
}
template <class Q, int F, int V>
BLZ_DEV void rr_store(uint32_t* p, const Frr<Q, F, V>& a) {
    constexpr int NL = Q::NL;
    uint4* q4 = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < NL / 4; ++i) q4[i] = make_uint4(a.v[4 * i], a.v[4 * i + 1], a.v[4 * i + 2], a.v[4 * i + 3]);
    if constexpr (NL % 4 >= 2) *reinterpret_cast<uint2*>(p + (NL & ~3)) = make_uint2(a.v[NL & ~3], a.v[(NL & ~3) + 1]);
    if constexpr (NL % 2 == 1) p[NL - 1] = a.v[NL - 1];
}

}  // namespace blz
