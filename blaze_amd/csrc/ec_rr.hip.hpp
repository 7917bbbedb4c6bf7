// Mixed addition on y^2 = x^3 + b in XYZZ coordinates over the reduced-radix field (field_rr.hip.hpp): the hot
// loop of the bucket accumulation.  Same formulas as ec.hip.hpp (EFD madd-2008-s / mdbl-2008-s-1), same
// completeness (infinity, P + P, P - P), different arithmetic: no carry chains, no conditional
// subtractions; every intermediate's limb and value bounds are carried in its type and checked at
// compile time (Frr<Q, F, V>: limbs < F 2^B, value < V m).
//
// Cost in v_mad_u64_u32 (NL = 14): 6 products x 392 + 2 squarings x 301 + 1 fused sum of two products x 588
// = 3542 per mixed add, against 2844 multiply-add PAIRS (v_mad_u64_u32 + v_addc_co_u32) in ec.hip.hpp.
// NL = 9 (BN254): 6 x 162 + 2 x 126 + 243 + 9 (the quotient reduction of X3) = 1476, against 1280 pairs on 8 x 32 bits.
#pragma once
#include "field_rr.hip.hpp"
#include "ec.hip.hpp"
#include <type_traits>

namespace blz {

template <class F>
constexpr bool USE_RR = !std::is_void_v<typename F::RR>;  // the field has a reduced-radix twin (curve_constants.h)

template <class Q>
struct AffineRR {
    Frr<Q, 1, 2> x, y;  // Montgomery (Rrr), normalised, < 2m
};

// Two budgets.  LOOSE (the BLS base fields, 14 x 28 bits: 4 spare bits per limb, Rrr / m >= 2^11, column sums that
// admit sum(Fa Fb) <= 17): differences feed products as they are, and X3 stays the lazy difference it is (< 14 m).
// TIGHT (BN254's base field, 9 x 29 bits: 3 spare bits, Rrr / m = 2^7, sum(Fa Fb) <= 6): a difference is
// carry-propagated before it enters a product (rr_tn) and X3 goes through the one-digit quotient reduction
// (rr_xfix -> rr_reduce2m: 9 multiply-adds), so that every accumulator coordinate is < 2m.  One body serves both: the
// helpers below are the identity in the loose budget, and every bound is re-derived by the compiler from the types.
template <class Q>
constexpr bool RR_TIGHT = rr_head<Q>() < 10;
template <class Q>
constexpr int RR_JX = RR_TIGHT<Q> ? 2 : 5;  // 2^JX m dominates an accumulator's x (value < 2m / < 16m)
template <class Q>
constexpr int RR_JY = RR_TIGHT<Q> ? 2 : 3;  // ... its y (< 2m / < 4m)
template <class Q, int F, int V>
BLZ_DEV auto rr_tn(const Frr<Q, F, V>& a) {
    if constexpr (RR_TIGHT<Q> && F > 1) return rr_norm(a);
    else return a;
}
template <class Q, int F, int V>
BLZ_DEV auto rr_xfix(const Frr<Q, F, V>& a) {
    if constexpr (RR_TIGHT<Q>) return rr_reduce2m(a);
    else return rr_norm(a);
}

template <class Q, int F, int V>
BLZ_DEV auto rr_xfix_if_tight(const Frr<Q, F, V>& a) {
    if constexpr (RR_TIGHT<Q>) return rr_reduce2m(a);
    else return a;
}

// accumulator.  Loose: X3 = R^2 - PPP - 2Q is kept as the lazy difference it is (normalised limbs, value < 14 m)
template <class Q>
struct XYZZRR {
    static constexpr int VX = RR_TIGHT<Q> ? 2 : 16, VY = RR_TIGHT<Q> ? 2 : 4;
    Frr<Q, 1, VX> x;
    Frr<Q, 1, VY> y;
    Frr<Q, 1, 2> zz, zzz;
};

template <class Q>
BLZ_DEV void ptrr_set_inf(XYZZRR<Q>& p) {
    rr_zero(p.x);
    rr_zero(p.y);
    rr_zero(p.zz);
    rr_zero(p.zzz);
}
template <class Q>
BLZ_DEV bool ptrr_is_inf(const XYZZRR<Q>& p) { return rr_all_zero(p.zz); }

// Bounds in the comments: (limb factor, value factor), loose budget first, tight budget after the bar.

// 2 (x, y) for an affine point whose y may be the lazy negation 4m - y.  By value and out of line: the rare
// P + P branch must not force the hot loop's accumulator into scratch (see ec.hip.hpp's pt_mdbl_val).
// TAG: one copy per calling kernel (a shared out-of-line callee is compiled for its most permissive caller and
// then sets the register count of every kernel that reaches it).
template <class Q, int TAG = 0>
__device__ __noinline__ XYZZRR<Q> ptrr_mdbl_val(Frr<Q, 1, 2> x, Frr<Q, 2, 4> y) {
    XYZZRR<Q> r;
    // (the products are fenced from each other: this is the rare P + P branch, and left alone the scheduler overlaps
    // the four independent ones - on BLS12-377 that made this callee, and with it k_accumulate, 212 VGPRs wide, 4 more
    // than what lets the next task's sort fit beside the accumulation: msm.hip run())
    const auto U = rr_tn(rr_add(y, y));                // (4, 8) | (1, 8)
    Frr<Q, 1, 2> V, W, S, t, Msq, y3;
    rr_sqr(V, U);
    __builtin_amdgcn_sched_barrier(0);
    rr_mul(W, U, V);
    __builtin_amdgcn_sched_barrier(0);
    rr_mul(S, x, V);
    __builtin_amdgcn_sched_barrier(0);
    rr_sqr(t, x);
    __builtin_amdgcn_sched_barrier(0);
    const auto M = rr_tn(rr_add(rr_add(t, t), t));     // (3, 6) | (1, 6)
    rr_sqr(Msq, M);
    __builtin_amdgcn_sched_barrier(0);
    const auto X3 = rr_xfix(rr_sub_twice<2>(Msq, S));  // M^2 - 2S + 8m: (1, 10) | (1, 2)
    const auto D = rr_tn(rr_sub<RR_JX<Q>>(S, X3));     // S - X3 + 32m: (3, 34) | + 4m: (1, 6)
    const auto nW = rr_neg<2>(W);                      // 4m - W: (2, 4)
    rr_mul2(y3, M, D, nW, y);                          // M (S - X3) - W y
    r.x = rr_as<1, XYZZRR<Q>::VX>(X3);
    r.y = rr_as<1, XYZZRR<Q>::VY>(y3);
    r.zz = V;
    r.zzz = W;
    return r;
}

// acc += +-(x2, y2)   (affine operand, never infinity; `neg` subtracts it)
template <class Q, int TAG = 0>
BLZ_DEV void ptrr_madd(XYZZRR<Q>& acc, const AffineRR<Q>& q, bool neg) {
    const auto y2 = rr_cneg<2>(q.y, neg);              // (2, 4)
    if (ptrr_is_inf(acc)) {
        acc.x = rr_as<1, XYZZRR<Q>::VX>(q.x);
        if constexpr (RR_TIGHT<Q>) acc.y = rr_reduce2m(y2);
        else acc.y = rr_norm(y2);
        rr_one(acc.zz);
        rr_one(acc.zzz);
        return;
    }
    Frr<Q, 1, 2> U2, S2, PP, PPP, Qv, t, y3;
    rr_mul_pair(U2, q.x, acc.zz, S2, y2, acc.zzz);
    const auto P0 = rr_sub<RR_JX<Q>>(U2, acc.x);       // U2 - X1 + 32m: (3, 34) | + 4m: (3, 6)
    const auto R0 = rr_sub<RR_JY<Q>>(S2, acc.y);       // S2 - Y1 + 8m: (3, 10) | + 4m: (3, 6)
    if (__builtin_expect(rr_maybe_equal(U2, acc.x), 0)) {
        if (rr_is_zero(P0)) {
            if (rr_is_zero(R0)) acc = ptrr_mdbl_val<Q, TAG>(q.x, y2);
            else ptrr_set_inf(acc);
            return;
        }
    }
    const auto P = rr_tn(P0);                          // | (1, 6)
    const auto R = rr_tn(R0);                          // | (1, 6)
    // independent products go in pairs (field_rr.hip.hpp rr_mul_pair: two column chains per wave), ordered so that every
    // input coordinate dies as early as possible (see ec.hip.hpp's pt_madd)
    rr_sqr_pair(PP, P, t, R);
    rr_mul_pair(acc.zz, acc.zz, PP, PPP, P, PP);       // ZZ3, PPP
    rr_mul_pair(acc.zzz, acc.zzz, PPP, Qv, acc.x, PP); // ZZZ3, Q
    const auto X3 = rr_xfix(rr_sub_twice<2>(rr_sub<2>(t, PPP), Qv));  // R^2 - PPP - 2Q + 12m: (1, 14) | (1, 2)
    const auto D = rr_tn(rr_sub<RR_JX<Q>>(Qv, X3));    // Q - X3 + 32m: (3, 34) | + 4m: (1, 6)
    const auto nY = rr_neg<RR_JY<Q>>(acc.y);           // 8m - Y1: (2, 8) | 4m - Y1: (2, 4)
    rr_mul2(y3, R, D, nY, PPP);                        // R (Q - X3) - Y1 PPP, one reduction
    acc.x = rr_as<1, XYZZRR<Q>::VX>(X3);
    acc.y = rr_as<1, XYZZRR<Q>::VY>(y3);
}

// +-(x1, y1) + +-(x2, y2), both affine: the first addition of a run.  With ZZ1 = ZZZ1 = 1 the mixed add loses its four
// products by the accumulator's Z powers (U2, S2, ZZ3, ZZZ3): 1974 multiply-adds instead of 3542 (14 x 28 bits).
template <class Q, int TAG = 0>
BLZ_DEV void ptrr_aadd(XYZZRR<Q>& acc, const AffineRR<Q>& p, bool negp, const AffineRR<Q>& q, bool negq) {
    const auto y1 = rr_norm(rr_cneg<2>(p.y, negp));    // (1, 4)
    const auto y2 = rr_cneg<2>(q.y, negq);             // (2, 4)
    const auto P0 = rr_sub<2>(q.x, p.x);               // x2 - x1 + 4m: (3, 6)
    const auto R0 = rr_sub<3>(y2, y1);                 // y2 - y1 + 8m: (4, 12)
    if (__builtin_expect(rr_maybe_equal(q.x, p.x), 0)) {
        if (rr_is_zero(P0)) {
            if (rr_is_zero(R0)) acc = ptrr_mdbl_val<Q, TAG>(q.x, y2);
            else ptrr_set_inf(acc);
            return;
        }
    }
    const auto P = rr_tn(P0);                          // | (1, 6)
    const auto R = rr_xfix_if_tight(R0);               // | (1, 2): 12^2 would leave the tight value range
    Frr<Q, 1, 2> PP, PPP, Qv, t, y3;
    rr_sqr(PP, P);
    rr_mul(PPP, P, PP);
    rr_mul(Qv, p.x, PP);
    rr_sqr(t, R);
    const auto X3 = rr_xfix(rr_sub_twice<2>(rr_sub<2>(t, PPP), Qv));  // (1, 14) | (1, 2)
    const auto D = rr_tn(rr_sub<RR_JX<Q>>(Qv, X3));    // (3, 34) | (1, 6)
    const auto nY = rr_neg<3>(y1);                     // (2, 8)
    rr_mul2(y3, R, D, nY, PPP);                        // R (Q - X3) - y1 PPP
    acc.x = rr_as<1, XYZZRR<Q>::VX>(X3);
    acc.y = rr_as<1, XYZZRR<Q>::VY>(y3);
    acc.zz = PP;
    acc.zzz = PPP;
}

// Jacobian coordinates (x = X / Z^2, y = Y / Z^3) for chains of doublings only - the table check's 32 in a row
// (msm_impl.hip.hpp k_check_precompute).  On a = 0 a doubling is 3 squarings + 2 products + one fused sum of two products
// (dbl-2009-l with D = 4 X Y^2 as a product and 8 Y^4 = (8 Y^2) Y^2 folded into Y3's reduction): 2275 multiply-adds on 14 x 28
// bits against the XYZZ doubling's 3059 (963 against 1278 on 9 x 29) - XYZZ earns its two Z powers only in mixed ADDITIONS.
template <class Q>
struct JacRR {
    static constexpr int VX = RR_TIGHT<Q> ? 2 : 34;   // loose: X3 = E^2 - 8S stays the lazy difference it is (+ 32m)
    static constexpr int JX = RR_TIGHT<Q> ? 2 : 7;    // 2^JX m dominates it
    Frr<Q, 1, VX> x;
    Frr<Q, 1, 2> y;
    Frr<Q, 2, 4> z;                                   // 2 (Y Z): the doubled product as it is
};
// p = 2 p.  No infinity branch: a point of order two (Y = 0) gives Z3 = 0, and Z stays 0 mod m from there on.
// Bounds (limb factor, value factor), loose budget first, tight budget after the bar; in the tight budget (sum of limb-factor
// products <= 6, values <= 128) 4S and X3 go through the one-digit quotient reduction, the rest are carry propagations.
template <class Q>
BLZ_DEV void ptrr_jdbl(JacRR<Q>& p) {
    Frr<Q, 1, 2> A, B, S, E2, y3, yz;
    rr_sqr_pair(A, p.x, B, p.y);
    rr_mul_pair(S, p.x, B, yz, p.y, p.z);              // S = X Y^2, Y Z
    const auto E = rr_tn(rr_add(rr_add(A, A), A));     // 3 X^2: (3, 6) | (1, 6)
    rr_sqr(E2, E);
    const auto s2 = rr_add(S, S);
    const auto S4 = rr_xfix(rr_add(s2, s2));           // 4S: (1, 8) | (1, 2)
    const auto X3 = rr_xfix(rr_sub_twice<RR_TIGHT<Q> ? 2 : 4>(E2, S4));   // E^2 - 8S + 32m: (1, 34) | + 8m: (1, 2)
    const auto D = rr_tn(rr_sub<JacRR<Q>::JX>(S4, X3));   // 4S - X3 + 128m: (3, 136) | + 4m: (1, 6)
    const auto b2 = rr_add(B, B);                      // (2, 4)
    const auto nB4 = rr_neg<4>(rr_norm(rr_add(b2, b2)));   // 16m - 4 Y^2: (2, 16)
    // E (4S - X3) - 8 Y^4 = E D + (16m - 4 Y^2)(2 Y^2), one reduction: columns 9 + 4, values 816 + 64 | 1 + 4, 36 + 64
    rr_mul2(y3, E, D, nB4, b2);
    p.x = X3;
    p.y = y3;
    p.z = rr_add(yz, yz);
}

// 2 p for an accumulator (dbl-2008-s-1).  By value and out of line, like ptrr_mdbl_val.
template <class Q, int TAG = 0>
__device__ __noinline__ XYZZRR<Q> ptrr_dbl_val(XYZZRR<Q> p) {
    XYZZRR<Q> r;
    if (ptrr_is_inf(p)) { ptrr_set_inf(r); return r; }
    const auto U = rr_tn(rr_add(p.y, p.y));            // (2, 8) | (1, 4)
    Frr<Q, 1, 2> V, W, S, t, Msq, y3;
    rr_sqr(V, U);
    rr_mul(W, U, V);
    rr_mul(S, p.x, V);
    rr_sqr(t, p.x);
    const auto M = rr_tn(rr_add(rr_add(t, t), t));     // (3, 6) | (1, 6)
    rr_sqr(Msq, M);
    const auto X3 = rr_xfix(rr_sub_twice<2>(Msq, S));  // (1, 10) | (1, 2)
    const auto D = rr_tn(rr_sub<RR_JX<Q>>(S, X3));     // (3, 34) | (1, 6)
    const auto nW = rr_neg<2>(W);                      // (2, 4)
    rr_mul2(y3, M, D, nW, p.y);                        // M (S - X3) - W Y1
    r.x = rr_as<1, XYZZRR<Q>::VX>(X3);
    r.y = rr_as<1, XYZZRR<Q>::VY>(y3);
    rr_mul(r.zz, V, p.zz);
    rr_mul(r.zzz, W, p.zzz);
    return r;
}

// acc += q   (both accumulators; add-2008-s), inlined for the throughput-bound bucket reduce
template <class Q, int TAG = 0>
BLZ_DEV void ptrr_add(XYZZRR<Q>& acc, const XYZZRR<Q>& q) {
    if (ptrr_is_inf(q)) return;
    if (ptrr_is_inf(acc)) { acc = q; return; }
    Frr<Q, 1, 2> U1, U2, S1, S2, PP, PPP, Qv, t, y3, zt;
    rr_mul(U1, acc.x, q.zz);
    rr_mul(U2, q.x, acc.zz);
    rr_mul(S1, acc.y, q.zzz);
    rr_mul(S2, q.y, acc.zzz);
    const auto P0 = rr_sub<2>(U2, U1);                 // (3, 6)
    const auto R0 = rr_sub<2>(S2, S1);                 // (3, 6)
    if (__builtin_expect(rr_maybe_equal(U2, U1), 0)) {
        if (rr_is_zero(P0)) {
            if (rr_is_zero(R0)) acc = ptrr_dbl_val<Q, TAG>(q);
            else ptrr_set_inf(acc);
            return;
        }
    }
    const auto P = rr_tn(P0);                          // | (1, 6)
    const auto R = rr_tn(R0);                          // | (1, 6)
    rr_sqr(PP, P);
    rr_mul(PPP, P, PP);
    rr_mul(Qv, U1, PP);
    rr_sqr(t, R);
    const auto X3 = rr_xfix(rr_sub_twice<2>(rr_sub<2>(t, PPP), Qv));  // (1, 14) | (1, 2)
    const auto D = rr_tn(rr_sub<RR_JX<Q>>(Qv, X3));    // (3, 34) | (1, 6)
    const auto nS1 = rr_neg<2>(S1);                    // (2, 4)
    rr_mul2(y3, R, D, nS1, PPP);                       // R (Q - X3) - S1 PPP
    acc.x = rr_as<1, XYZZRR<Q>::VX>(X3);
    acc.y = rr_as<1, XYZZRR<Q>::VY>(y3);
    rr_mul(zt, acc.zz, q.zz);
    rr_mul(acc.zz, zt, PP);
    rr_mul(zt, acc.zzz, q.zzz);
    rr_mul(acc.zzz, zt, PPP);
}

// memory image of an accumulator: 4 coordinates (x | y | zz | zzz), rr_stride<Q>() dwords apart (NL rounded up to even,
// so every coordinate is 8-byte aligned: 14 dwords for 14 limbs, 10 for 9)
template <class Q>
constexpr int ptrr_dwords() { return 4 * rr_stride<Q>(); }
template <class Q>
BLZ_DEV void ptrr_load(XYZZRR<Q>& a, const uint32_t* base, size_t idx) {
    constexpr int S = rr_stride<Q>();
    const uint32_t* q = base + idx * 4 * S;
    rr_load(a.x, q);
    rr_load(a.y, q + S);
    rr_load(a.zz, q + 2 * S);
    rr_load(a.zzz, q + 3 * S);
}
template <class Q>
BLZ_DEV void ptrr_store(uint32_t* base, size_t idx, const XYZZRR<Q>& a) {
    constexpr int S = rr_stride<Q>();
    uint32_t* q = base + idx * 4 * S;
    rr_store(q, a.x);
    rr_store(q + S, a.y);
    rr_store(q + 2 * S, a.zz);
    rr_store(q + 3 * S, a.zzz);
}

// accumulator -> ec.hip.hpp's XYZZ over the 32-bit twin field (Montgomery R32, lazy [0, 2m)); infinity stays
// literal zero
template <class F>
BLZ_DEV void ptrr_to_xyzz32(XYZZ<F>& r, const XYZZRR<typename F::RR>& a) {
    using Q = typename F::RR;
    rr_to_mont32_words<Q>(r.x.v, a.x);
    rr_to_mont32_words<Q>(r.y.v, a.y);
    rr_to_mont32_words<Q>(r.zz.v, a.zz);
    rr_to_mont32_words<Q>(r.zzz.v, a.zzz);
}

// ... and back (test hooks; the pipeline never needs it)
template <class F>
BLZ_DEV void ptrr_from_xyzz32(XYZZRR<typename F::RR>& r, const XYZZ<F>& a) {
    using Q = typename F::RR;
    if (pt_is_inf(a)) { ptrr_set_inf(r); return; }
    Frr<Q, 1, 2> t;
    rr_from_mont32_words<Q>(t, a.x.v);
    r.x = rr_as<1, XYZZRR<Q>::VX>(t);
    rr_from_mont32_words<Q>(t, a.y.v);
    r.y = rr_as<1, XYZZRR<Q>::VY>(t);
    rr_from_mont32_words<Q>(r.zz, a.zz.v);
    rr_from_mont32_words<Q>(r.zzz, a.zzz.v);
}

}  // namespace blz
