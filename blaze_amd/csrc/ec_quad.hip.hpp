// Quad-cooperative XYZZ group law for the latency-bound tail of the MSM (upper bucket-reduce levels,
// Horner over the windows): the chains there are sequential and only a handful of lanes are busy, so
// one lane per operation leaves the chip idle and pays 9 (doubling) / 14 (addition) dependent field
// multiplications.  Here the 4 lanes of a DPP quad hold the SAME point (replicated) and each computes
// a different product of the formula in the same instruction stream; results are exchanged with
// quad_perm broadcasts (one v_mov_dpp per limb).  Doubling = 3 rounds, addition = 4 rounds.
// All 4 lanes of a quad must be active and hold identical inputs; outputs are identical on all 4.
#pragma once
#include "ec.hip.hpp"
#include "ec_rr.hip.hpp"

namespace blz {

// value of quad lane K on every lane of the quad
template <int K, class F>
BLZ_DEV void quad_bcast(Fp<F>& r, const Fp<F>& v) {
#pragma unroll
    for (int i = 0; i < F::N; ++i)
        r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.v[i], K * 0x55, 0xf, 0xf, true);
}
// per-lane operand choice: lane l of the quad gets a_l
template <class F>
BLZ_DEV void quad_sel(Fp<F>& r, uint32_t l, const Fp<F>& a0, const Fp<F>& a1, const Fp<F>& a2, const Fp<F>& a3) {
#pragma unroll
    for (int i = 0; i < F::N; ++i) {
        uint32_t lo = l & 1u ? a1.v[i] : a0.v[i];
        uint32_t hi = l & 1u ? a3.v[i] : a2.v[i];
        r.v[i] = l & 2u ? hi : lo;
    }
}

// p = 2p
template <class F>
__device__ __noinline__ void quad_dbl(XYZZ<F>& p, uint32_t l) {
    if (pt_is_inf(p)) return;
    Fp<F> U, a, b, r, V, A, W, S, ZZ3, MM, M, X3, D, t, WY, ZZZ3;
    fp_dbl(U, p.y);
    // round 1: V = U^2 | A = X^2
    quad_sel(a, l, U, p.x, U, p.x);
    fp_mul(r, a, a);
    quad_bcast<0>(V, r);
    quad_bcast<1>(A, r);
    fp_dbl(M, A);
    fp_add(M, M, A);
    // round 2: W = U V | S = X V | ZZ3 = V ZZ | MM = M^2
    quad_sel(a, l, U, p.x, V, M);
    quad_sel(b, l, V, V, p.zz, M);
    fp_mul(r, a, b);
    quad_bcast<0>(W, r);
    quad_bcast<1>(S, r);
    quad_bcast<2>(ZZ3, r);
    quad_bcast<3>(MM, r);
    fp_sub(X3, MM, S);
    fp_sub(X3, X3, S);
    fp_sub(D, S, X3);
    // round 3: t = M D | WY = W Y | ZZZ3 = W ZZZ
    quad_sel(a, l, M, W, W, W);
    quad_sel(b, l, D, p.y, p.zzz, p.zzz);
    fp_mul(r, a, b);
    quad_bcast<0>(t, r);
    quad_bcast<1>(WY, r);
    quad_bcast<2>(ZZZ3, r);
    p.x = X3;
    fp_sub(p.y, t, WY);
    p.zz = ZZ3;
    p.zzz = ZZZ3;
}

// acc += q
template <class F>
__device__ __noinline__ void quad_add(XYZZ<F>& acc, const XYZZ<F>& q, uint32_t l) {
    if (pt_is_inf(q)) return;
    if (pt_is_inf(acc)) { acc = q; return; }
    Fp<F> a, b, r, U1, U2, S1, S2, P, R, PP, RR, Z12, Z123, PPP, Q, ZZ3, X3, D, t, SP, ZZZ3;
    // round 1: U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1
    quad_sel(a, l, acc.x, q.x, acc.y, q.y);
    quad_sel(b, l, q.zz, acc.zz, q.zzz, acc.zzz);
    fp_mul(r, a, b);
    quad_bcast<0>(U1, r);
    quad_bcast<1>(U2, r);
    quad_bcast<2>(S1, r);
    quad_bcast<3>(S2, r);
    fp_sub(P, U2, U1);
    fp_sub(R, S2, S1);
    if (__builtin_expect(fp_maybe_zero(P), 0)) {
        if (fp_is_zero(P)) {  // same x: P + P or P - P (identical on the 4 lanes: inputs are replicated)
            if (fp_is_zero(R)) { acc = q; quad_dbl(acc, l); }
            else pt_set_inf(acc);
            return;
        }
    }
    // round 2: PP = P^2 | RR = R^2 | Z12 = ZZ1 ZZ2 | Z123 = ZZZ1 ZZZ2
    quad_sel(a, l, P, R, acc.zz, acc.zzz);
    quad_sel(b, l, P, R, q.zz, q.zzz);
    fp_mul(r, a, b);
    quad_bcast<0>(PP, r);
    quad_bcast<1>(RR, r);
    quad_bcast<2>(Z12, r);
    quad_bcast<3>(Z123, r);
    // round 3: PPP = P PP | Q = U1 PP | ZZ3 = Z12 PP
    quad_sel(a, l, P, U1, Z12, Z12);
    fp_mul(r, a, PP);
    quad_bcast<0>(PPP, r);
    quad_bcast<1>(Q, r);
    quad_bcast<2>(ZZ3, r);
    fp_sub(X3, RR, PPP);
    fp_sub(X3, X3, Q);
    fp_sub(X3, X3, Q);
    fp_sub(D, Q, X3);
    // round 4: t = R D | SP = S1 PPP | ZZZ3 = Z123 PPP
    quad_sel(a, l, R, S1, Z123, Z123);
    quad_sel(b, l, D, PPP, PPP, PPP);
    fp_mul(r, a, b);
    quad_bcast<0>(t, r);
    quad_bcast<1>(SP, r);
    quad_bcast<2>(ZZZ3, r);
    acc.x = X3;
    fp_sub(acc.y, t, SP);
    acc.zz = ZZ3;
    acc.zzz = ZZZ3;
}

// ------------------------------------------------------------------------------------------------
// The same quad-cooperative group law on the reduced-radix field (ec_rr.hip.hpp) for the curves that have one.  The tail is
// a chain of sequential point operations on ONE wave, and a single wave issues a vector instruction every four cycles at
// best: what counts is the instruction count of a field product - 576 (multiply-add / add-carry pairs) on 32-bit limbs,
// about 480 on 28-bit limbs.  k_finish is 250 doublings in sequence and dominates the latency of a small MSM (2^13
// elements: 1.93 of 4.4 ms; 1.56 of 3.9 with this).  Measured and dropped: a product with one accumulator per column
// (27 independent chains for the scheduler) is no faster - the wave is issue-bound, not latency-bound; splitting a
// product over the four lanes of a quad saves a quarter of the instructions at best (operand rotation, digit
// broadcasts and cross-lane carries eat the rest).  Same rounds as above; an operand that is chosen per
// lane takes the widest bounds of its four alternatives (rrq_sel), every product static_asserts its column and value
// bounds as everywhere in the reduced radix, and the two differences that are not followed by a product (Y3) come back
// below 2m through the one-digit quotient reduction.
// ------------------------------------------------------------------------------------------------
template <int K, class Q>
BLZ_DEV void rrq_bcast(Frr<Q, 1, 2>& r, const Frr<Q, 1, 2>& v) {
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.v[i], K * 0x55, 0xf, 0xf, true);
}
constexpr int rrq_max(int a, int b) { return a > b ? a : b; }
template <class Q, int F0, int V0, int F1, int V1, int F2, int V2, int F3, int V3>
BLZ_DEV auto rrq_sel(uint32_t l, const Frr<Q, F0, V0>& a0, const Frr<Q, F1, V1>& a1, const Frr<Q, F2, V2>& a2, const Frr<Q, F3, V3>& a3) {
    Frr<Q, rrq_max(rrq_max(F0, F1), rrq_max(F2, F3)), rrq_max(rrq_max(V0, V1), rrq_max(V2, V3))> r;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) {
        const uint32_t lo = l & 1u ? a1.v[i] : a0.v[i];
        const uint32_t hi = l & 1u ? a3.v[i] : a2.v[i];
        r.v[i] = l & 2u ? hi : lo;
    }
    return r;
}
// a difference that no product follows, back into an accumulator's range: t - u + 4m < 6m -> < 2m
template <class Q>
BLZ_DEV Frr<Q, 1, 2> rrq_diff(const Frr<Q, 1, 2>& t, const Frr<Q, 1, 2>& u) { return rr_reduce2m(rr_sub<2>(t, u)); }

// p = 2p
template <class Q>
BLZ_DEV void quadrr_dbl(XYZZRR<Q>& p, uint32_t l) {
    if (ptrr_is_inf(p)) return;
    Frr<Q, 1, 2> r, V, A, W, S, ZZ3, MM, t, WY, ZZZ3;
    const auto U = rr_tn(rr_add(p.y, p.y));
    // round 1: V = U^2 | A = X^2
    {
        const auto a = rrq_sel(l, U, p.x, U, p.x);
        rr_sqr(r, a);
    }
    rrq_bcast<0>(V, r);
    rrq_bcast<1>(A, r);
    const auto M = rr_tn(rr_add(rr_add(A, A), A));
    // round 2: W = U V | S = X V | ZZ3 = V ZZ | MM = M^2
    {
        const auto a = rrq_sel(l, U, p.x, V, M);
        const auto b = rrq_sel(l, V, V, p.zz, M);
        rr_mul(r, a, b);
    }
    rrq_bcast<0>(W, r);
    rrq_bcast<1>(S, r);
    rrq_bcast<2>(ZZ3, r);
    rrq_bcast<3>(MM, r);
    const auto X3 = rr_xfix(rr_sub_twice<2>(MM, S));
    const auto D = rr_tn(rr_sub<RR_JX<Q>>(S, X3));
    // round 3: t = M D | WY = W Y | ZZZ3 = W ZZZ
    {
        const auto a = rrq_sel(l, M, W, W, W);
        const auto b = rrq_sel(l, D, p.y, p.zzz, p.zzz);
        rr_mul(r, a, b);
    }
    rrq_bcast<0>(t, r);
    rrq_bcast<1>(WY, r);
    rrq_bcast<2>(ZZZ3, r);
    p.x = rr_as<1, XYZZRR<Q>::VX>(X3);
    p.y = rr_as<1, XYZZRR<Q>::VY>(rrq_diff(t, WY));
    p.zz = ZZ3;
    p.zzz = ZZZ3;
}

// acc += q
template <class Q>
BLZ_DEV void quadrr_add(XYZZRR<Q>& acc, const XYZZRR<Q>& q, uint32_t l) {
    if (ptrr_is_inf(q)) return;
    if (ptrr_is_inf(acc)) { acc = q; return; }
    Frr<Q, 1, 2> r, U1, U2, S1, S2, PP, RRv, Z12, Z123, PPP, Qv, ZZ3, t, SP, ZZZ3;
    // round 1: U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1
    {
        const auto a = rrq_sel(l, acc.x, q.x, acc.y, q.y);
        const auto b = rrq_sel(l, q.zz, acc.zz, q.zzz, acc.zzz);
        rr_mul(r, a, b);
    }
    rrq_bcast<0>(U1, r);
    rrq_bcast<1>(U2, r);
    rrq_bcast<2>(S1, r);
    rrq_bcast<3>(S2, r);
    const auto P0 = rr_sub<2>(U2, U1);
    const auto R0 = rr_sub<2>(S2, S1);
    if (__builtin_expect(rr_maybe_equal(U2, U1), 0)) {
        if (rr_is_zero(P0)) {  // same x: P + P or P - P (identical on the 4 lanes: inputs are replicated)
            if (rr_is_zero(R0)) { acc = q; quadrr_dbl(acc, l); }
            else ptrr_set_inf(acc);
            return;
        }
    }
    const auto P = rr_tn(P0);
    const auto R = rr_tn(R0);
    // round 2: PP = P^2 | RR = R^2 | Z12 = ZZ1 ZZ2 | Z123 = ZZZ1 ZZZ2
    {
        const auto a = rrq_sel(l, P, R, acc.zz, acc.zzz);
        const auto b = rrq_sel(l, P, R, q.zz, q.zzz);
        rr_mul(r, a, b);
    }
    rrq_bcast<0>(PP, r);
    rrq_bcast<1>(RRv, r);
    rrq_bcast<2>(Z12, r);
    rrq_bcast<3>(Z123, r);
    // round 3: PPP = P PP | Q = U1 PP | ZZ3 = Z12 PP
    {
        const auto a = rrq_sel(l, P, U1, Z12, Z12);
        rr_mul(r, a, PP);
    }
    rrq_bcast<0>(PPP, r);
    rrq_bcast<1>(Qv, r);
    rrq_bcast<2>(ZZ3, r);
    const auto X3 = rr_xfix(rr_sub_twice<2>(rr_sub<2>(RRv, PPP), Qv));
    const auto D = rr_tn(rr_sub<RR_JX<Q>>(Qv, X3));
    // round 4: t = R D | SP = S1 PPP | ZZZ3 = Z123 PPP
    {
        const auto a = rrq_sel(l, R, S1, Z123, Z123);
        const auto b = rrq_sel(l, D, PPP, PPP, PPP);
        rr_mul(r, a, b);
    }
    rrq_bcast<0>(t, r);
    rrq_bcast<1>(SP, r);
    rrq_bcast<2>(ZZZ3, r);
    acc.x = rr_as<1, XYZZRR<Q>::VX>(X3);
    acc.y = rr_as<1, XYZZRR<Q>::VY>(rrq_diff(t, SP));
    acc.zz = ZZ3;
    acc.zzz = ZZZ3;
}

}  // namespace blz
