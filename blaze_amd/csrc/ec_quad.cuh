// Quad-cooperative XYZZ group law for the latency-bound tail of the MSM (upper bucket-reduce levels,
// Horner over the windows): the chains there are sequential and only a handful of lanes are busy, so
// one lane per operation leaves the chip idle and pays 9 (doubling) / 14 (addition) dependent field
// multiplications.  Here the 4 lanes of a DPP quad hold the SAME point (replicated) and each computes
// a different product of the formula in the same instruction stream; results are exchanged with
// quad_perm broadcasts (one v_mov_dpp per limb).  Doubling = 3 rounds, addition = 4 rounds.
// All 4 lanes of a quad must be active and hold identical inputs; outputs are identical on all 4.
#pragma once
#include "ec.cuh"

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

}  // namespace blz
