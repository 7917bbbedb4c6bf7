// G1 group law on y^2 = x^3 + b (a = 0) in extended-Jacobian XYZZ coordinates:
//   x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2;  infinity <=> ZZ == 0 (literal zero limbs).
// Mixed add 8M+2S, full add 12M+2S, doubling 6M+3S (EFD "madd-2008-s", "add-2008-s", "dbl-2008-s-1");
// the closing Y3 = u v - w z of each is one sum-of-products with a single reduction (fp_mulsub2).
// Every routine is complete: infinity, P+P and P+(-P) are handled, because the reference harness
// repeats a 256-element tile (tests/msm/mod.rs:337-354) so equal / opposite operands meet in the
// same bucket constantly.
#pragma once
#include "field.hip.hpp"

namespace blz {

template <class F>
struct Affine {
    Fp<F> x, y;
};
template <class F>
struct XYZZ {
    Fp<F> x, y, zz, zzz;
};

template <class F>
BLZ_DEV void pt_set_inf(XYZZ<F>& p) {
    fp_zero(p.x);
    fp_zero(p.y);
    fp_zero(p.zz);
    fp_zero(p.zzz);
}
template <class F>
BLZ_DEV bool pt_is_inf(const XYZZ<F>& p) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < F::N; ++i) o |= p.zz.v[i];
    return o == 0;
}
template <class F>
BLZ_DEV void pt_from_affine(XYZZ<F>& r, const Affine<F>& a) {
    r.x = a.x;
    r.y = a.y;
    fp_one(r.zz);
    fp_one(r.zzz);
}

// cheap filter before the exact zero test: a value == 0 (mod m) in [0,2m] is 0, m or 2m
template <class F>
BLZ_DEV bool fp_maybe_zero(const Fp<F>& a) {
    return a.v[0] == 0u || a.v[0] == F::MOD[0] || a.v[0] == F::MOD2[0];
}

// r = 2*(x,y), affine input (never infinity; y != 0 on prime-order curves).
// Out of line: the rare P+P branch of the hot mixed add, and the cold reduce / finish kernels,
// share one copy per curve (keeps code size and compile time in check).
template <class F>
__device__ __noinline__ void pt_mdbl(XYZZ<F>& r, const Affine<F>& a) {
    Fp<F> U, V, W, S, M, t;
    fp_dbl(U, a.y);
    fp_sqr(V, U);
    fp_mul(W, U, V);
    fp_mul(S, a.x, V);
    fp_sqr(t, a.x);
    fp_dbl(M, t);
    fp_add(M, M, t);
    fp_sqr(r.x, M);
    fp_sub(r.x, r.x, S);
    fp_sub(r.x, r.x, S);
    fp_sub(t, S, r.x);
    fp_mulsub2(r.y, M, t, W, a.y);   // M (S - X3) - W Y1, one reduction
    r.zz = V;
    r.zzz = W;
}

template <class F>
__device__ __noinline__ XYZZ<F> pt_mdbl_val(Affine<F> a) {
    XYZZ<F> r;
    pt_mdbl(r, a);
    return r;
}

// r = 2*p
template <class F>
__device__ __noinline__ void pt_dbl(XYZZ<F>& r, const XYZZ<F>& p) {
    if (pt_is_inf(p)) { pt_set_inf(r); return; }
    Fp<F> U, V, W, S, M, t, x3;
    fp_dbl(U, p.y);
    fp_sqr(V, U);
    fp_mul(W, U, V);
    fp_mul(S, p.x, V);
    fp_sqr(t, p.x);
    fp_dbl(M, t);
    fp_add(M, M, t);
    fp_sqr(x3, M);
    fp_sub(x3, x3, S);
    fp_sub(x3, x3, S);
    fp_sub(t, S, x3);
    fp_mulsub2(U, M, t, W, p.y);     // M (S - X3) - W Y1, one reduction (r may alias p)
    r.y = U;
    r.x = x3;
    fp_mul(r.zz, V, p.zz);
    fp_mul(r.zzz, W, p.zzz);
}

// acc += (x2, y2)   (affine operand, never infinity)
template <class F>
BLZ_DEV void pt_madd(XYZZ<F>& acc, const Affine<F>& q) {
    if (pt_is_inf(acc)) { pt_from_affine(acc, q); return; }
    Fp<F> P, R, PP, PPP, Q, t;
    fp_mul(P, q.x, acc.zz);
    fp_mul(R, q.y, acc.zzz);
    fp_sub(P, P, acc.x);
    fp_sub(R, R, acc.y);
    if (__builtin_expect(fp_maybe_zero(P), 0)) {
        if (fp_is_zero(P)) {
            if (fp_is_zero(R)) {
                // out-of-line doubling, operands BY VALUE: neither `acc` nor `q` has its address
                // taken, so both stay in registers across the hot loop (a by-reference call made
                // hipcc keep the point in scratch and store it every iteration: 80 GB of HBM writes
                // per 2^26 MSM in the PMC counters)
                acc = pt_mdbl_val(q);
            } else {
                pt_set_inf(acc);
            }
            return;
        }
    }
    // ordered so that every input coordinate dies as early as possible (ZZ1, ZZZ1, X1, Y1 are
    // overwritten as soon as their last product is formed): peak liveness decides whether the hot
    // loop fits 3 waves per SIMD without scratch spills
    fp_sqr(PP, P);
    fp_mul(acc.zz, acc.zz, PP);    // ZZ3
    fp_mul(PPP, P, PP);
    fp_mul(acc.zzz, acc.zzz, PPP); // ZZZ3
    fp_mul(Q, acc.x, PP);
    fp_sqr(t, R);
    fp_sub(t, t, PPP);
    fp_sub(t, t, Q);
    fp_sub(t, t, Q);               // X3
    acc.x = t;
    fp_sub(Q, Q, t);
    fp_mulsub2(acc.y, R, Q, acc.y, PPP);   // R (Q - X3) - Y1 PPP with one reduction
}

// TAG: a kernel with its own register budget instantiates its own copy of the out-of-line doubling
// (a shared callee is compiled for the most generous of its callers, and that then becomes the
// register count of every kernel that can reach it)
template <class F, int TAG = 0>
__device__ __noinline__ XYZZ<F> pt_dbl_val(XYZZ<F> p) {
    XYZZ<F> r;
    if (pt_is_inf(p)) { pt_set_inf(r); return r; }
    Fp<F> U, V, W, S, M, t, x3;
    fp_dbl(U, p.y);
    fp_sqr(V, U);
    fp_mul(W, U, V);
    fp_mul(S, p.x, V);
    fp_sqr(t, p.x);
    fp_dbl(M, t);
    fp_add(M, M, t);
    fp_sqr(x3, M);
    fp_sub(x3, x3, S);
    fp_sub(x3, x3, S);
    fp_sub(t, S, x3);
    fp_mulsub2(r.y, M, t, W, p.y);
    r.x = x3;
    fp_mul(r.zz, V, p.zz);
    fp_mul(r.zzz, W, p.zzz);
    return r;
}

// acc += q   (both XYZZ), inlined form for throughput-bound kernels (operands stay in registers)
template <class F, int TAG = 0>
BLZ_DEV void pt_add_inl(XYZZ<F>& acc, const XYZZ<F>& q) {
    if (pt_is_inf(q)) return;
    if (pt_is_inf(acc)) { acc = q; return; }
    Fp<F> U1, S1, P, R, PP, PPP, Q, t;
    fp_mul(U1, acc.x, q.zz);
    fp_mul(P, q.x, acc.zz);
    fp_mul(S1, acc.y, q.zzz);
    fp_mul(R, q.y, acc.zzz);
    fp_sub(P, P, U1);
    fp_sub(R, R, S1);
    if (__builtin_expect(fp_maybe_zero(P), 0)) {
        if (fp_is_zero(P)) {
            if (fp_is_zero(R)) acc = pt_dbl_val<F, TAG>(q);
            else pt_set_inf(acc);
            return;
        }
    }
    fp_sqr(PP, P);
    fp_mul(PPP, P, PP);
    fp_mul(Q, U1, PP);
    fp_sqr(t, R);
    fp_sub(t, t, PPP);
    fp_sub(t, t, Q);
    fp_sub(t, t, Q);  // X3
    fp_sub(Q, Q, t);
    acc.x = t;
    fp_mulsub2(acc.y, R, Q, S1, PPP);      // R (Q - X3) - S1 PPP with one reduction
    fp_mul(t, acc.zz, q.zz);
    fp_mul(acc.zz, t, PP);
    fp_mul(t, acc.zzz, q.zzz);
    fp_mul(acc.zzz, t, PPP);
}

// acc += q   (both XYZZ)
template <class F>
__device__ __noinline__ void pt_add(XYZZ<F>& acc, const XYZZ<F>& q) {
    if (pt_is_inf(q)) return;
    if (pt_is_inf(acc)) { acc = q; return; }
    Fp<F> U1, S1, P, R, PP, PPP, Q, t;
    fp_mul(U1, acc.x, q.zz);
    fp_mul(P, q.x, acc.zz);
    fp_mul(S1, acc.y, q.zzz);
    fp_mul(R, q.y, acc.zzz);
    fp_sub(P, P, U1);
    fp_sub(R, R, S1);
    if (__builtin_expect(fp_maybe_zero(P), 0)) {
        if (fp_is_zero(P)) {
            if (fp_is_zero(R)) { XYZZ<F> d; pt_dbl(d, q); acc = d; }
            else pt_set_inf(acc);
            return;
        }
    }
    fp_sqr(PP, P);
    fp_mul(PPP, P, PP);
    fp_mul(Q, U1, PP);
    fp_sqr(t, R);
    fp_sub(t, t, PPP);
    fp_sub(t, t, Q);
    fp_sub(t, t, Q);  // X3
    fp_sub(Q, Q, t);
    acc.x = t;
    fp_mulsub2(acc.y, R, Q, S1, PPP);      // R (Q - X3) - S1 PPP with one reduction
    fp_mul(t, acc.zz, q.zz);
    fp_mul(acc.zz, t, PP);
    fp_mul(t, acc.zzz, q.zzz);
    fp_mul(acc.zzz, t, PPP);
}

// affine (Montgomery) of p; returns false for infinity
template <class F>
__device__ bool pt_to_affine(Affine<F>& a, const XYZZ<F>& p) {
    if (pt_is_inf(p)) { fp_zero(a.x); fp_zero(a.y); return false; }
    Fp<F> w, zi;
    fp_inv(w, p.zzz);       // 1/z^3
    fp_mul(zi, p.zz, w);    // 1/z
    fp_mul(a.y, p.y, w);
    fp_sqr(zi, zi);         // 1/z^2
    fp_mul(a.x, p.x, zi);
    return true;
}

}  // namespace blz
