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
// Montgomery multiplication, coarsely-integrated operand scanning, fully unrolled.
// ------------------------------------------------------------------------------------------
template <class P>
BLZ_DEV void fp_mul(Fp<P>& r, const Fp<P>& a, const Fp<P>& b) {
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
