// Curve-templated kernels and host launchers of the MSM pipeline (phases 1-3).  Included by one
// translation unit per curve (msm_bls377.hip, msm_bls381.hip, msm_bn254.hip) so the three
// instantiations compile in parallel.  See msm.hip for the pipeline overview.
#pragma once
#include "msm_engine.hpp"
#include "ec.hip.hpp"
#include "ec_rr.hip.hpp"
#include "ec_quad.hip.hpp"
#include "ec_row.hip.hpp"
#include <type_traits>

namespace blz {

// ------------------------------------------------------------------------------------------------
// points: wire format -> Montgomery AoS.  A 96-byte point at a 96-byte stride straddles two 128-byte
// lines half of the time, and every bucket add gathers one point: the Montgomery copy of a BLS point is
// therefore padded to one 128-byte line (BN254: 64 bytes, two per line).
// ------------------------------------------------------------------------------------------------
template <class F>
constexpr int MONT_STRIDE = 2 * F::N > 16 ? 32 : 2 * F::N;  // dwords per Montgomery point
// Fields with a reduced-radix twin (USE_RR, ec_rr.hip.hpp: the two BLS base fields) run the bucket accumulation in
// it: their Montgomery point copy holds 2 x NL limbs of B bits (x R_rr, y R_rr: 112 of the line's 128 bytes) and
// k_accumulate converts a unit's sum to the 32-bit form once, when it stores it.

// Does the reduced-radix point (2 NL limb dwords) fit the Montgomery stride?  BLS: 2 x 14 dwords in a 32-dword line.
// BN254: 2 x 9 = 18 dwords do not fit its 64-byte half line, so the copy holds x Rrr and y Rrr (each < 2m < 2^255) as
// 2 x 8 packed 32-bit words and a gather unpacks them into 29-bit limbs (18 alignbit / and pairs per point).
template <class F>
constexpr bool rr_point_packed() {
    if constexpr (USE_RR<F>) return 2 * F::RR::NL > MONT_STRIDE<F>;
    else return false;
}

// one point (canonical x, y: 32-bit words) into slot `p` of a Montgomery point array
template <class F>
BLZ_DEV void store_mont_point(uint32_t* __restrict__ mont, size_t p, Fp<F>& x, Fp<F>& y) {
    if constexpr (USE_RR<F>) {
        using Q = typename F::RR;
        Frr<Q, 1, 2> xr, yr;
        rr_to_mont_from_words<Q>(xr, x.v);
        rr_to_mont_from_words<Q>(yr, y.v);
        if constexpr (rr_point_packed<F>()) {
            static_assert(2 * Q::N32 <= MONT_STRIDE<F>, "packed reduced-radix point does not fit its stride");
            rr_to_words<Q>(x.v, xr);
            rr_to_words<Q>(y.v, yr);
            fp_store(mont + p * MONT_STRIDE<F>, x);
            fp_store(mont + p * MONT_STRIDE<F> + F::N, y);
        } else {
            static_assert(2 * Q::NL <= MONT_STRIDE<F> && Q::NL % 2 == 0, "reduced-radix point does not fit its line");
            rr_store(mont + p * MONT_STRIDE<F>, xr);
            rr_store(mont + p * MONT_STRIDE<F> + Q::NL, yr);
        }
    } else {
        fp_to_mont(x, x);
        fp_to_mont(y, y);
        fp_store(mont + p * MONT_STRIDE<F>, x);
        fp_store(mont + p * MONT_STRIDE<F> + F::N, y);
    }
}

template <class F>
__global__ __launch_bounds__(256) void k_points_to_mont(const uint32_t* __restrict__ raw, uint32_t* __restrict__ mont,
                                                        uint32_t npts) {
    uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= npts) return;
    Fp<F> x, y;
    fp_load(x, raw + (size_t)p * 2 * F::N);
    fp_load(y, raw + (size_t)p * 2 * F::N + F::N);
    store_mont_point<F>(mont, p, x, y);
}

// ------------------------------------------------------------------------------------------------
// Checked-table plan of precompute handles (opt-in: blz_msm_set_precompute_plan; arena_tables.hip arena_precompute_check).
// The reference's precompute mode (MSMInit.is_precompute, msm_api.rs:39-50) has the CALLER supply, per element, the 8 bases
// B_j = 2^(32 j) P (precompute_base_*: tests/msm/mod.rs:360-380) and the device sums s_j B_j over the 32-bit chunks s_j of the
// scalar.  Served as the 8n-point MSM it is, that is 16 bucket additions per element (two 17-bit windows per chunk) against 12
// without the table.  IF the table is what precompute_base_* produces, the same sum is sum_k (s_2k + 2^32 s_(2k+1)) B_2k:
// 4n points with 64-bit scalars - three windows of 22 / 22 / 21 bits, 12 additions per element into 3 bucket sets of 2^21 +
// 2^21 + 2^20 slots.  The library cannot take the caller's word for it (a table that is NOT consistent has a defined result
// too: the exact sum over all 8 bases), so the resident table is checked once per load, base by base:
//   k_check_precompute      one lane per (element, j in 1..7): 32 doublings of B_(j-1) in Jacobian coordinates, compared with
//                           B_j projectively (X == x_j Z^2, Y == y_j Z^3, Z != 0: no inversion); the j = 1 lane also checks that B_0 is
//                           on the curve (then every B_j is, and every evaluation order of the sum gives the same point).
//                           Any miss raises *flag and the extent stays on the exact path.
//   k_points_to_mont_even   the Montgomery copy of the even bases only (B_0, B_2, B_4, B_6 of every element, contiguous):
//                           half the copy's memory, and the 4n-point task gathers from it with plain indices.
// ------------------------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(256) void k_points_to_mont_even(const uint32_t* __restrict__ raw, uint32_t* __restrict__ mont, uint32_t nq) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= nq) return;
    const size_t p = (size_t)(q >> 2) * 8u + (size_t)(q & 3u) * 2u;
    Fp<F> x, y;
    fp_load(x, raw + p * 2 * F::N);
    fp_load(y, raw + p * 2 * F::N + F::N);
    store_mont_point<F>(mont, q, x, y);
}

template <class F>
BLZ_DEV void load_affine_rr(AffineRR<typename F::RR>& a, const uint32_t* pts, uint32_t idx);   // (below, with the accumulation's loaders)
template <class F>
BLZ_DEV void load_affine(Affine<F>& a, const uint32_t* pts, uint32_t idx);

// Arena diet (opt-in: blz_arena_set_policy, arena.hip): once an extent's Montgomery copy is complete the raw bytes are a second
// copy of the same points (BLS: 96 + 128 bytes per base) that only get_data_from_hbm, a later write, an export or a table
// build would ever read.  They can be dropped IF the copy gives them back exactly: canonical coordinates (x, y < q:
// k_points_all_canonical checks the raw bytes before they go) convert to Montgomery form and back without loss.
template <class F>
__global__ __launch_bounds__(256) void k_points_from_mont(const uint32_t* __restrict__ mont, uint32_t* __restrict__ raw, uint64_t npts) {
    const uint64_t p = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (p >= npts) return;
    Fp<F> x, y;
    if constexpr (USE_RR<F>) {
        using Q = typename F::RR;
        AffineRR<Q> a;
        load_affine_rr<F>(a, mont, (uint32_t)p);
        rr_to_mont32_words<Q>(x.v, a.x);
        rr_to_mont32_words<Q>(y.v, a.y);
    } else {
        Affine<F> a;
        load_affine<F>(a, mont, (uint32_t)p);
        x = a.x;
        y = a.y;
    }
    fp_from_mont(x, x);
    fp_from_mont(y, y);
    fp_store(raw + p * 2 * F::N, x);
    fp_store(raw + p * 2 * F::N + F::N, y);
}
template <class F>
__global__ __launch_bounds__(256) void k_points_all_canonical(const uint32_t* __restrict__ raw, uint64_t npts, uint32_t* __restrict__ flag) {
    const uint64_t p = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (p >= npts) return;
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        Fp<F> v;
        fp_load(v, raw + p * 2 * F::N + c * F::N);
        // v >= m ?  (compare from the top limb down)
        bool ge = true;
#pragma unroll
        for (int i = F::N - 1; i >= 0; --i) {
            if (v.v[i] != F::MOD[i]) { ge = v.v[i] > F::MOD[i]; break; }
        }
        bad |= ge;
    }
    if (bad) atomicOr(flag, 1u);
}

template <class F>
__global__ __launch_bounds__(64, 3) void k_check_precompute(const uint32_t* __restrict__ raw, uint64_t nelem, uint32_t* __restrict__ flag) {
    const uint64_t items = nelem * 7u;
    for (uint64_t t = (uint64_t)blockIdx.x * 64u + threadIdx.x; t < items; t += (uint64_t)gridDim.x * 64u) {
        if (*(volatile uint32_t*)flag) return;   // refuted already
        const uint64_t i = t / 7u;
        const uint32_t j = (uint32_t)(t - i * 7u) + 1u;
        const uint32_t* src = raw + (i * 8u + (j - 1u)) * 2 * F::N;
        const uint32_t* dst = src + 2 * F::N;
        Fp<F> x0, y0, x1, y1;
        fp_load(x0, src); fp_load(y0, src + F::N);
        fp_load(x1, dst); fp_load(y1, dst + F::N);
        bool ok = true;
        if (j == 1u) {
            // B_0 on the curve: y^2 == x^3 + b
            Fp<F> xm, ym, l, r, b;
            fp_to_mont(xm, x0);
            fp_to_mont(ym, y0);
            fp_sqr(l, ym);
            fp_sqr(r, xm);
            fp_mul(r, r, xm);
#pragma unroll
            for (int k = 0; k < F::N; ++k) b.v[k] = F::CURVE_B[k];
            fp_add(r, r, b);
            ok = fp_eq(l, r);
        }
        if constexpr (USE_RR<F>) {
            using Q = typename F::RR;
            Frr<Q, 1, 2> ax, ay;
            AffineRR<Q> bj;
            rr_to_mont_from_words<Q>(ax, x0.v);
            rr_to_mont_from_words<Q>(ay, y0.v);
            rr_to_mont_from_words<Q>(bj.x, x1.v);
            rr_to_mont_from_words<Q>(bj.y, y1.v);
            // a chain of doublings and nothing else: Jacobian coordinates (ec_rr.hip.hpp JacRR: 2275 multiply-adds per doubling
            // against XYZZ's 3059)
            JacRR<Q> p;
            p.x = rr_as<1, JacRR<Q>::VX>(ax);
            p.y = ay;
            rr_one(p.z);
            // (a doubling that reaches infinity - Z == 0 mod m - stays there: every later Z is a multiple of it)
#pragma unroll 1
            for (int d = 0; d < 32; ++d) ptrr_jdbl(p);
            if (rr_is_zero(p.z)) ok = false;   // 2^32 B_(j-1) is the point at infinity: no affine B_j equals it
            Frr<Q, 1, 2> ZZ, ZZZ, U2, S2;
            rr_sqr(ZZ, p.z);
            rr_mul(ZZZ, ZZ, p.z);
            rr_mul_pair(U2, bj.x, ZZ, S2, bj.y, ZZZ);
            const auto P0 = rr_sub<JacRR<Q>::JX>(U2, p.x);
            const auto R0 = rr_sub<2>(S2, p.y);
            if (!rr_is_zero(P0) || !rr_is_zero(R0)) ok = false;
        } else {
            Affine<F> a, bj;
            fp_to_mont(a.x, x0);
            fp_to_mont(a.y, y0);
            fp_to_mont(bj.x, x1);
            fp_to_mont(bj.y, y1);
            XYZZ<F> p;
            pt_mdbl(p, a);
            for (int d = 1; d < 32; ++d) { XYZZ<F> t2; pt_dbl(t2, p); p = t2; }
            if (fp_is_zero(p.zz)) ok = false;
            Fp<F> U2, S2;
            fp_mul(U2, bj.x, p.zz);
            fp_mul(S2, bj.y, p.zzz);
            if (!fp_eq(U2, p.x) || !fp_eq(S2, p.y)) ok = false;
        }
        if (!ok) {
            atomicOr(flag, 1u);
            return;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Window table of resident bases (opt-in: blz_msm_set_window_table): table[i W + j] = 2^(c j) P_i, j < W, in the
// Montgomery point format above.  With the window weights moved into the points, all W windows of a scalar add into
// ONE bucket set, so the bucket count no longer multiplies with the window count and the windows can be as wide as
// the entries-per-bucket ratio allows: 2^26 bases run 10 windows of 26 bits (671 M additions, 2^25 buckets) instead of
// 12 windows of 21 - 23 bits (805 M additions, 12 x 2^21 buckets).  Costs W x the memory of the bases and, once per
// load, this kernel: one lane per base walks the c (W - 1) doublings (32-bit XYZZ arithmetic, ec.hip.hpp), parks the
// W - 1 unnormalised multiples and the running products of their ZZZ in a lane-private scratch row, inverts the
// last product once and normalises on the way back (Montgomery's trick inside the lane).
// A base whose multiple comes out as infinity (only a point of even order can: none in the r-torsion) cannot be
// tabulated: *flag is raised and the host falls back to the plain path.
// ------------------------------------------------------------------------------------------------
#ifndef BLZ_TABLE_BUILD_JACOBIAN
#define BLZ_TABLE_BUILD_JACOBIAN 1
#endif
template <class F>
constexpr size_t TABLE_SCRATCH_ROW = 5 * F::N;   // dwords per parked multiple: X, Y, ZZ, ZZZ, running product
template <class F>
__global__ __launch_bounds__(64, 3) void k_build_window_table(const uint32_t* __restrict__ raw, uint32_t* __restrict__ table,
                                                              uint32_t npts, int c, int W, int base_shift,
                                                              uint32_t* __restrict__ scratch, uint32_t* __restrict__ flag) {
    // entry j of a base is 2^(base_shift + c j) P.  base_shift = 0: entry 0 is the base as given; a handle with a scalar range
    // [lo, hi) tabulates from base_shift = lo, so that entry 0 needs normalising like the others (one more scratch row)
    const uint32_t lane = blockIdx.x * 64u + threadIdx.x, nlanes = gridDim.x * 64u;
    uint32_t* row = scratch + (size_t)lane * (size_t)W * TABLE_SCRATCH_ROW<F>;
    const int j0 = base_shift > 0 ? 0 : 1;   // first entry that goes through the scratch rows
    for (uint32_t i = lane; i < npts; i += nlanes) {
        Fp<F> x, y;
        fp_load(x, raw + (size_t)i * 2 * F::N);
        fp_load(y, raw + (size_t)i * 2 * F::N + F::N);
        XYZZ<F> p;
        Fp<F> prod;
        bool bad = false;
        // The chain of doublings - 95 % of the kernel - runs in Jacobian coordinates on the reduced radix where the field has one
        // (ec_rr.hip.hpp ptrr_jdbl: ~3200 instructions per doubling against ~5600 for the 32-bit XYZZ doubling); every c-th
        // point leaves it as (X, Y, Z^2, Z^3) in the 32-bit form the parking and normalising code below works on.
        // (-DBLZ_TABLE_BUILD_JACOBIAN=0: the 32-bit chain everywhere, for A/B runs - profiles/r05_table_build_ab.txt)
        constexpr bool JAC = BLZ_TABLE_BUILD_JACOBIAN != 0 && USE_RR<F>;
        typename std::conditional<JAC, JacRR<typename F::RR>, int>::type jp;
        Affine<F> a;
        if constexpr (JAC) {
            using Q = typename F::RR;
            Frr<Q, 1, 2> ax;
            rr_to_mont_from_words<Q>(ax, x.v);
            rr_to_mont_from_words<Q>(jp.y, y.v);
            jp.x = rr_as<1, JacRR<Q>::VX>(ax);
            rr_one(jp.z);
        } else {
            (void)jp;
            fp_to_mont(a.x, x);
            fp_to_mont(a.y, y);
        }
        if (j0 == 1) store_mont_point<F>(table, (size_t)i * W, x, y);   // the base itself (x, y are consumed)
        for (int j = j0; j < W; ++j) {
            int nd = j == 0 ? base_shift : c;   // doublings from the previous entry (or from the base)
            if constexpr (JAC) {
                using Q = typename F::RR;
#pragma unroll 1
                for (int d = 0; d < nd; ++d) ptrr_jdbl(jp);
                if (rr_is_zero(jp.z)) { bad = true; break; }
                Frr<Q, 1, 2> zz, zzz;
                rr_sqr(zz, jp.z);
                rr_mul(zzz, zz, jp.z);
                rr_to_mont32_words<Q>(p.x.v, jp.x);
                rr_to_mont32_words<Q>(p.y.v, jp.y);
                rr_to_mont32_words<Q>(p.zz.v, zz);
                rr_to_mont32_words<Q>(p.zzz.v, zzz);
            } else {
                XYZZ<F> t;
                if (j == j0) { pt_mdbl(p, a); --nd; }
                for (int d = 0; d < nd; ++d) { pt_dbl(t, p); p = t; }
                if (pt_is_inf(p)) { bad = true; break; }
            }
            if (j == j0) prod = p.zzz;
            else fp_mul(prod, prod, p.zzz);
            uint32_t* q = row + (size_t)(j - j0) * TABLE_SCRATCH_ROW<F>;
            fp_store(q, p.x); fp_store(q + F::N, p.y); fp_store(q + 2 * F::N, p.zz); fp_store(q + 3 * F::N, p.zzz);
            fp_store(q + 4 * F::N, prod);
        }
        if (bad) {
            atomicOr(flag, 1u);
            continue;
        }
        if (W - j0 < 1) continue;   // (a table of the bases alone)
        Fp<F> inv;
        fp_inv(inv, prod);   // 1 / (ZZZ of every parked entry)
        for (int j = W - 1; j >= j0; --j) {
            const uint32_t* q = row + (size_t)(j - j0) * TABLE_SCRATCH_ROW<F>;
            Fp<F> X, Y, ZZ, ZZZ, w;
            fp_load(X, q); fp_load(Y, q + F::N); fp_load(ZZ, q + 2 * F::N); fp_load(ZZZ, q + 3 * F::N);
            if (j > j0) {
                Fp<F> before;
                fp_load(before, q - TABLE_SCRATCH_ROW<F> + 4 * F::N);   // product of the ZZZ below this entry
                fp_mul(w, inv, before);                                  // 1 / ZZZ_j
                fp_mul(inv, inv, ZZZ);
            } else {
                w = inv;
            }
            Fp<F> zi;
            fp_mul(zi, ZZ, w);     // 1 / z
            fp_mul(Y, Y, w);
            fp_sqr(zi, zi);
            fp_mul(X, X, zi);
            fp_from_mont(x, X);
            fp_from_mont(y, Y);
            store_mont_point<F>(table, (size_t)i * W + j, x, y);
        }
    }
}

template <class F>
BLZ_DEV void load_affine_rr(AffineRR<typename F::RR>& a, const uint32_t* pts, uint32_t idx) {
    using Q = typename F::RR;
    const uint4* q4 = reinterpret_cast<const uint4*>(pts + (size_t)idx * MONT_STRIDE<F>);
    if constexpr (rr_point_packed<F>()) {
        // 2 x N32 packed words (one 64-byte half line for BN254), unpacked into limbs
        uint32_t wx[Q::N32], wy[Q::N32];
        static_assert(Q::N32 % 4 == 0, "packed coordinate not a whole number of 16-byte pieces");
#pragma unroll
        for (int i = 0; i < Q::N32 / 4; ++i) {
            uint4 v = q4[i];
            wx[4 * i] = v.x; wx[4 * i + 1] = v.y; wx[4 * i + 2] = v.z; wx[4 * i + 3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < Q::N32 / 4; ++i) {
            uint4 v = q4[Q::N32 / 4 + i];
            wy[4 * i] = v.x; wy[4 * i + 1] = v.y; wy[4 * i + 2] = v.z; wy[4 * i + 3] = v.w;
        }
        rr_from_words<Q>(a.x, wx);
        rr_from_words<Q>(a.y, wy);
    } else {
        // 2 NL dwords, 16-byte aligned: x and y share a vector load in the middle when NL is not a multiple of 4
        uint32_t w[2 * Q::NL];
        static_assert((2 * Q::NL) % 4 == 0, "point not a whole number of 16-byte pieces");
#pragma unroll
        for (int i = 0; i < 2 * Q::NL / 4; ++i) {
            uint4 v = q4[i];
            w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < Q::NL; ++i) { a.x.v[i] = w[i]; a.y.v[i] = w[Q::NL + i]; }
    }
}

// dwords per unit / bucket sum in `partial`: fields with a reduced-radix twin keep the sums in it (4 x 14 limbs),
// so neither the accumulation nor the first bucket-reduce level converts anything
template <class F>
constexpr int partial_dwords() {
    if constexpr (USE_RR<F>) return ptrr_dwords<typename F::RR>();
    else return 4 * F::N;
}
// (for the few kernels that still work on 32-bit limbs: k_combine_units)
template <class F>
BLZ_DEV void load_partial32(XYZZ<F>& a, const uint32_t* partial, size_t idx) {
    if constexpr (USE_RR<F>) {
        XYZZRR<typename F::RR> r;
        ptrr_load(r, partial, idx);
        ptrr_to_xyzz32<F>(a, r);
    } else {
        const uint32_t* q = partial + idx * 4 * F::N;
        fp_load(a.x, q); fp_load(a.y, q + F::N); fp_load(a.zz, q + 2 * F::N); fp_load(a.zzz, q + 3 * F::N);
    }
}
template <class F>
BLZ_DEV void store_partial32(uint32_t* partial, size_t idx, const XYZZ<F>& a) {
    if constexpr (USE_RR<F>) {
        XYZZRR<typename F::RR> r;
        ptrr_from_xyzz32<F>(r, a);
        ptrr_store(partial, idx, r);
    } else {
        uint32_t* q = partial + idx * 4 * F::N;
        fp_store(q, a.x); fp_store(q + F::N, a.y); fp_store(q + 2 * F::N, a.zz); fp_store(q + 3 * F::N, a.zzz);
    }
}

template <class F>
BLZ_DEV void load_affine(Affine<F>& a, const uint32_t* pts, uint32_t idx) {
    const uint32_t* q = pts + (size_t)idx * MONT_STRIDE<F>;
    fp_load(a.x, q);
    fp_load(a.y, q + F::N);
}
template <class F>
BLZ_DEV void load_xyzz(XYZZ<F>& a, const uint32_t* base, size_t idx) {
    const uint32_t* q = base + idx * 4 * F::N;
    fp_load(a.x, q);
    fp_load(a.y, q + F::N);
    fp_load(a.zz, q + 2 * F::N);
    fp_load(a.zzz, q + 3 * F::N);
}
template <class F>
BLZ_DEV void store_xyzz(uint32_t* base, size_t idx, const XYZZ<F>& a) {
    uint32_t* q = base + idx * 4 * F::N;
    fp_store(q, a.x);
    fp_store(q + F::N, a.y);
    fp_store(q + 2 * F::N, a.zz);
    fp_store(q + 3 * F::N, a.zzz);
}

// NOTE on __launch_bounds__(T, 3) below: the out-of-line group-law routines (pt_mdbl, pt_add, pt_dbl,
// quad_*) are compiled once per translation unit with the register budget of their MOST permissive
// caller, and a kernel's VGPR count is the maximum over its callees.  One kernel left at the default
// (1 wave per SIMD, 512 VGPRs) let pt_mdbl grow to 178 VGPRs and silently dropped k_accumulate from
// 3 to 2 waves per SIMD (-7 %).  Every kernel of this file that reaches those routines therefore
// declares >= 3 waves per SIMD.
// ------------------------------------------------------------------------------------------------
// phase 1: bucket accumulation.  One lane per unit (a run of <= L entries of one bucket).
// ------------------------------------------------------------------------------------------------
#ifndef BLZ_ACC_RR_WAVES
#define BLZ_ACC_RR_WAVES 2
#endif
// Register cap of k_accumulate: the next task's digit sort only runs BESIDE two of its waves per SIMD up to 200 VGPRs
// (msm.hip run()).  BLS12-381 compiles to 194 on its own; BLS12-377 to 206 (same code, other constants), which left its
// sorts in the open (+4 ms per MSM) until this cap: 200 registers and 8 more spilled dwords outside the inner loop.
// The attribute counts in units of the pre-gfx90a file (the backend doubles it for the unified 512-register file): 100 = 200.
// waves per SIMD of the 32-bit-limb kernel (BN254 precompute handles): 2 like the reduced-radix kernels - measured equal to 3
// (config 3: 82.9 against 82.8 ms) - so that the register cap applies and the next task's sort fits beside it
#ifndef BLZ_ACC_W32_WAVES
#define BLZ_ACC_W32_WAVES 2
#endif
#ifndef BLZ_ACC_VGPR_CAP
#define BLZ_ACC_VGPR_CAP 100
#endif
// point index of an entry.  -DBLZ_GATHER_MASK=0x... (A/B builds only: WRONG RESULTS, timing experiments) confines every
// gather to the first mask + 1 points of the table, which separates the kernel's arithmetic from its memory side.
#ifdef BLZ_GATHER_MASK
#define BLZ_PT_IDX(e) ((e) & 0x7fffffffu & (uint32_t)(BLZ_GATHER_MASK))
#else
#define BLZ_PT_IDX(e) ((e) & 0x7fffffffu)
#endif
// CONT (piecewise tasks, msm.hip begin() / accumulate_slice()): the task's elements arrive in pieces, every piece is sorted
// and accumulated on its own and the bucket sums live in `sums`, indexed by BUCKET, across the pieces.  A unit that is its
// bucket's only unit of this piece - every unit, bar the runs longer than L - takes the bucket's sum so far as its
// starting value and puts the new sum back (`first`: the task's first piece - nothing to load, the run's first addition is
// the cheap affine + affine one); the units of a bucket that needed several go through `partial` and the unit folds as
// ever, and k_merge_buckets adds their total to sums[g].  Against merging every piece's sums bucket by bucket this saves a
// full addition per occupied bucket and piece and costs the first addition of a run its affine + affine shortcut.
template <class F, bool CONT>
BLZ_DEV void accumulate_body(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ entries,
                             const uint32_t* __restrict__ off, const uint32_t* __restrict__ unit_off,
                             const uint32_t* __restrict__ unit_bucket, const uint32_t* __restrict__ unit_order,
                             const uint32_t* __restrict__ stats, uint32_t L, uint32_t* __restrict__ partial,
                             uint32_t* __restrict__ sums, bool first) {
    // the grid covers the host's upper bound of the unit count; the real count is on the device (stats[0])
    uint32_t t = blockIdx.x * 128u + threadIdx.x;
    if (t >= stats[0]) return;
    const uint32_t u = unit_order[t];  // units of equal run length sit in the same wave
    uint32_t g = unit_bucket[u];
    const uint32_t u0 = unit_off[g];
    uint32_t k = u - u0;
    uint32_t start = off[g] + k * L;
    uint32_t end = off[g + 1];
    if (end - start > L) end = start + L;
    uint32_t* dst = partial;
    size_t didx = u;
    bool resume = false;
    if constexpr (CONT) {
        if (unit_off[g + 1] - u0 == 1) {
            dst = sums;
            didx = g;
            resume = !first;
        }
    }
    if constexpr (USE_RR<F>) {
        // reduced-radix arithmetic (ec_rr.hip.hpp): 2 waves per SIMD reach 95 % of the multiplier's rate
        // (profiles/r02_mul_variants.txt), which leaves 256 VGPRs: room for the next point's prefetch
        using Q = typename F::RR;
        XYZZRR<Q> acc;
        ptrr_set_inf(acc);
        if constexpr (CONT) {
            if (resume) ptrr_load(acc, sums, g);
        }
        uint32_t e = entries[start];
#if BLZ_ACC_RR_WAVES == 2
        // Two loads run ahead of the arithmetic: the POINT of entry j + 1 (28 / 16 VGPRs) and the INDEX of entry j + 2.
        // With the index only one step ahead, every iteration stalled once for a dependent pair of loads (index, then
        // the point it names) before its addition could start; two steps ahead the point's address is already in a
        // register when its load is issued.
        uint32_t en = start + 1 < end ? entries[start + 1] : 0u;   // entry after `e`
        AffineRR<Q> nxt;
        load_affine_rr<F>(nxt, pts, BLZ_PT_IDX(e));
        uint32_t j = start;
#ifndef BLZ_ACC_NO_AADD
        if ((!CONT || first) && end - start >= 2) {
            // the run's first two points are both affine: a cheaper addition than the mixed one (ptrr_aadd);
            // units are ordered by length, so the lanes of a wave take this branch together
            const AffineRR<Q> p0 = nxt;
            const bool neg0 = (e & 0x80000000u) != 0;
            const uint32_t e1 = en;
            AffineRR<Q> p1;
            load_affine_rr<F>(p1, pts, BLZ_PT_IDX(e1));
            if (start + 2 < end) {
                e = entries[start + 2];
                load_affine_rr<F>(nxt, pts, BLZ_PT_IDX(e));
                en = start + 3 < end ? entries[start + 3] : 0u;
            }
            ptrr_aadd<Q, 1>(acc, p0, neg0, p1, (e1 & 0x80000000u) != 0);
            j = start + 2;
        }
#endif
        for (; j < end; ++j) {
            const AffineRR<Q> cur = nxt;
            const bool neg = (e & 0x80000000u) != 0;
            if (j + 1 < end) {
                e = en;
                load_affine_rr<F>(nxt, pts, BLZ_PT_IDX(e));
                if (j + 2 < end) en = entries[j + 2];
            }
            ptrr_madd<Q, 1>(acc, cur, neg);
        }
#else
        for (uint32_t j = start; j < end; ++j) {
            AffineRR<Q> cur;
            load_affine_rr<F>(cur, pts, BLZ_PT_IDX(e));
            const bool neg = (e & 0x80000000u) != 0;
            if (j + 1 < end) e = entries[j + 1];
            ptrr_madd<Q, 1>(acc, cur, neg);
        }
#endif
        ptrr_store(dst, didx, acc);
    } else {
        // The 32-bit-limb kernel needs only ~125 VGPRs and would run four waves per SIMD - no registers left for the hidden
        // sort's waves (4 x 128 + 72 > 512).  Naming v135 as clobbered makes the allocation 136: three waves, 3 x 136 + 72 = 480.
        // (amdgpu_waves_per_eu's maximum is a hint to the register allocator; it does not pad the allocation.)
        asm volatile("" ::: "v135");
        XYZZ<F> acc;
        pt_set_inf(acc);
        if constexpr (CONT) {
            if (resume) load_xyzz(acc, sums, g);
        }
        // only the next entry INDEX is prefetched: holding the next point as well costs 24 VGPRs, which at
        // 3 waves per SIMD (168 VGPRs) turned into scratch spills (100 GB of HBM writes per 2^26 MSM in
        // the WRITE_SIZE counter); the other two waves of the SIMD cover the gather latency instead
        uint32_t e = entries[start];
        for (uint32_t j = start; j < end; ++j) {
            Affine<F> cur;
            load_affine(cur, pts, BLZ_PT_IDX(e));
            const uint32_t ecur = e;
            if (j + 1 < end) e = entries[j + 1];
            if (ecur & 0x80000000u) fp_neg(cur.y, cur.y);
            pt_madd(acc, cur);
        }
        store_xyzz(dst, didx, acc);
    }
}

template <class F>
__global__ __launch_bounds__(128, USE_RR<F> ? BLZ_ACC_RR_WAVES : BLZ_ACC_W32_WAVES) __attribute__((amdgpu_num_vgpr(BLZ_ACC_VGPR_CAP))) void k_accumulate(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ entries,
                                                    const uint32_t* __restrict__ off, const uint32_t* __restrict__ unit_off,
                                                    const uint32_t* __restrict__ unit_bucket,
                                                    const uint32_t* __restrict__ unit_order,
                                                    const uint32_t* __restrict__ stats, uint32_t L,
                                                    uint32_t* __restrict__ partial) {
    accumulate_body<F, false>(pts, entries, off, unit_off, unit_bucket, unit_order, stats, L, partial, nullptr, false);
}
// the piecewise twin (same register cap: the next piece's sort hides underneath it)
template <class F>
__global__ __launch_bounds__(128, USE_RR<F> ? BLZ_ACC_RR_WAVES : BLZ_ACC_W32_WAVES) __attribute__((amdgpu_num_vgpr(BLZ_ACC_VGPR_CAP))) void k_accumulate_cont(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ entries,
                                                    const uint32_t* __restrict__ off, const uint32_t* __restrict__ unit_off,
                                                    const uint32_t* __restrict__ unit_bucket,
                                                    const uint32_t* __restrict__ unit_order,
                                                    const uint32_t* __restrict__ stats, uint32_t L,
                                                    uint32_t* __restrict__ partial, uint32_t* __restrict__ sums, uint32_t first) {
    accumulate_body<F, true>(pts, entries, off, unit_off, unit_bucket, unit_order, stats, L, partial, sums, first != 0);
}

// buckets that needed several units: fold partial[u0 + k*stride] for k in the same 16-group.
// Every unit that leads such a group is a full-length unit, and those are the first hist[L]
// entries of the length-ordered unit list, so only that prefix is visited.
template <class F>
__global__ __launch_bounds__(128, 3) void k_combine_units(const uint32_t* __restrict__ unit_off,
                                                       const uint32_t* __restrict__ unit_bucket,
                                                       const uint32_t* __restrict__ unit_order,
                                                       const uint32_t* __restrict__ nfull_ptr,
                                                       const uint32_t* __restrict__ stats, uint32_t L,
                                                       uint32_t stride, uint32_t thr, uint32_t hot_start, uint32_t* __restrict__ partial) {
    // the host launches one pass per power of 16 up to the LARGEST possible bucket; the passes beyond this
    // task's longest bucket (stats[1] entries) have nothing to fold.  thr > 0: buckets of up to thr units are
    // k_combine_buckets' (below); this tree only folds the hot ones - none at all in most tasks
    const uint32_t max_units = (stats[1] + L - 1) / L;
    if (stride >= max_units || max_units <= thr) return;
    // One DPP quad per 16 consecutive entries of the full-unit list (ec_quad.hip.hpp): it scans them for
    // group leaders (at most two: a bucket's full units are contiguous in the list) and folds each
    // leader's group.  The chain of up to 15 additions is sequential and only hot buckets have any, so
    // latency is what counts: 4 dependent product rounds per add instead of 14, and every lane of a wave
    // busy (one lane per list entry left 1 quad in 16 working: 1.4 ms for 16 K groups).
    const uint32_t gt = blockIdx.x * 128u + threadIdx.x;
    const uint32_t q = gt >> 2, ql = gt & 3u;
    const uint32_t nfull = *nfull_ptr;
    // scan first, fold afterwards: the quads of a wave find their leaders at different positions, and a
    // chain started inside the scan loop would run once per distinct position (measured: 4x the work)
    uint32_t leaders = 0;
    for (uint32_t i = 0; i < 16; ++i) {
        const uint32_t t0 = q * 16u + i;
        if (t0 >= nfull) break;
        uint32_t u = unit_order[t0];
        uint32_t g = unit_bucket[u];
        uint32_t u0 = unit_off[g], u1 = unit_off[g + 1];
        if (g < hot_start && u1 - u0 > stride && u1 - u0 > thr && (u - u0) % (16u * stride) == 0) leaders |= 1u << i;   // (hot_start..: k_fold_hot's)
    }
    while (leaders) {
        const uint32_t i = (uint32_t)__builtin_ctz(leaders);
        leaders &= leaders - 1u;
        const uint32_t u = unit_order[q * 16u + i];
        const uint32_t u1 = unit_off[unit_bucket[u] + 1];
        XYZZ<F> acc;
        load_partial32(acc, partial, u);
        for (uint32_t j = 1; j < 16; ++j) {
            uint32_t v = u + j * stride;
            if (v >= u1) break;
            XYZZ<F> t;
            load_partial32(t, partial, v);
            quad_add(acc, t, ql);
        }
        if (ql == 0) store_partial32(partial, u, acc);
    }
}

// The same fold for inputs where EVERY bucket has many units (the precompute shapes: 2^29 points in 2^17 buckets are
// runs of 8192 = 32 units of 256): one lane per bucket adds its units up in sequence.  131 K lanes x 31 full additions
// keep the chip busy for ~0.4 ms; the quad tree above, built for a handful of hot buckets, took 3.1 ms of config 3
// scanning the 4 M entries of the full-unit list for group leaders.  Buckets of more than `thr` units stay the tree's
// (a lane must not be handed a run of thousands of units: the reference harness's repeated tile).
template <class F>
__global__ __launch_bounds__(128, 3) void k_combine_buckets(const uint32_t* __restrict__ unit_off, uint64_t G, uint32_t thr,
                                                          uint32_t* __restrict__ partial) {
    const uint64_t g = (uint64_t)blockIdx.x * 128u + threadIdx.x;
    if (g >= G) return;   // (the host passes G = hot_start when k_fold_hot takes the buckets above)
    const uint32_t u0 = unit_off[g], u1 = unit_off[g + 1];
    if (u1 - u0 < 2 || u1 - u0 > thr) return;
    if constexpr (USE_RR<F>) {
        using Q = typename F::RR;
        XYZZRR<Q> acc, t;
        ptrr_load(acc, partial, u0);
        for (uint32_t u = u0 + 1; u < u1; ++u) {
            ptrr_load(t, partial, u);
            ptrr_add<Q, 5>(acc, t);
        }
        ptrr_store(partial, u0, acc);
    } else {
        XYZZ<F> acc, t;
        load_xyzz(acc, partial, u0);
        for (uint32_t u = u0 + 1; u < u1; ++u) {
            load_xyzz(t, partial, u);
            pt_add_inl<F, 3>(acc, t);
        }
        store_xyzz(partial, u0, acc);
    }
}

#ifndef BLZ_TAIL_PRIO
#define BLZ_TAIL_PRIO 3
#endif
// ------------------------------------------------------------------------------------------------
// phase 2: per window Sum_i (i + woff) * A_i by segments.  See DESIGN.md for the recurrence:
//   F = Sum_t s_t + SEG * Sum_t t * r_t,  r_t = Sum_j A_(t SEG + j),  s_t = Sum_j (j + woff) A_(t SEG + j)
// outA[t] = r_t (weights t at the next level, woff = 0), outC[t] = Sum C_in + 2^shift * s_t.
// ------------------------------------------------------------------------------------------------
template <class F, bool FIRST>
__global__ __launch_bounds__(64, FIRST ? 2 : 3) void k_reduce_level(const uint32_t* __restrict__ inA, const uint32_t* __restrict__ inC,
                                                     const uint32_t* __restrict__ unit_off, uint32_t M, uint32_t SEG,
                                                     uint32_t T, int W, int shift, uint32_t* __restrict__ outA,
                                                     uint32_t* __restrict__ outC) {
    // level 0 (FIRST): one lane per segment, throughput-bound.  Upper levels: one DPP quad per
    // segment (ec_quad.hip.hpp), because there the sequential chain, not the work, is the cost.
    // The upper levels (and k_finish) are a few waves of sequential work on the tail stream, underneath the next task's
    // accumulation: raised wave priority, or they crawl (k_finish 1.8 ms alone, 3.0 ms underneath) - and the host, which
    // hands out the next-but-one task when this one's result arrives, enqueues that task's hidden sort too late for it
    // to finish before the main stream wants its buckets (2^22: 0.8 ms of idle main stream per MSM).
    if constexpr (!FIRST) __builtin_amdgcn_s_setprio(BLZ_TAIL_PRIO);
    const uint32_t gtid = blockIdx.x * 64u + threadIdx.x;
    const uint32_t tid = FIRST ? gtid : gtid >> 2;
    const uint32_t ql = gtid & 3u;
    if (tid >= T * (uint32_t)W) return;
    uint32_t w = tid / T, t = tid - w * T;
    uint32_t lo = t * SEG;
    uint32_t hi = lo + SEG;
    if (hi > M) hi = M;
    XYZZ<F> run, s, cs;
    pt_set_inf(run);
    pt_set_inf(s);
    pt_set_inf(cs);
    for (uint32_t i = hi; i-- > lo;) {
        XYZZ<F> a;
        size_t idx = (size_t)w * M + i;
        if constexpr (FIRST) {
            uint32_t u0 = unit_off[idx], u1 = unit_off[idx + 1];
            if (u1 > u0) load_xyzz(a, inA, u0);
            else pt_set_inf(a);
            pt_add_inl<F, 1>(run, a);
            pt_add_inl<F, 1>(s, run);  // weights i + 1 at the first level
        } else {
            load_xyzz(a, inA, idx);
            XYZZ<F> cc;
            load_xyzz(cc, inC, idx);
            quad_add(cs, cc, ql);
            quad_add(run, a, ql);
            if (i != lo) quad_add(s, run, ql);  // weights i afterwards
        }
    }
    if constexpr (FIRST) {
        cs = s;  // level 0: shift = 0 and no incoming C (and no call into the shared out-of-line group law:
                 // this kernel has its own register budget, see pt_dbl_val's TAG)
    } else {
        for (int d = 0; d < shift; ++d) quad_dbl(s, ql);
        quad_add(cs, s, ql);
        if (ql != 0) return;
    }
    store_xyzz(outA, (size_t)w * T + t, run);
    store_xyzz(outC, (size_t)w * T + t, cs);
}

// level 0 on the reduced-radix field (ec_rr.hip.hpp): the bucket sums arrive in it straight from k_accumulate, the two
// running sums stay in it, and only the segment's two results are converted to the 32-bit form the upper levels
// (DPP quads) work in.  20 % fewer multiply-adds than the 32-bit full add, and no conversion per bucket.
#ifndef BLZ_REDUCE_RR_WAVES
#define BLZ_REDUCE_RR_WAVES 2
#endif
template <class F>
__global__ __launch_bounds__(64, BLZ_REDUCE_RR_WAVES) void k_reduce_level0_rr(const uint32_t* __restrict__ inA,
                                                                            const uint32_t* __restrict__ unit_off, uint32_t M,
                                                                            uint32_t SEG, uint32_t T, int W,
                                                                            uint32_t* __restrict__ outA, uint32_t* __restrict__ outC) {
    using Q = typename F::RR;
    const uint32_t tid = blockIdx.x * 64u + threadIdx.x;
    if (tid >= T * (uint32_t)W) return;
    const uint32_t w = tid / T, t = tid - w * T;
    const uint32_t lo = t * SEG;
    uint32_t hi = lo + SEG;
    if (hi > M) hi = M;
    XYZZRR<Q> run, s;
    ptrr_set_inf(run);
    ptrr_set_inf(s);
    for (uint32_t i = hi; i-- > lo;) {
        const size_t idx = (size_t)w * M + i;
        const uint32_t u0 = unit_off[idx], u1 = unit_off[idx + 1];
        if (u1 > u0) {
            XYZZRR<Q> a;
            ptrr_load(a, inA, u0);
            ptrr_add<Q, 3>(run, a);
        }
        ptrr_add<Q, 3>(s, run);  // weights i + 1 at the first level
    }
    ptrr_store(outA, (size_t)w * T + t, run);   // the upper levels and k_finish work in the reduced radix too (ec_quad.hip.hpp)
    ptrr_store(outC, (size_t)w * T + t, s);
}

// level 0 with one DPP QUAD per segment (ec_quad.hip.hpp) for the sizes in between: too many segments for a wave each
// (k_reduce_level_row), too few to fill the chip with a lane each - 2^18 elements are 10 240 segments, 160 waves of lanes
// whose 16-addition chains at ~12 us are the kernel's 0.2 ms; a quad adds in ~6.
template <class F>
__global__ __launch_bounds__(64, 2) void k_reduce_level0_quad(const uint32_t* __restrict__ inA, const uint32_t* __restrict__ unit_off, uint32_t M,
                                                             uint32_t SEG, uint32_t T, int W, uint32_t* __restrict__ outA,
                                                             uint32_t* __restrict__ outC) {
    using Q = typename F::RR;
    const uint32_t gtid = blockIdx.x * 64u + threadIdx.x;
    const uint32_t tid = gtid >> 2, ql = gtid & 3u;
    if (tid >= T * (uint32_t)W) return;
    const uint32_t w = tid / T, t = tid - w * T;
    const uint32_t lo = t * SEG;
    uint32_t hi = lo + SEG;
    if (hi > M) hi = M;
    XYZZRR<Q> run, s;
    ptrr_set_inf(run);
    ptrr_set_inf(s);
    for (uint32_t i = hi; i-- > lo;) {
        const size_t idx = (size_t)w * M + i;
        const uint32_t u0 = unit_off[idx], u1 = unit_off[idx + 1];
        if (u1 > u0) {   // (uniform over the quad)
            XYZZRR<Q> a;
            ptrr_load(a, inA, u0);
            quadrr_add(run, a, ql);
        }
        quadrr_add(s, run, ql);  // weights i + 1 at the first level
    }
    if (ql != 0) return;
    ptrr_store(outA, (size_t)w * T + t, run);
    ptrr_store(outC, (size_t)w * T + t, s);
}

// upper levels on the reduced-radix field: one DPP quad per segment, as k_reduce_level<F, false>, with the quad group law
// of ec_quad.hip.hpp's second half (the chain of a segment is sequential: the latency of a field product is the cost)
template <class F>
__global__ __launch_bounds__(64, 2) void k_reduce_level_rr(const uint32_t* __restrict__ inA, const uint32_t* __restrict__ inC, uint32_t M,
                                                          uint32_t SEG, uint32_t T, int W, int shift, uint32_t* __restrict__ outA,
                                                          uint32_t* __restrict__ outC) {
    using Q = typename F::RR;
    __builtin_amdgcn_s_setprio(BLZ_TAIL_PRIO);
    const uint32_t gtid = blockIdx.x * 64u + threadIdx.x;
    const uint32_t tid = gtid >> 2, ql = gtid & 3u;
    if (tid >= T * (uint32_t)W) return;
    const uint32_t w = tid / T, t = tid - w * T;
    const uint32_t lo = t * SEG;
    uint32_t hi = lo + SEG;
    if (hi > M) hi = M;
    XYZZRR<Q> run, s, cs;
    ptrr_set_inf(run);
    ptrr_set_inf(s);
    ptrr_set_inf(cs);
    for (uint32_t i = hi; i-- > lo;) {
        const size_t idx = (size_t)w * M + i;
        XYZZRR<Q> a, cc;
        ptrr_load(a, inA, idx);
        ptrr_load(cc, inC, idx);
        quadrr_add(cs, cc, ql);
        quadrr_add(run, a, ql);
        if (i != lo) quadrr_add(s, run, ql);  // weights i at the upper levels
    }
    for (int d = 0; d < shift; ++d) quadrr_dbl(s, ql);
    quadrr_add(cs, s, ql);
    if (ql != 0) return;
    ptrr_store(outA, (size_t)w * T + t, run);
    ptrr_store(outC, (size_t)w * T + t, cs);
}

// A level whose segments are FEW (a small task's every level; a large task's last ones): the chain of a segment is what the
// level costs, and the row law of ec_row.hip.hpp - one point spread over the wave, an addition in ~3 us against the quad law's
// ~6 and a lone lane's ~12 - runs it two to four times faster: one segment per wave.  Same sums as k_reduce_level0_rr (LEVEL0:
// bucket sums through unit_off, weights i + 1) and k_reduce_level_rr (weights i, the carried C sums, `shift` doublings).  The
// results stay in the row law's weakly normalised form, so from the first level that takes this kernel every later level and
// the Horner walk (k_finish_row) take the row law too (run_reduce_t).
template <class F, bool LEVEL0>
__global__ __launch_bounds__(64, 4) void k_reduce_level_row(const uint32_t* __restrict__ inA, const uint32_t* __restrict__ inC,
                                                           const uint32_t* __restrict__ unit_off, uint32_t M, uint32_t SEG, uint32_t T, int W,
                                                           int shift, uint32_t* __restrict__ outA, uint32_t* __restrict__ outC) {
    using Q = typename F::RR;
    __builtin_amdgcn_s_setprio(BLZ_TAIL_PRIO);
    const uint32_t tid = blockIdx.x;   // (one wave per block)
    if (tid >= T * (uint32_t)W) return;
    const uint32_t w = tid / T, t = tid - w * T;
    const uint32_t lo = t * SEG;
    uint32_t hi = lo + SEG;
    if (hi > M) hi = M;
    const RowCtx<Q> c = row_ctx<Q>();
    RowPt run, s, cs;
    rowpt_set_inf(run);
    rowpt_set_inf(s);
    rowpt_set_inf(cs);
    for (uint32_t i = hi; i-- > lo;) {
        const size_t idx = (size_t)w * M + i;
        RowPt a;
        if constexpr (LEVEL0) {
            const uint32_t u0 = unit_off[idx], u1 = unit_off[idx + 1];
            if (u1 > u0) {
                rowpt_load<Q>(c, a, inA, u0);
                rowpt_add<Q>(c, run, a);
            }
            rowpt_add<Q>(c, s, run);
        } else {
            RowPt cc;
            rowpt_load<Q>(c, a, inA, idx);
            rowpt_load<Q>(c, cc, inC, idx);
            rowpt_add<Q>(c, cs, cc);
            rowpt_add<Q>(c, run, a);
            if (i != lo) rowpt_add<Q>(c, s, run);
        }
    }
    if constexpr (!LEVEL0) {
        for (int d = 0; d < shift; ++d) rowpt_dbl<Q>(c, s);
        rowpt_add<Q>(c, cs, s);
    }
    rowpt_store<Q>(c, outA, (size_t)w * T + t, run);
    rowpt_store<Q>(c, outC, (size_t)w * T + t, LEVEL0 ? s : cs);
}

// ------------------------------------------------------------------------------------------------
// phase 3: Horner over the window sums, normalise, emit Z | Y | X canonical little-endian
// (layout read back by tests/msm/mod.rs:397-399).  Infinity: Z = 0, Y = 1, X = 0.
// ------------------------------------------------------------------------------------------------
template <class F>
__device__ void emit_result(uint32_t* out, const XYZZ<F>& p) {
    Affine<F> a;
    bool fin = pt_to_affine(a, p);
    Fp<F> x, y, z;
    fp_zero(z);
    if (fin) {
        fp_from_mont(x, a.x);
        fp_from_mont(y, a.y);
        z.v[0] = 1;
    } else {
        fp_zero(x);
        fp_zero(y);
        y.v[0] = 1;
    }
    fp_store(out, z);
    fp_store(out + F::N, y);
    fp_store(out + 2 * F::N, x);
}

// window table of k_finish: window w owns virtual windows v0 .. v0 + m - 1 and starts at scalar bit `off`
struct FinishPlan {
    int W, logV;
    uint16_t v0[MSM_MAX_W], off[MSM_MAX_W];   // make_plan keeps m <= 128 and sum m < 2^16 (checked when filled)
    uint8_t m[MSM_MAX_W];
};

// The bucket reduce leaves, per virtual window v (V buckets), A_v = sum of its buckets and
// C_v = sum_b (b + 1) B_b.  A window made of m virtual windows has
//   S_w = sum_j C_(v0+j) + V * sum_j j * A_(v0+j),
// so the result is a Horner evaluation over the terms (off_w + log2 V, sum_j j A_j) and (off_w, sum_j C_j)
// in descending bit offset (off_(w+1) >= off_w + log2 V + 1, so the order is strict).
// (the body is written once over "a point type with quad operations": the reduced-radix accumulator for the curves that
// have one, the 32-bit XYZZ otherwise)
template <class F>
struct FinishOps32 {
    using Pt = XYZZ<F>;
    static BLZ_DEV void inf(Pt& p) { pt_set_inf(p); }
    static BLZ_DEV void load(Pt& p, const uint32_t* base, size_t idx) { load_xyzz(p, base, idx); }
    static BLZ_DEV void add(Pt& a, const Pt& b, uint32_t ql) { quad_add(a, b, ql); }
    static BLZ_DEV void dbl(Pt& a, uint32_t ql) { quad_dbl(a, ql); }
    static BLZ_DEV void to32(XYZZ<F>& o, const Pt& p) { o = p; }
};
template <class F>
struct FinishOpsRR {
    using Q = typename F::RR;
    using Pt = XYZZRR<Q>;
    static BLZ_DEV void inf(Pt& p) { ptrr_set_inf(p); }
    static BLZ_DEV void load(Pt& p, const uint32_t* base, size_t idx) { ptrr_load(p, base, idx); }
    static BLZ_DEV void add(Pt& a, const Pt& b, uint32_t ql) { quadrr_add(a, b, ql); }
    static BLZ_DEV void dbl(Pt& a, uint32_t ql) { quadrr_dbl(a, ql); }
    static BLZ_DEV void to32(XYZZ<F>& o, const Pt& p) { ptrr_to_xyzz32<F>(o, p); }
};
template <class F>
using FinishOps = std::conditional_t<USE_RR<F>, FinishOpsRR<F>, FinishOps32<F>>;

template <class F>
__global__ __launch_bounds__(64, USE_RR<F> ? 2 : 3) void k_finish(const uint32_t* __restrict__ vsumA, const uint32_t* __restrict__ vsumC,
                                                  FinishPlan fp, uint32_t* __restrict__ out) {
    // one wave; every DPP quad runs the same chain cooperatively (ec_quad.hip.hpp), lane 0 emits
    if (blockIdx.x != 0) return;
    __builtin_amdgcn_s_setprio(BLZ_TAIL_PRIO);   // (see k_reduce_level)
    using Ops = FinishOps<F>;
    using Pt = typename Ops::Pt;
    const uint32_t ql = threadIdx.x & 3u;
    Pt acc;
    Ops::inf(acc);
    int pos = 0;
    bool started = false;
    for (int w = fp.W - 1; w >= 0; --w) {
        const int v0 = fp.v0[w], m = fp.m[w], off = fp.off[w];
        if (m > 1) {
            Pt t, u;
            Ops::inf(t);
            Ops::inf(u);
            for (int j = m - 1; j >= 1; --j) {
                Pt a;
                Ops::load(a, vsumA, (size_t)(v0 + j));
                Ops::add(t, a, ql);
                Ops::add(u, t, ql);
            }
            const int o2 = off + fp.logV;
            if (started) for (int d = 0; d < pos - o2; ++d) Ops::dbl(acc, ql);
            Ops::add(acc, u, ql);
            pos = o2;
            started = true;
        }
        Pt cs;
        Ops::load(cs, vsumC, (size_t)v0);
        for (int j = 1; j < m; ++j) {
            Pt a;
            Ops::load(a, vsumC, (size_t)(v0 + j));
            Ops::add(cs, a, ql);
        }
        if (started) for (int d = 0; d < pos - off; ++d) Ops::dbl(acc, ql);
        Ops::add(acc, cs, ql);
        pos = off;
        started = true;
    }
    // scalar-range tasks: the lowest window starts at bit_lo of the scalar, the partial result carries that weight
    for (int d = 0; d < pos; ++d) Ops::dbl(acc, ql);
    if (threadIdx.x == 0) {
        XYZZ<F> o;
        Ops::to32(o, acc);
        emit_result(out, o);
    }
}

// The same walk with ONE point spread over the wave (ec_row.hip.hpp: a limb per lane, the four products of a formula round on
// the four DPP rows): a doubling costs ~1.5 us instead of the quad law's ~5, and the ~255 of them are the whole kernel.  For
// the curves on the loose 28-bit budget (BLS12-377 / 381); BN254 keeps the quad walk.
template <class F>
__global__ __launch_bounds__(64, 2) void k_finish_row(const uint32_t* __restrict__ vsumA, const uint32_t* __restrict__ vsumC, FinishPlan fp,
                                                      uint32_t* __restrict__ out) {
    using Q = typename F::RR;
    if (blockIdx.x != 0) return;
    __builtin_amdgcn_s_setprio(BLZ_TAIL_PRIO);
    __shared__ uint32_t sh[4][16];
    const RowCtx<Q> c = row_ctx<Q>();
    RowPt acc;
    rowpt_set_inf(acc);
    int pos = 0;
    bool started = false;
    for (int w = fp.W - 1; w >= 0; --w) {
        const int v0 = fp.v0[w], m = fp.m[w], off = fp.off[w];
        if (m > 1) {
            RowPt t, u;
            rowpt_set_inf(t);
            rowpt_set_inf(u);
            for (int j = m - 1; j >= 1; --j) {
                RowPt a;
                rowpt_load<Q>(c, a, vsumA, (size_t)(v0 + j));
                rowpt_add<Q>(c, t, a);
                rowpt_add<Q>(c, u, t);
            }
            const int o2 = off + fp.logV;
            if (started) for (int d = 0; d < pos - o2; ++d) rowpt_dbl<Q>(c, acc);
            rowpt_add<Q>(c, acc, u);
            pos = o2;
            started = true;
        }
        RowPt cs;
        rowpt_load<Q>(c, cs, vsumC, (size_t)v0);
        for (int j = 1; j < m; ++j) {
            RowPt a;
            rowpt_load<Q>(c, a, vsumC, (size_t)(v0 + j));
            rowpt_add<Q>(c, cs, a);
        }
        if (started) for (int d = 0; d < pos - off; ++d) rowpt_dbl<Q>(c, acc);
        rowpt_add<Q>(c, acc, cs);
        pos = off;
        started = true;
    }
    for (int d = 0; d < pos; ++d) rowpt_dbl<Q>(c, acc);
    XYZZRR<Q> r;
    rowpt_export<Q>(c, acc, sh, r);
    if (threadIdx.x == 0) {
        XYZZ<F> o;
        ptrr_to_xyzz32<F>(o, r);
        emit_result(out, o);
    }
}

template <class F>
__global__ void k_emit_infinity(uint32_t* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    XYZZ<F> acc;
    pt_set_inf(acc);
    emit_result(out, acc);
}

// partials: count x (Z | Y | X) canonical, homogeneous projective x = X/Z, y = Y/Z
template <class F>
__global__ __launch_bounds__(64, 3) void k_combine_partials(const uint32_t* __restrict__ partials, uint32_t count,
                                                         uint32_t* __restrict__ out) {
    // lanes normalise the partials in parallel (a partial with Z = 1, which is all this build ever
    // emits, needs no inversion); lane 0 then adds them in rank order and emits
    __shared__ uint32_t sh_xy[64][2 * F::N];
    __shared__ uint32_t sh_inf[64];
    XYZZ<F> acc;
    pt_set_inf(acc);
    for (uint32_t base = 0; base < count; base += 64) {
        uint32_t i = base + threadIdx.x;
        if (i < count) {
            const uint32_t* q = partials + (size_t)i * 3 * F::N;
            Fp<F> Z, Y, X;
            fp_load(Z, q);
            fp_load(Y, q + F::N);
            fp_load(X, q + 2 * F::N);
            uint32_t rest = 0;
#pragma unroll
            for (int k = 1; k < F::N; ++k) rest |= Z.v[k];
            const bool z_one = rest == 0 && Z.v[0] == 1u;
            fp_to_mont(Y, Y);
            fp_to_mont(X, X);
            bool inf = false;
            if (!z_one) {
                fp_to_mont(Z, Z);
                inf = fp_is_zero(Z);
                if (!inf) {
                    Fp<F> zi;
                    fp_inv(zi, Z);
                    fp_mul(X, X, zi);
                    fp_mul(Y, Y, zi);
                }
            }
            fp_store(&sh_xy[threadIdx.x][0], X);
            fp_store(&sh_xy[threadIdx.x][F::N], Y);
            sh_inf[threadIdx.x] = inf ? 1u : 0u;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t m = count - base < 64 ? count - base : 64;
            for (uint32_t k = 0; k < m; ++k) {
                if (sh_inf[k]) continue;
                Affine<F> a;
                fp_load(a.x, &sh_xy[k][0]);
                fp_load(a.y, &sh_xy[k][F::N]);
                pt_madd(acc, a);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) emit_result(out, acc);
}


// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
template <class F>
int points_to_mont_t(MsmEngine& E, const void* d_raw, void* d_mont, uint32_t npts) {
    if (npts == 0) return BLZ_OK;
    hipLaunchKernelGGL(k_points_to_mont<F>, dim3((npts + 255) / 256), dim3(256), 0, E.stream, (const uint32_t*)d_raw,
                       (uint32_t*)d_mont, npts);
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

template <class F>
int emit_infinity_t(MsmEngine& E) {
    hipLaunchKernelGGL(k_emit_infinity<F>, dim3(1), dim3(64), 0, E.stream, E.slot_result(E.cur));
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

// The same fold for SMALL tasks, where a lane's chain of additions is the latency of the whole pipeline: the top window of a
// 2^13-element plan holds 5 real bits, so its 32 buckets get 256 entries = 16 - 32 units each, and one lane folding 31 units at
// ~14 us per addition kept everything waiting for 0.44 ms of a 3.6 ms MSM.  Here one wave takes a bucket: lane i loads unit i,
// six rounds of pairwise additions (operands moved between lanes with ds_bpermute: 56 dwords per round, nothing against an
// addition) leave the sum in lane 0 after log2(units) additions' worth of time.  One block per bucket - for small bucket
// spaces only.
template <class F>
__global__ __launch_bounds__(64, 2) void k_combine_buckets_wave(const uint32_t* __restrict__ unit_off, uint32_t thr,
                                                               uint32_t* __restrict__ partial) {
    using Q = typename F::RR;
    const uint64_t g = blockIdx.x;
    const uint32_t u0 = unit_off[g], U = unit_off[g + 1] - u0;
    if (U < 2 || U > thr) return;   // (uniform over the wave)
    const uint32_t lane = threadIdx.x;
    XYZZRR<Q> acc;
    if (lane < U) ptrr_load(acc, partial, u0 + lane);
    else ptrr_set_inf(acc);
    for (uint32_t r = 1; r < U && r < 64; r <<= 1) {
        XYZZRR<Q> o;
#pragma unroll
        for (int i = 0; i < Q::NL; ++i) {
            o.x.v[i] = __shfl_down(acc.x.v[i], r, 64);
            o.y.v[i] = __shfl_down(acc.y.v[i], r, 64);
            o.zz.v[i] = __shfl_down(acc.zz.v[i], r, 64);
            o.zzz.v[i] = __shfl_down(acc.zzz.v[i], r, 64);
        }
        if ((lane & (2u * r - 1u)) == 0 && lane + r < U) ptrr_add<Q, 6>(acc, o);
    }
    if (lane == 0) ptrr_store(partial, u0, acc);
}

// The same fold on the row law (ec_row.hip.hpp) for the tasks whose whole tail runs on it (small_row_tail: every reduce level is a
// row level): one wave per bucket adds the bucket's units in sequence, ~3 us per addition - against ~20 us per ROUND of the
// shuffle tree above and k_fold_hot's 0.09 ms for the 16-unit buckets of a 2^13 task's top window - and leaves the sum, in
// the row law's weakly normalised form (read by k_reduce_level_row only), in the bucket's first unit.
template <class F, bool STRICT>   // STRICT: the sum leaves in the accumulator form every reader takes (level 0 on the thread-level law)
__global__ __launch_bounds__(64, 4) void k_combine_buckets_row(const uint32_t* __restrict__ unit_off, uint32_t thr, uint32_t* __restrict__ partial) {
    using Q = typename F::RR;
    const uint64_t g = blockIdx.x;
    const uint32_t u0 = unit_off[g], U = unit_off[g + 1] - u0;
    if (U < 2 || U > thr) return;   // (uniform over the wave; longer buckets were folded by k_combine_units' tree)
    const RowCtx<Q> c = row_ctx<Q>();
    RowPt acc;
    rowpt_load<Q>(c, acc, partial, u0);
    for (uint32_t u = 1; u < U; ++u) {
        RowPt a;
        rowpt_load<Q>(c, a, partial, u0 + u);
        rowpt_add<Q>(c, acc, a);
    }
    if constexpr (STRICT) rowpt_store_strict<Q>(c, partial, u0, acc);
    else rowpt_store<Q>(c, partial, u0, acc);
}

// k_fold_hot on the row law: sixteen waves take a hot bucket, wave w adds the units w, w + 16, ... in sequence (~3 us each: 32 of
// the 512 units a 2^16 task's top buckets hold), the sixteen sums meet in the LDS and four rounds of pairwise additions leave
// the bucket's sum, in the accumulator form, in its first unit: 0.33 -> 0.1 ms at 2^16.
template <class F>
__global__ __launch_bounds__(1024, 1) void k_fold_hot_row(const uint32_t* __restrict__ unit_off, uint32_t hot_start, uint32_t* __restrict__ partial) {
    using Q = typename F::RR;
    __shared__ uint32_t sh[16][4][16];
    const uint64_t g = (uint64_t)hot_start + blockIdx.x;
    const uint32_t u0 = unit_off[g], U = unit_off[g + 1] - u0;
    if (U < 2) return;   // (uniform over the block)
    const uint32_t wave = threadIdx.x >> 6;
    const RowCtx<Q> c = row_ctx<Q>();
    RowPt acc;
    rowpt_set_inf(acc);
    for (uint32_t u = wave; u < U; u += 16u) {
        RowPt a;
        rowpt_load<Q>(c, a, partial, u0 + u);
        rowpt_add<Q>(c, acc, a);
    }
    for (uint32_t r = 8; r >= 1; r >>= 1) {
        if (wave >= r && wave < 2u * r && c.row == 0u) {
            sh[wave][0][c.li] = acc.x;
            sh[wave][1][c.li] = acc.y;
            sh[wave][2][c.li] = acc.zz;
            sh[wave][3][c.li] = acc.zzz;
        }
        __syncthreads();
        if (wave < r) {
            RowPt o;
            o.x = sh[wave + r][0][c.li];
            o.y = sh[wave + r][1][c.li];
            o.zz = sh[wave + r][2][c.li];
            o.zzz = sh[wave + r][3][c.li];
            rowpt_add<Q>(c, acc, o);
        }
        __syncthreads();
    }
    if (wave == 0) rowpt_store_strict<Q>(c, partial, u0, acc);
}

// ... and for the windows the PLAN knows to be hot - the top windows of a small or mid-sized task hold the last few bits of
// the scalars: 2^18 elements in 20 windows of 13 bits leave 8 bits for the top one, 256 buckets of 1024 entries = 64 units
// each; 2^16 in 22 x 12: 3 bits, 8 buckets of 512 units - eight waves take a bucket: every wave folds chunks of 64 units by
// the shuffle tree, lane 0 adds the chunk sums up, the eight wave sums meet in LDS.  A lone 2^18 MSM spent 1.7 of its 5.8 ms
// in the lane-per-bucket fold (63 additions in sequence) and the quad tree's passes; this takes 0.15.
template <class F>
__global__ __launch_bounds__(512, 1) void k_fold_hot(const uint32_t* __restrict__ unit_off, uint32_t hot_start, uint32_t* __restrict__ partial) {
    using Q = typename F::RR;
    constexpr int S = ptrr_dwords<Q>();
    __shared__ uint32_t sh[8 * S];
    const uint64_t g = (uint64_t)hot_start + blockIdx.x;
    const uint32_t u0 = unit_off[g], U = unit_off[g + 1] - u0;
    if (U < 2) return;   // (uniform over the block)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    XYZZRR<Q> tot;
    ptrr_set_inf(tot);
    for (uint32_t base = wave * 64u; base < U; base += 8u * 64u) {
        const uint32_t cnt = U - base < 64u ? U - base : 64u;
        XYZZRR<Q> acc;
        if (lane < cnt) ptrr_load(acc, partial, u0 + base + lane);
        else ptrr_set_inf(acc);
        for (uint32_t r = 1; r < cnt; r <<= 1) {
            XYZZRR<Q> o;
#pragma unroll
            for (int i = 0; i < Q::NL; ++i) {
                o.x.v[i] = __shfl_down(acc.x.v[i], r, 64);
                o.y.v[i] = __shfl_down(acc.y.v[i], r, 64);
                o.zz.v[i] = __shfl_down(acc.zz.v[i], r, 64);
                o.zzz.v[i] = __shfl_down(acc.zzz.v[i], r, 64);
            }
            if ((lane & (2u * r - 1u)) == 0 && lane + r < cnt) ptrr_add<Q, 7>(acc, o);
        }
        if (lane == 0) ptrr_add<Q, 7>(tot, acc);
    }
    if (lane == 0) ptrr_store(sh, wave, tot);
    __syncthreads();
    if (wave != 0) return;
    XYZZRR<Q> acc;
    if (lane < 8) ptrr_load(acc, sh, lane);
    else ptrr_set_inf(acc);
    for (uint32_t r = 1; r < 8; r <<= 1) {
        XYZZRR<Q> o;
#pragma unroll
        for (int i = 0; i < Q::NL; ++i) {
            o.x.v[i] = __shfl_down(acc.x.v[i], r, 64);
            o.y.v[i] = __shfl_down(acc.y.v[i], r, 64);
            o.zz.v[i] = __shfl_down(acc.zz.v[i], r, 64);
            o.zzz.v[i] = __shfl_down(acc.zzz.v[i], r, 64);
        }
        if ((lane & (2u * r - 1u)) == 0 && lane + r < 8) ptrr_add<Q, 7>(acc, o);
    }
    if (lane == 0) ptrr_store(partial, u0, acc);
}

// bucket_sums[g] += sum of bucket g in this piece, for the buckets whose run needed SEVERAL units (piecewise tasks, msm.hip:
// the single-unit buckets were summed in place by k_accumulate_cont).  After the unit folds the sum of such a run sits in its
// first unit.
template <class F>
__global__ __launch_bounds__(128, 2) void k_merge_buckets(const uint32_t* __restrict__ partial, const uint32_t* __restrict__ unit_off,
                                                         uint64_t G, uint32_t* __restrict__ sums) {
    const uint64_t g = (uint64_t)blockIdx.x * 128u + threadIdx.x;
    if (g >= G) return;
    const uint32_t u0 = unit_off[g], u1 = unit_off[g + 1];
    if (u1 - u0 < 2 || u1 < u0) return;
    if constexpr (USE_RR<F>) {
        using Q = typename F::RR;
        XYZZRR<Q> a, b;
        ptrr_load(a, sums, g);
        ptrr_load(b, partial, u0);
        ptrr_add<Q, 4>(a, b);
        ptrr_store(sums, g, a);
    } else {
        XYZZ<F> a, b;
        load_xyzz(a, sums, g);
        load_xyzz(b, partial, u0);
        pt_add_inl<F, 2>(a, b);
        store_xyzz(sums, g, a);
    }
}

// phase 1 after a digit sort: at most U units (the real count is in E.sb().stats on the device)
// level 0's segment length (run_reduce_t) and whether the task's reduce runs on the row law from level 0 on
inline uint32_t reduce_seg0(const MsmPlan& P) {
    uint32_t seg0_auto = 64;
    while (seg0_auto > 8 && P.G / seg0_auto < 262144) seg0_auto >>= 1;
    return (uint32_t)exp_knob("BLAZE_MSM_SEG", (int)seg0_auto);
}
inline uint32_t reduce_row_max() { return (uint32_t)exp_knob("BLAZE_REDUCE_ROW_MAX", 8192); }
template <class F>
bool small_row_tail(const MsmPlan& P) {
    if constexpr (USE_RR<F>) {
        if constexpr (!RR_TIGHT<typename F::RR>) {
            if (exp_knob("BLAZE_FINISH_ROW", 1) == 0) return false;
            const uint32_t seg0 = reduce_seg0(P);
            return (uint64_t)((P.Bw + seg0 - 1) / seg0) * (uint64_t)P.Wv <= reduce_row_max();
        }
    }
    return false;
}

template <class F>
int run_accumulate_t(MsmEngine& E, const void* d_pts, uint32_t U, int slice) {
    hipStream_t st = E.stream;
    MsmSlot& S = E.slots[E.cur];
    const MsmPlan& P = E.last_plan;
    BLZ_TRY(E.partial.reserve(((size_t)U + 1) * 4 * partial_dwords<F>()));
    if (slice <= 0) BLZ_HIP(hipEventRecord(S.ev[1], st), BLZ_ERR_UNKNOWN);   // (piecewise task: the FIRST piece's sort stage is done)
    S.accum_timed = true;
    // ev5..ev6 (or the slice's pair) bracket the dominant kernel alone
    BLZ_HIP(hipEventRecord(slice < 0 ? S.ev[5] : S.slice_ev[2 * slice], st), BLZ_ERR_UNKNOWN);
    if (slice < 0)
        hipLaunchKernelGGL(k_accumulate<F>, dim3((U + 127) / 128), dim3(128), 0, st, (const uint32_t*)d_pts,
                           E.sb().entries.as<uint32_t>(), E.sb().off.as<uint32_t>(), E.sb().unit_off.as<uint32_t>(),
                           E.sb().unit_bucket.as<uint32_t>(), E.sb().unit_order.as<uint32_t>(), E.sb().stats.as<uint32_t>(), P.L,
                           E.partial.as<uint32_t>());
    else   // piecewise task: single-unit buckets carry their sums in bucket_sums from piece to piece
        hipLaunchKernelGGL(k_accumulate_cont<F>, dim3((U + 127) / 128), dim3(128), 0, st, (const uint32_t*)d_pts,
                           E.sb().entries.as<uint32_t>(), E.sb().off.as<uint32_t>(), E.sb().unit_off.as<uint32_t>(),
                           E.sb().unit_bucket.as<uint32_t>(), E.sb().unit_order.as<uint32_t>(), E.sb().stats.as<uint32_t>(), P.L,
                           E.partial.as<uint32_t>(), E.bucket_sums.as<uint32_t>(), slice == 0 ? 1u : 0u);
    BLZ_HIP(hipEventRecord(slice < 0 ? S.ev[6] : S.slice_ev[2 * slice + 1], st), BLZ_ERR_UNKNOWN);
    // a bucket holds at most one entry per point (window-table tasks: one per point and window)
    const uint64_t maxunits = ((uint64_t)P.npts * (P.table ? P.W : 1) + P.L - 1) / P.L;
    uint64_t full_bound = (uint64_t)P.npts * P.W / P.L + 1;  // units of length L: at most entries / L
    if (full_bound > U) full_bound = U;
    // where the plan itself says that buckets hold several units each (mean run > L / 2), the lane-per-bucket fold takes
    // every bucket of up to 64 units and the tree only the hot ones beyond
    // (... and for every task of up to 2^22 points: one window of such a task can be twice as dense as the mean - the top real
    // window of 255-bit scalars below r covers 0x39f6 of its 2^15 buckets at c = 16 - and the quad tree, built for a handful of
    // hot buckets, spent 0.48 ms of a 4.8 ms 2^20 task on its 14 K two-unit buckets; one pass over the unit offsets is nothing here)
    const uint32_t thr = ((uint64_t)P.npts * P.W / (P.G ? P.G : 1) > P.L / 2 || P.npts <= (1u << 22)) ? 64u : 0u;
    // The windows the plan knows to be hot (the top ones, where the scalars' bits run out: a few buckets with long runs):
    // a suffix [hot_start, G) of the bucket space goes to k_fold_hot - eight waves per bucket - and the two folds below
    // leave it alone.  Only for a small suffix of a larger space: where EVERY window is like that (the precompute shapes) the
    // lane-per-bucket fold below is the throughput-bound answer.
    uint32_t hot_start = (uint32_t)P.G;
    // Where the curve has the row law (ec_row.hip.hpp), a task of up to 2^17 bucket slots and units folds its buckets on it: one wave per bucket
    // (k_combine_buckets_row, buckets of up to 64 units), sixteen per bucket of the plan's hot windows (k_fold_hot_row).  The
    // sums leave in the accumulator form unless everything behind them runs on the row law too (small_row_tail).
    bool row_law = false;
    if constexpr (USE_RR<F>) row_law = !RR_TIGHT<typename F::RR> && exp_knob("BLAZE_FINISH_ROW", 1) != 0 && exp_knob("BLAZE_FOLD_ROW", 1) != 0;
    // (a latency tool: chip-wide the row law adds ~5 x slower than one lane per point - 2^18 elements, 82 K buckets of four units:
    // 0.41 ms against the lane-per-bucket fold's 0.19 - so only while the units to fold are few)
    const bool row_fold = row_law && slice < 0 && thr != 0 && P.G <= (1u << 17) && (uint64_t)P.npts * P.W / P.L <= (1u << 17);
    const bool row_tail = row_fold && small_row_tail<F>(P);
    if constexpr (USE_RR<F>) {
        if (!P.table && slice < 0 && P.ebits > 0 && exp_knob("BLAZE_FOLD_HOT", 1) != 0) {
            int lowest = -1, off = 0;
            bool any = false;
            int offs[MSM_MAX_W];
            for (int w = 0; w < P.W; ++w) { offs[w] = off; off += P.width[w]; }
            for (int w = P.W - 1; w >= 0; --w) {
                const int cw = P.width[w];
                int t = P.ebits - offs[w];
                if (t > cw) t = cw;
                const double slots = (double)(1ull << (cw - 1));
                double active = t >= cw ? slots : t > 0 ? (double)(1ull << t) + 1.0 : t == 0 ? 1.0 : 0.0;
                if (active > slots) active = slots;
                const double entries = t >= 0 ? (double)P.npts : 0.0;
                const bool hot = active > 0 && entries / active >= 12.0 * (double)P.L;   // a dozen units or more per bucket
                if (!hot && entries > 0) break;       // a normal window: the suffix ends above it
                lowest = w;
                any = any || hot;
            }
            if (any && lowest > 0 && P.G - P.boff[lowest] <= 16384) hot_start = P.boff[lowest];
        }
    }
    // (64-bit stride: with BLAZE_MSM_L < 8 and close to 2^31 points, maxunits exceeds 2^28 and a u32 stride would
    // wrap to 0 - an endless launch loop)
    for (uint64_t stride = 1; stride < maxunits; stride *= 16)
        hipLaunchKernelGGL(k_combine_units<F>, dim3((uint32_t)((full_bound / 16 + 1) * 4 / 128 + 1)), dim3(128), 0, st,
                           E.sb().unit_off.as<uint32_t>(), E.sb().unit_bucket.as<uint32_t>(), E.sb().unit_order.as<uint32_t>(),
                           E.sb().lenhist.as<uint32_t>() + P.L, E.sb().stats.as<uint32_t>(), P.L, (uint32_t)stride, thr, hot_start,
                           E.partial.as<uint32_t>());
    if constexpr (USE_RR<F>) {
        if (hot_start < P.G) {
            bool done = false;
            if constexpr (!RR_TIGHT<typename F::RR>) {
                if (row_law) {
                    hipLaunchKernelGGL(k_fold_hot_row<F>, dim3((uint32_t)(P.G - hot_start)), dim3(1024), 0, st, E.sb().unit_off.as<uint32_t>(), hot_start,
                                       E.partial.as<uint32_t>());
                    done = true;
                }
            }
            if (!done)
                hipLaunchKernelGGL(k_fold_hot<F>, dim3((uint32_t)(P.G - hot_start)), dim3(512), 0, st, E.sb().unit_off.as<uint32_t>(), hot_start,
                                   E.partial.as<uint32_t>());
        }
    }
    if (row_fold) {
        if constexpr (USE_RR<F>) {
            if constexpr (!RR_TIGHT<typename F::RR>) {
                if (hot_start > 0) {
                    if (row_tail)
                        hipLaunchKernelGGL((k_combine_buckets_row<F, false>), dim3(hot_start), dim3(64), 0, st, E.sb().unit_off.as<uint32_t>(), thr,
                                           E.partial.as<uint32_t>());
                    else
                        hipLaunchKernelGGL((k_combine_buckets_row<F, true>), dim3(hot_start), dim3(64), 0, st, E.sb().unit_off.as<uint32_t>(), thr,
                                           E.partial.as<uint32_t>());
                }
            }
        }
    } else if (thr && hot_start > 0) {
        bool wave = false;
        if constexpr (USE_RR<F>) wave = hot_start <= 32768;   // small bucket spaces: one wave per bucket (latency), else one lane (throughput)
        if (wave) {
            if constexpr (USE_RR<F>)
                hipLaunchKernelGGL(k_combine_buckets_wave<F>, dim3(hot_start), dim3(64), 0, st, E.sb().unit_off.as<uint32_t>(), thr,
                                   E.partial.as<uint32_t>());
        } else {
            hipLaunchKernelGGL(k_combine_buckets<F>, dim3((hot_start + 127) / 128), dim3(128), 0, st, E.sb().unit_off.as<uint32_t>(),
                               (uint64_t)hot_start, thr, E.partial.as<uint32_t>());
        }
    }
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

template <class F>
int merge_buckets_t(MsmEngine& E) {
    const uint64_t G = E.last_plan.G;
    hipLaunchKernelGGL(k_merge_buckets<F>, dim3((uint32_t)((G + 127) / 128)), dim3(128), 0, E.stream, E.partial.as<uint32_t>(),
                       E.sb().unit_off.as<uint32_t>(), G, E.bucket_sums.as<uint32_t>());
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

// phases 2 - 3: bucket reduce over the sums at sums[unit_off[g]], then the window combine
template <class F>
int run_reduce_t(MsmEngine& E, const void* sums, const void* unit_off_v) {
    hipStream_t st = E.stream;
    MsmSlot& S = E.slots[E.cur];
    const MsmPlan& P = E.last_plan;
    const uint32_t* unit_off = (const uint32_t*)unit_off_v;
    BLZ_HIP(hipEventRecord(S.ev[2], st), BLZ_ERR_UNKNOWN);

    // ---- phase 2
    // level 0 is throughput-bound (2 adds per bucket): long segments; the upper levels have few
    // lanes and are latency-bound on their sequential chain: short segments, more levels
    // (a small bucket space cannot fill the chip with long segments: the level wants >= 2^18 lanes - two full rounds of
    // two waves per SIMD - before it wants long segments; the extra segment sums are absorbed by the upper levels, which
    // run on the tail stream).  Same-box sweeps, profiles/r03_seg_sweep.txt: 17.8 M bucket slots (the 2^26 plan) 64 best; 12.6 M: 32 (61.9 against 62.9 ms per MSM);
    // 5 - 7 M: 16 (17.2 against 17.8); 2.1 M (2^22): 8 (10.1 against 11.1).  Powers of two only: the upper levels weigh
    // segment t by shifts.
    const uint32_t SEG0 = reduce_seg0(P);
    const uint32_t SEGU = (uint32_t)exp_knob("BLAZE_MSM_SEG_UPPER", 8);
    uint32_t M = P.Bw;
    int level = 0, shift = 0;
    // the Horner walk and the levels of few segments run on the row law (ec_row.hip.hpp) where the curve has it
    bool row_walk = false, row_levels = false;
    if constexpr (USE_RR<F>) row_walk = !RR_TIGHT<typename F::RR> && exp_knob("BLAZE_FINISH_ROW", 1) != 0;
    const uint32_t row_max = reduce_row_max();
    const uint32_t* curA = (const uint32_t*)sums;
    const uint32_t* curC = nullptr;
    for (;;) {
        const uint32_t SEG = level == 0 ? SEG0 : SEGU;
        int seglog = 0;
        while ((1u << seglog) < SEG) ++seglog;
        uint32_t T = (M + SEG - 1) / SEG;
        DevBuf& oA = S.lvlA[level & 1];
        DevBuf& oC = S.lvlC[level & 1];
        BLZ_TRY(oA.reserve((size_t)T * P.Wv * 4 * partial_dwords<F>()));   // (reduced-radix accumulators where the curve has them)
        BLZ_TRY(oC.reserve((size_t)T * P.Wv * 4 * partial_dwords<F>()));
        uint32_t nthreads = T * (uint32_t)P.Wv;
        // few segments: one per wave on the row law, from here to the end (k_reduce_level_row)
        if (row_walk && nthreads <= row_max) row_levels = true;
        if (row_levels) {
            if constexpr (USE_RR<F>) {
                if constexpr (!RR_TIGHT<typename F::RR>) {
                    if (level == 0)
                        hipLaunchKernelGGL((k_reduce_level_row<F, true>), dim3(nthreads), dim3(64), 0, st, curA, curC, unit_off, M, SEG, T, P.Wv,
                                           shift, oA.as<uint32_t>(), oC.as<uint32_t>());
                    else
                        hipLaunchKernelGGL((k_reduce_level_row<F, false>), dim3(nthreads), dim3(64), 0, st, curA, curC, unit_off, M, SEG, T, P.Wv,
                                           shift, oA.as<uint32_t>(), oC.as<uint32_t>());
                }
            }
            if (level == 0) {
                BLZ_HIP(hipEventRecord(S.ev_l0, st), BLZ_ERR_UNKNOWN);
                st = E.tail_stream;
                BLZ_HIP(hipStreamWaitEvent(st, S.ev_l0, 0), BLZ_ERR_UNKNOWN);
            }
        } else if (level == 0) {
            if constexpr (USE_RR<F>) {
                if (nthreads <= (uint32_t)exp_knob("BLAZE_REDUCE_QUAD_MAX", 131072))
                    hipLaunchKernelGGL(k_reduce_level0_quad<F>, dim3((nthreads * 4 + 63) / 64), dim3(64), 0, st, curA, unit_off, M,
                                       SEG, T, P.Wv, oA.as<uint32_t>(), oC.as<uint32_t>());
                else
                    hipLaunchKernelGGL(k_reduce_level0_rr<F>, dim3((nthreads + 63) / 64), dim3(64), 0, st, curA, unit_off, M,
                                       SEG, T, P.Wv, oA.as<uint32_t>(), oC.as<uint32_t>());
            } else
                hipLaunchKernelGGL((k_reduce_level<F, true>), dim3((nthreads + 63) / 64), dim3(64), 0, st, curA, curC,
                                   unit_off, M, SEG, T, P.Wv, shift, oA.as<uint32_t>(), oC.as<uint32_t>());
            // the rest is a few lanes of sequential work: hand it to the tail stream, so this stream can
            // start the next task's sort while it runs
            BLZ_HIP(hipEventRecord(S.ev_l0, st), BLZ_ERR_UNKNOWN);
            st = E.tail_stream;
            BLZ_HIP(hipStreamWaitEvent(st, S.ev_l0, 0), BLZ_ERR_UNKNOWN);
        } else {
            if constexpr (USE_RR<F>)
                hipLaunchKernelGGL(k_reduce_level_rr<F>, dim3((nthreads * 4 + 63) / 64), dim3(64), 0, st, curA, curC, M, SEG, T, P.Wv,
                                   shift, oA.as<uint32_t>(), oC.as<uint32_t>());
            else
                hipLaunchKernelGGL((k_reduce_level<F, false>), dim3((nthreads * 4 + 63) / 64), dim3(64), 0, st, curA, curC,
                                   unit_off, M, SEG, T, P.Wv, shift, oA.as<uint32_t>(), oC.as<uint32_t>());
        }
        curA = oA.as<uint32_t>();
        curC = oC.as<uint32_t>();
        shift += seglog;
        M = T;
        ++level;
        if (T == 1) break;
    }
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    BLZ_HIP(hipEventRecord(S.ev[3], st), BLZ_ERR_UNKNOWN);

    // ---- phase 3
    FinishPlan fp;
    fp.W = P.W;
    fp.logV = 0;
    while ((1u << fp.logV) < P.Bw) ++fp.logV;
    if (P.table) {
        // one bucket set for all the scalar's windows (their weights are in the table's points): a single window at bit 0
        fp.W = 1;
        fp.v0[0] = 0;
        fp.m[0] = (uint8_t)(P.G >> fp.logV);
        fp.off[0] = 0;
    } else {
        int off = P.base_bit;
        for (int w = 0; w < P.W; ++w) {
            const uint32_t v0 = P.boff[w] >> fp.logV, m = (P.boff[w + 1] - P.boff[w]) >> fp.logV;
            if (v0 > 0xffffu || m > 0xffu || off > 0xffff)
                return fail(BLZ_ERR_UNKNOWN, "window plan outside k_finish's table range (window %d: v0=%u m=%u off=%d)", w, v0, m, off);
            fp.v0[w] = (uint16_t)v0;
            fp.m[w] = (uint8_t)m;
            fp.off[w] = (uint16_t)off;
            off += P.width[w];
        }
    }
    if (row_walk) {
        if constexpr (USE_RR<F>) {
            if constexpr (!RR_TIGHT<typename F::RR>)
                hipLaunchKernelGGL(k_finish_row<F>, dim3(1), dim3(64), 0, st, curA, curC, fp, E.slot_result(E.cur));
        }
    } else {
        hipLaunchKernelGGL(k_finish<F>, dim3(1), dim3(64), 0, st, curA, curC, fp, E.slot_result(E.cur));
    }
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    BLZ_HIP(hipEventRecord(S.ev[4], st), BLZ_ERR_UNKNOWN);
    BLZ_TRY(copy_words_to_pinned(S.result_h, E.slot_result(E.cur), 3 * F::N, st));   // (not a copy-engine transfer: msm_engine.hpp)
    BLZ_HIP(hipEventRecord(S.ev_done, st), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

template <class F>
int combine_t(MsmEngine& E, const uint8_t* partials, size_t count, uint8_t* out, bool on_device) {
    // own stream: it must not queue behind a task in flight (its tail waits for its accumulation)
    hipStream_t st = E.aux_stream;
    size_t rs = 3 * F::N * 4;
    DevBuf tmp;
    uint32_t* d_out = E.result.as<uint32_t>() + 128;
    uint32_t* d_in = E.result.as<uint32_t>() + 256;  // 15 KiB of the result buffer: up to 100 partials without an allocation
    if (on_device) {
        d_in = (uint32_t*)partials;   // e.g. the receive buffer of the RCCL all-gather, written on this stream
    } else {
        if (rs * count > 15 * 1024) {
            BLZ_TRY(tmp.reserve(rs * count));
            d_in = tmp.as<uint32_t>();
        }
        if (count) BLZ_HIP(hipMemcpyAsync(d_in, partials, rs * count, hipMemcpyHostToDevice, st), BLZ_ERR_WRITE);
    }
    hipLaunchKernelGGL(k_combine_partials<F>, dim3(1), dim3(64), 0, st, d_in, (uint32_t)count, d_out);
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    BLZ_TRY(copy_words_to_pinned(E.combine_h, d_out, (uint32_t)(rs / 4), st));
    const int rc = sync_stream_bounded(st, on_device ? "all_gather_combine: exchange + combine" : "combine_partials");
    if (rc == BLZ_OK) memcpy(out, E.combine_h, rs);
    if (rc == BLZ_OK || !wait_timed_out()) tmp.release();   // (a wedged stream may still read it: leak rather than block)
    return rc;
}

constexpr uint32_t TABLE_BUILD_BLOCKS = 256 * 4 * 3;   // 64-lane blocks: three waves on every SIMD
template <class F>
size_t table_scratch_bytes_t(int W) {
    return (size_t)TABLE_BUILD_BLOCKS * 64 * (size_t)(W > 1 ? W : 1) * TABLE_SCRATCH_ROW<F> * 4;
}
template <class F>
int build_table_t(MsmEngine& E, const void* d_raw, void* d_table, uint32_t npts, int c, int W, int base_shift, void* scratch,
                  uint32_t* flag, hipStream_t st) {
    (void)E;
    if (npts == 0) return BLZ_OK;
    uint32_t blocks = (npts + 63) / 64;
    if (blocks > TABLE_BUILD_BLOCKS) blocks = TABLE_BUILD_BLOCKS;
    hipLaunchKernelGGL(k_build_window_table<F>, dim3(blocks), dim3(64), 0, st, (const uint32_t*)d_raw, (uint32_t*)d_table, npts, c, W,
                       base_shift, (uint32_t*)scratch, flag);
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

template <class F>
int points_from_mont_t(const void* d_mont, void* d_raw, uint64_t npts, hipStream_t st) {
    if (npts == 0) return BLZ_OK;
    hipLaunchKernelGGL(k_points_from_mont<F>, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, st, (const uint32_t*)d_mont, (uint32_t*)d_raw, npts);
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}
template <class F>
int points_all_canonical_t(const void* d_raw, uint64_t npts, uint32_t* flag, hipStream_t st) {
    if (npts == 0) return BLZ_OK;
    hipLaunchKernelGGL(k_points_all_canonical<F>, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, st, (const uint32_t*)d_raw, npts, flag);
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

constexpr uint32_t CHECK_BLOCKS = 256 * 4 * 3;   // 64-lane blocks: three waves on every SIMD
template <class F>
int points_to_mont_even_t(MsmEngine& E, const void* d_raw, void* d_mont, uint32_t nq) {
    if (nq == 0) return BLZ_OK;
    hipLaunchKernelGGL(k_points_to_mont_even<F>, dim3((nq + 255) / 256), dim3(256), 0, E.stream, (const uint32_t*)d_raw, (uint32_t*)d_mont, nq);
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}
template <class F>
int check_precompute_t(MsmEngine& E, const void* d_raw, uint64_t nelem, uint32_t* flag, hipStream_t st) {
    (void)E;
    if (nelem == 0) return BLZ_OK;
    const uint64_t want = (nelem * 7u + 63u) / 64u;
    const uint32_t blocks = (uint32_t)(want < CHECK_BLOCKS ? want : CHECK_BLOCKS);
    hipLaunchKernelGGL(k_check_precompute<F>, dim3(blocks), dim3(64), 0, st, (const uint32_t*)d_raw, nelem, flag);
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

template <class F>
int accumulate_vgprs_t() {
    hipFuncAttributes a;
    if (hipFuncGetAttributes(&a, (const void*)k_accumulate<F>) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return a.numRegs;
}

template <class F>
MsmCurveOps make_ops() {
    MsmCurveOps o;
    o.points_to_mont = &points_to_mont_t<F>;
    o.emit_infinity = &emit_infinity_t<F>;
    o.run_accumulate = &run_accumulate_t<F>;
    o.merge_buckets = &merge_buckets_t<F>;
    o.run_reduce = &run_reduce_t<F>;
    o.partial_dwords = partial_dwords<F>();
    o.accumulate_vgprs = &accumulate_vgprs_t<F>;
    o.build_table = &build_table_t<F>;
    o.table_scratch_bytes = &table_scratch_bytes_t<F>;
    o.combine = &combine_t<F>;
    o.points_to_mont_even = &points_to_mont_even_t<F>;
    o.check_precompute = &check_precompute_t<F>;
    o.points_from_mont = &points_from_mont_t<F>;
    o.points_all_canonical = &points_all_canonical_t<F>;
    return o;
}

}  // namespace blz
