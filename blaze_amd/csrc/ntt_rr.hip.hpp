// The 512-point NTT pass (k_ntt512 of ntt_impl.hip.hpp) on the reduced-radix scalar field: 9 limbs of 29 bits
// (field_rr.hip.hpp; curve_constants.h Fr_*_RR).  Same decomposition (512 = 8 * 8 * 8, a lane holds 8 elements, two
// LDS exchanges per pass - the first across the block, the second inside a wave - inter-pass twiddle stepped along the
// lane's outputs), different arithmetic:
//   * a field product is 162 v_mad_u64_u32 (+ 17 shifts, 9 quotient digits) instead of 128 x (v_mad_u64_u32 +
//     v_addc_co_u32);
//   * a butterfly is 27 plain 32-bit adds / subs - no carry chain, no compare, no select: u - v is
//     u + (2^J m - v) limb by limb, with the multiple of m in borrow form (tools/gen_constants.py).  The 3 spare
//     bits per limb and the 6-8 spare bits of R_rr / m are a tight budget, so the subtrahend is always a
//     normalised value (three of an 8-point DFT's twelve are sums and get a carry propagation first), every DFT
//     twiddled output leaves through a product, which brings its value back under 2m, an un-twiddled one (and a
//     forward transform's last output) through a one-digit quotient reduction (rr_reduce2m), and the twiddle
//     tables are canonical (< m);
//   * every intermediate's limb and value bounds are part of its type, so the compiler proves that no 32-bit
//     limb and no 64-bit column sum can overflow anywhere in the three DFT steps.
// Data in HBM stay 32-byte words (canonical on the wire, < 2m between passes); tile elements in LDS and table
// entries are rr_stride = 10 dwords apart (9 used).  Round 3: every product by a TABLE twiddle - the five inside an
// 8-point DFT, the in-tile twiddles, the boundary table of pass 1 - is a Shoup product (field_rr.hip.hpp rr_mul_shoup: the
// table holds the canonical twiddle and its quotient floor(t R_rr / m), 2 x 10 dwords; 143 multiply-adds and no
// quotient-digit chain against 153); the stepped boundary twiddles of pass 2 and the closing factor of the inverse
// transform stay Montgomery products (their tables hold t R_rr mod m).
#pragma once
#include "field_rr.hip.hpp"
#include "ntt_engine.hpp"

namespace blz {

constexpr int NTT_RR_COLS_LOG = 2;   // columns per tile of k_ntt512_rr (256 lanes: 64 rows x 4 columns)

template <class T> struct rr_bounds;
template <class Q, int F, int V>
struct rr_bounds<Frr<Q, F, V>> {
    static constexpr int f = F, v = V;
};

// smallest J >= 1 with 2^(J-1) >= vb: the multiple 2^J m dominates any normalised b < vb m
constexpr int rr_j_norm(int vb) {
    int j = 1;
    while ((1 << (j - 1)) < vb) ++j;
    return j;
}
// smallest J >= 1 with (fb + 1) 2^J >= vb + 1: (fb + 1) x (2^J m in borrow form) dominates a lazy b limb by limb
constexpr int rr_j_lazy(int fb, int vb) {
    int j = 1;
    while ((fb + 1) * (1 << j) < vb + 1) ++j;
    return j;
}

template <class A, class B>
struct RRPair {
    A s;  // u + v
    B d;  // u - v + (multiple of m)
};

// butterfly against a normalised v
template <class Q, int Fu, int Vu, int Vv>
BLZ_DEV auto rr_bfly(const Frr<Q, Fu, Vu>& u, const Frr<Q, 1, Vv>& v) {
    constexpr int J = rr_j_norm(Vv);
    static_assert(J <= Q::NKM, "no multiple of m that large");
    RRPair<Frr<Q, Fu + 1, Vu + Vv>, Frr<Q, Fu + 2, Vu + (1 << J)>> r;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) {
        r.s.v[i] = u.v[i] + v.v[i];
        r.d.v[i] = u.v[i] + (Q::KM[J - 1][i] - v.v[i]);
    }
    return r;
}
// butterfly against a lazy v (limbs < Fv 2^B).  Where the limb slack allows it the constant is (Fv + 1) x the borrow
// form, whose limbs are >= (Fv + 1)(2^B - 1) >= Fv 2^B - 1 (10 x 27 bits); otherwise (9 x 29 bits) v is
// carry-propagated first.
template <class Q, int Fu, int Fv>
constexpr bool rr_bfly_lazy_fits() {
    return ((unsigned long long)(Fv + 1) << (Q::B + 1)) <= (1ull << 32) && Fu + 2 * (Fv + 1) < (1 << (32 - Q::B));
}
template <class Q, int Fu, int Vu, int Fv, int Vv, std::enable_if_t<(Fv >= 2), int> = 0>
BLZ_DEV auto rr_bfly(const Frr<Q, Fu, Vu>& u, const Frr<Q, Fv, Vv>& v) {
    if constexpr (rr_bfly_lazy_fits<Q, Fu, Fv>()) {
        constexpr int K = Fv + 1, J = rr_j_lazy(Fv, Vv);
        static_assert(J <= Q::NKM, "no multiple of m that large");
        RRPair<Frr<Q, Fu + Fv, Vu + Vv>, Frr<Q, Fu + 2 * K, Vu + K * (1 << J)>> r;
#pragma unroll
        for (int i = 0; i < Q::NL; ++i) {
            r.s.v[i] = u.v[i] + v.v[i];
            r.d.v[i] = u.v[i] + ((uint32_t)K * Q::KM[J - 1][i] - v.v[i]);
        }
        return r;
    } else {
        return rr_bfly(u, rr_norm(v));
    }
}

// a product whose lazy operand is carry-propagated first if the 64-bit column sums need it
template <class Q, int Fa, int Va, int Fb, int Vb>
BLZ_DEV void rr_mul_n(Frr<Q, 1, 2>& r, const Frr<Q, Fa, Va>& a, const Frr<Q, Fb, Vb>& b) {
    if constexpr (rr_cols_ok<Q>(Fa * Fb)) rr_mul(r, a, b);
    else rr_mul(r, rr_norm(a), b);
}
// ... and by a table twiddle in Shoup form (field_rr.hip.hpp rr_mul_shoup): the plain product, no Montgomery factor
template <class Q, int Fa, int Va>
BLZ_DEV void rr_mul_n(Frr<Q, 1, 2>& r, const Frr<Q, Fa, Va>& a, const RRShoup<Q>& t) {
    if constexpr (rr_cols_ok<Q>(Fa)) rr_mul_shoup(r, a, t);
    else rr_mul_shoup(r, rr_norm(a), t);
}
template <class Q, int Fa, int Va>
BLZ_DEV void rr_mul_n(Frr<Q, 1, 2>& r, const Frr<Q, Fa, Va>& a, const RRShoupU<Q>& t) {
    if constexpr (rr_cols_ok<Q>(Fa)) rr_mul_shoup(r, a, t);
    else rr_mul_shoup(r, rr_norm(a), t);
}
// Shoup table entries: w | wq, rr_stride dwords each
template <class Q>
constexpr int rr_shoup_stride() { return 2 * rr_stride<Q>(); }
template <class Q>
BLZ_DEV void rr_load_shoup(RRShoup<Q>& t, const uint32_t* p) {
    Frr<Q, 1, 1> a, b;
    rr_load(a, p);
    rr_load(b, p + rr_stride<Q>());
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) {
        t.w[i] = a.v[i];
        t.wq[i] = b.v[i];
    }
}
template <class Q>
BLZ_DEV void rr_store_shoup(uint32_t* p, const RRShoup<Q>& t) {
    Frr<Q, 1, 1> a, b;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) {
        a.v[i] = t.w[i];
        b.v[i] = t.wq[i];
    }
    rr_store(p, a);
    rr_store(p + rr_stride<Q>(), b);
}
// x Rrr (Montgomery, < 2m) -> the plain canonical x with its Shoup quotient
template <class Q>
BLZ_DEV void rr_shoup_from_mont(RRShoup<Q>& t, const Frr<Q, 1, 2>& xm) {
    Frr<Q, 1, 1> one;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) one.v[i] = i == 0 ? 1u : 0u;
    Frr<Q, 1, 2> x;
    rr_mul(x, xm, one);   // x Rrr * 1 / Rrr
    rr_shoup_quot<Q>(t, rr_canon(x));
}

// a value < 2m (normalised) -> canonical
template <class Q>
BLZ_DEV Frr<Q, 1, 1> rr_canon(const Frr<Q, 1, 2>& a) {
    Frr<Q, 1, 1> r;
    uint32_t d[Q::NL], borrow = 0;
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) {
        const uint32_t t = a.v[i] - Q::MOD[i] - borrow;
        borrow = t >> 31;              // limbs are < 2^B <= 2^31: a negative difference has bit 31 set
        d[i] = t & Q::MASK;
    }
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) r.v[i] = borrow ? a.v[i] : d[i];
    return r;
}

template <class T0, class T1, class T2, class T3, class T4, class T5, class T6, class T7>
struct RROct {
    T0 x0; T1 x1; T2 x2; T3 x3; T4 x4; T5 x5; T6 x6; T7 x7;
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    static constexpr int maxv = cmax(cmax(cmax(rr_bounds<T0>::v, rr_bounds<T1>::v), cmax(rr_bounds<T2>::v, rr_bounds<T3>::v)),
                                     cmax(cmax(rr_bounds<T4>::v, rr_bounds<T5>::v), cmax(rr_bounds<T6>::v, rr_bounds<T7>::v)));
};
template <class T0, class T1, class T2, class T3, class T4, class T5, class T6, class T7>
BLZ_DEV RROct<T0, T1, T2, T3, T4, T5, T6, T7> rr_oct(const T0& a, const T1& b, const T2& c, const T3& d, const T4& e, const T5& f,
                                                     const T6& g, const T7& h) {
    return {a, b, c, d, e, f, g, h};
}

// in-register 8-point DFT, decimation in time: a[] holds x[0],x[4],x[2],x[6],x[1],x[5],x[3],x[7] (normalised, value
// < VIN m); outputs in natural order, lazy.  w1, w2, w3 = w8, w8^2, w8^3 (Montgomery, R_rr; canonical).
template <class Q, int VIN, class W>
BLZ_DEV auto dft8_rr(const Frr<Q, 1, VIN> (&a)[8], const W& w1, const W& w2, const W& w3) {
    const auto p01 = rr_bfly(a[0], a[1]);
    const auto p23 = rr_bfly(a[2], a[3]);
    const auto p45 = rr_bfly(a[4], a[5]);
    const auto p67 = rr_bfly(a[6], a[7]);
    Frr<Q, 1, 2> m3, m7;
    rr_mul_n(m3, p23.d, w2);
    rr_mul_n(m7, p67.d, w2);
    const auto q02 = rr_bfly(p01.s, p23.s);
    const auto q13 = rr_bfly(p01.d, m3);
    const auto q46 = rr_bfly(p45.s, p67.s);
    const auto q57 = rr_bfly(p45.d, m7);
    Frr<Q, 1, 2> m5, m6, m7b;
    rr_mul_n(m5, q57.s, w1);
    rr_mul_n(m6, q46.d, w2);
    rr_mul_n(m7b, q57.d, w3);
    const auto r04 = rr_bfly(q02.s, q46.s);
    const auto r15 = rr_bfly(q13.s, m5);
    const auto r26 = rr_bfly(q02.d, m6);
    const auto r37 = rr_bfly(q13.d, m7b);
    return rr_oct(r04.s, r15.s, r26.s, r37.s, r04.d, r15.d, r26.d, r37.d);
}

// expands STMT for the 8 outputs of an RROct: X names the member, K its index
// (variadic: the statement may contain template argument lists, whose commas the preprocessor would split on)
#define BLZ_RR_FOR8(o, ...)                                                                                       \
    { constexpr int K = 0; auto& X = (o).x0; __VA_ARGS__ } { constexpr int K = 1; auto& X = (o).x1; __VA_ARGS__ } \
    { constexpr int K = 2; auto& X = (o).x2; __VA_ARGS__ } { constexpr int K = 3; auto& X = (o).x3; __VA_ARGS__ } \
    { constexpr int K = 4; auto& X = (o).x4; __VA_ARGS__ } { constexpr int K = 5; auto& X = (o).x5; __VA_ARGS__ } \
    { constexpr int K = 6; auto& X = (o).x6; __VA_ARGS__ } { constexpr int K = 7; auto& X = (o).x7; __VA_ARGS__ }

// tile elements are rr_stride dwords apart: 8-byte LDS accesses (40-byte elements are not 16-byte aligned)
template <class Q, int F, int V>
BLZ_DEV void rr_lds_load(Frr<Q, F, V>& r, const uint32_t* lds, uint32_t dw) {
    const uint2* q = reinterpret_cast<const uint2*>(lds + dw);
#pragma unroll
    for (int i = 0; i < Q::NL / 2; ++i) {
        uint2 x = q[i];
        r.v[2 * i] = x.x;
        r.v[2 * i + 1] = x.y;
    }
    if constexpr (Q::NL % 2 == 1) r.v[Q::NL - 1] = lds[dw + Q::NL - 1];
}
template <class Q, int F, int V>
BLZ_DEV void rr_lds_store(uint32_t* lds, uint32_t dw, const Frr<Q, F, V>& a) {
    uint2* q = reinterpret_cast<uint2*>(lds + dw);
#pragma unroll
    for (int i = 0; i < Q::NL / 2; ++i) q[i] = make_uint2(a.v[2 * i], a.v[2 * i + 1]);
    if constexpr (Q::NL % 2 == 1) lds[dw + Q::NL - 1] = a.v[Q::NL - 1];
}

// out[j] = in[j] (32-bit Montgomery, R32) re-expressed in the reduced radix (Montgomery, R_rr), canonical
template <class Fr>
__global__ void k_ntt_table_to_rr(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int count) {
    using Q = typename Fr::RR;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fp<Fr> w;
    fp_load(w, in + (size_t)j * 8);
    Frr<Q, 1, 2> r;
    rr_from_mont32_words<Q>(r, w.v);
    rr_store(out + (size_t)j * rr_stride<Q>(), rr_canon(r));
}
// out[j] = in[j] as a Shoup table entry (canonical value | its quotient): the in-tile and DFT-internal twiddles
template <class Fr>
__global__ void k_ntt_table_to_shoup(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int count) {
    using Q = typename Fr::RR;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fp<Fr> w;
    fp_load(w, in + (size_t)j * 8);
    Frr<Q, 1, 2> r;
    rr_from_mont32_words<Q>(r, w.v);
    RRShoup<Q> t;
    rr_shoup_from_mont<Q>(t, r);
    rr_store_shoup<Q>(out + (size_t)j * rr_shoup_stride<Q>(), t);
}
// fin = n^-1 in Montgomery R_rr form (from the 32-bit ninv); forward transforms have no closing factor
template <class Fr>
__global__ void k_ntt_fin_rr(const uint32_t* __restrict__ ninv32, uint32_t* __restrict__ out) {
    using Q = typename Fr::RR;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Frr<Q, 1, 2> r;
    if (ninv32) {
        Fp<Fr> w;
        fp_load(w, ninv32);
        rr_from_mont32_words<Q>(r, w.v);
    } else {
        rr_one(r);
    }
    rr_store(out, rr_canon(r));
}

// out[j] = w^(j * mult) in the reduced radix, straight from the exponent (the boundary table tA)
template <class Fr, bool SHOUP = false>
__global__ void k_ntt_table_rr_pow(uint32_t* __restrict__ out, uint32_t count, const uint32_t* __restrict__ wbase, uint64_t mult) {
    using Q = typename Fr::RR;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fp<Fr> w, acc;
    fp_load(w, wbase);   // the transform's root (ntt_impl.hip.hpp k_ntt_root)
    const uint64_t e = (uint64_t)j * mult;
    fp_one(acc);
    for (int b = 63; b >= 0; --b) {
        fp_sqr(acc, acc);
        if ((e >> b) & 1) fp_mul(acc, acc, w);
    }
    Frr<Q, 1, 2> r;
    rr_from_mont32_words<Q>(r, acc.v);
    if constexpr (SHOUP) {
        RRShoup<Q> t;
        rr_shoup_from_mont<Q>(t, r);
        rr_store_shoup<Q>(out + (size_t)j * rr_shoup_stride<Q>(), t);
    } else {
        rr_store(out + (size_t)j * rr_stride<Q>(), rr_canon(r));
    }
}

// w^e for e < 2^27 from the three 512-entry tables
template <class Q>
BLZ_DEV void tw_pow_rr(Frr<Q, 1, 2>& r, const NttTablesRR& T, uint32_t e) {
    constexpr uint32_t ES = rr_stride<Q>();
    Frr<Q, 1, 1> a;
    rr_load(r, T.t0 + (size_t)(e & 511u) * ES);
    const uint32_t e1 = (e >> 9) & 511u, e2 = e >> 18;
    if (e1) { rr_load(a, T.t1 + (size_t)e1 * ES); rr_mul(r, r, a); }
    if (e2) { rr_load(a, T.t2 + (size_t)e2 * ES); rr_mul(r, r, a); }
}

// tB: pass 2's boundary factor w^((C k1 + k2) i0) (with the column part of pass 1's, see k_ntt512_rr) of every element,
// canonical Montgomery form packed in 32 bytes, IN THE ORDER PASS 2 CONSUMES THEM: entry ((tile 8 + K) 256 + thread) is the
// factor of output K of that lane of that tile, so a wave reads 2 KiB in one piece per output and the table streams through once
// (at the element's own index - rows 16 KiB apart, 128 bytes each, like the data - the pass gained 8 % instead of 14 %)
template <class Fr>
__global__ void k_ntt_table_b(uint32_t* __restrict__ out, NttGeom g, NttTablesRR T) {
    using Q = typename Fr::RR;
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >> g.logn) return;
    const uint32_t thread = (uint32_t)(idx & 255u), K = (uint32_t)((idx >> 8) & 7u);
    const uint64_t tile = idx >> 11;
    const uint64_t tiles_per = (1ull << g.logA) >> NTT_RR_COLS_LOG;
    const uint32_t k2 = (uint32_t)(tile / tiles_per);
    const uint32_t i0 = (uint32_t)((tile % tiles_per) << NTT_RR_COLS_LOG) + (thread & ((1u << NTT_RR_COLS_LOG) - 1u));
    const uint32_t n2 = thread >> NTT_RR_COLS_LOG;
    const uint32_t k1 = (n2 >> 3) + 8u * (n2 & 7u) + 64u * K;   // the lane's output row (k_ntt512_rr: kb + 64 K)
    Frr<Q, 1, 2> w;
    tw_pow_rr<Q>(w, T, ((k1 << g.logC) + k2) * i0);
    Fp<Fr> y;
    rr_to_words<Q>(y.v, rr_canon(w));
    fp_store(out + idx * 8, y);
}

// the boundary table tA (2^18 entries, read once per element after pass 1) as Shoup entries (20 MiB) or Montgomery ones
// (10 MiB, -DBLZ_NTT_TA_SHOUP=0).  Same-box: pass 1 5.48 ms with Shoup entries, 5.59 with Montgomery ones - and 5.49 before
// any product of the pass was a Shoup product: pass 1 is bound by its access pattern (rows 8 MiB apart), not by its products.
#ifndef BLZ_NTT_TA_SHOUP
#define BLZ_NTT_TA_SHOUP 1
#endif
constexpr bool NTT_TA_SHOUP = BLZ_NTT_TA_SHOUP != 0;

// Experiment of round 6 (profiles/r06_ntt_sq.txt): non-temporal loads / stores of the passes' data, a bit per (pass, direction):
// bit 2 (p - 1) = pass p's loads, bit 2 (p - 1) + 1 = its stores.  Shipped: 0.
#ifndef BLZ_NTT_NT
#define BLZ_NTT_NT 0
#endif
// Second experiment of round 6: four blocks per CU (4 waves per SIMD: 128 registers per lane, the LDS's 4 x 40 KiB) for the passes
// whose bit is set (bit p - 1 = pass p).  Shipped: 0 (three blocks; the kernels hold 143 - 166 VGPRs).
#ifndef BLZ_NTT_OCC4
#define BLZ_NTT_OCC4 0
#endif
constexpr int NR_COLS_LOG = NTT_RR_COLS_LOG;
constexpr int NR_COLS = 1 << NR_COLS_LOG;
constexpr int NR_THREADS = 64 * NR_COLS;

// The tile goes through the LDS in two halves, so that a block needs 40 KiB instead of 80 and THREE blocks share a CU
// (the registers allow three waves per SIMD; with a whole tile in the LDS two blocks are all that fit, and the waves then
// wait ~30 % of their time at the barriers and the tile loads with nobody to take the multiplier).  Both exchanges move
// outputs K = 0..3 first and K = 4..7 second: the consumers of an output K of the first exchange are the lanes with
// k1 = K - waves 0 and 1 for the first half, waves 2 and 3 for the second - and of the second exchange the lanes with
// k1' = K of the same wave.  Costs: three more block barriers and four outputs kept in registers across a half.
template <class Fr, int PASS, bool TABB = false>   // TABB: pass 2 reads its boundary factors from the per-element table tB
__global__ __launch_bounds__(NR_THREADS, ((BLZ_NTT_OCC4 >> (PASS - 1)) & 1) ? 4 : 3) void k_ntt512_rr(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, NttGeom g,
                                                            NttTablesRR T) {
    using Q = typename Fr::RR;
    constexpr uint32_t ES = rr_stride<Q>();  // element stride in LDS and in the tables (dwords)
    constexpr uint32_t RS = NR_COLS * ES;    // tile row stride in dwords (40: the 32-bit kernel's 4 x 8 + 8)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t A = 1u << g.logA, B = 1u << g.logB, C = 1u << g.logC;
    uint64_t col_base, fixed, in_base, in_rstride, in_cstride;
    const uint64_t tile = blockIdx.x;
    const uint32_t col = threadIdx.x & (NR_COLS - 1), n2 = threadIdx.x >> NR_COLS_LOG;
    if (PASS == 1) {   // rows i2 (stride AB), cols i0 (stride 1), fixed i1
        const uint64_t tiles_per = A >> NR_COLS_LOG;
        if (T.swz) {
            // tiles in flight together differ in the low 7 bits of i1 first (16 KiB apart) and only then in the column
            // group (128 B apart): a tile's 512 rows are 8 MiB apart, so neighbours in the column alone would put the
            // whole chip on a 64 KiB window of every row - a few HBM channels - at any moment (pass 1: 6.9 -> 6.4 ms).
            // swz = 1 + s: the lowest s bits of the tile number pick the column group instead, so that 2^s tiles dispatched
            // back to back read ADJACENT 128-byte pieces of every row (one DRAM row activation serves them): pass 1 5.46 -> 5.19
            // ms at s = 3 (15.34 -> 15.15 ms per transform, same box; s = 2: 15.20, s = 4: 15.45, s = 5: 15.8).
            const uint32_t sb = (T.swz & 15u) - 1u;               // 0 .. 5
            const uint32_t ib = (T.swz >> 4) ? (T.swz >> 4) : 7u; // bits of i1 walked before the next column groups
            const uint64_t lowcol = tile & ((1u << sb) - 1u);
            const uint64_t t2 = tile >> sb;
            fixed = (t2 & ((1u << ib) - 1u)) | ((tile >> (ib + 7u)) << ib);
            col_base = (lowcol | (((t2 >> ib) & (127u >> sb)) << sb)) << NR_COLS_LOG;
        } else {
            fixed = tile / tiles_per;
            col_base = (tile % tiles_per) << NR_COLS_LOG;
        }
        in_base = col_base + (uint64_t)A * fixed;
        in_rstride = (uint64_t)A * B;
        in_cstride = 1;
    } else if (PASS == 2) {  // rows i1 (stride A), cols i0, fixed k2
        const uint64_t tiles_per = A >> NR_COLS_LOG;
        fixed = tile / tiles_per;
        col_base = (tile % tiles_per) << NR_COLS_LOG;
        in_base = col_base + (uint64_t)A * B * fixed;
        in_rstride = A;
        in_cstride = 1;
    } else {  // rows i0 (stride 1, contiguous), cols k2 (stride AB), fixed k1
        const uint64_t tiles_per = C >> NR_COLS_LOG;
        fixed = tile / tiles_per;
        col_base = (tile % tiles_per) << NR_COLS_LOG;
        in_base = (uint64_t)A * fixed + (uint64_t)A * B * col_base;
        in_rstride = 1;
        in_cstride = (uint64_t)A * B;
    }
    const uint32_t* wp = T.wpass[PASS - 1];  // w512^j, j < 512, Shoup entries (canonical value | quotient)
    constexpr uint32_t ES2 = rr_shoup_stride<Q>();
    using WT = RRShoup<Q>;     // table twiddles: Shoup products (field_rr.hip.hpp): 143 multiply-adds, no Montgomery factor
    using W = Frr<Q, 1, 2>;    // stepped twiddles (Montgomery form) and data
    RRShoupU<Q> w1, w2, w3;   // the w8 powers inside the 8-point DFTs: the same for every lane, kept in SGPRs
    {
        WT t;
        rr_load_shoup<Q>(t, wp + 64 * ES2);
        rr_shoup_uniform<Q>(w1, t);
        rr_load_shoup<Q>(t, wp + 128 * ES2);
        rr_shoup_uniform<Q>(w2, t);
        rr_load_shoup<Q>(t, wp + 192 * ES2);
        rr_shoup_uniform<Q>(w3, t);
    }

    // ---- step 1: 8-point DFTs over n1 (rows 64 n1 + n2), straight from global memory.  The words on the wire are
    // any 256-bit value (canonical on the wire by contract; a stray one is still reduced correctly); between passes
    // they are < 2m.
    constexpr int BR[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    constexpr int VIN1 = 1 << (32 * Q::N32 + 1 - Q::BITS);   // 2^256 < VIN1 m: 4 (BLS12-381), 8 (BN254), 16 (BLS12-377)
    Frr<Q, 1, VIN1> a1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t row = 64u * BR[j] + n2;
        Fp<Fr> x;
        uint64_t iaddr = in_base + row * in_rstride + col * in_cstride;
        if (PASS == g.wire_pass && g.brin) iaddr = __brevll(iaddr) >> (64 - g.logn);   // the caller's buffer is in bit-reversed order
        if constexpr (((BLZ_NTT_NT >> (2 * (PASS - 1))) & 1) != 0) fp_load_nt(x, in + iaddr * 8);
        else fp_load(x, in + iaddr * 8);
        rr_from_words<Q>(a1[j], x.v);
    }
    auto o1 = dft8_rr<Q>(a1, w1, w2, w3);
    const uint32_t k1 = n2 >> 3, n2p = n2 & 7u;
    W a2[8];
    {
        // first half: outputs K = 0..3 at rows 64 K + n2, read by the lanes with k1 < 4 (waves 0, 1); second half:
        // K = 4..7 at rows 64 (K - 4) + n2, read by waves 2, 3
        W keep[4];
        BLZ_RR_FOR8(o1, {
            W t;
            if constexpr (K == 0) {
                t = rr_reduce2m(X);
            } else {
                WT w;
                rr_load_shoup<Q>(w, wp + (size_t)(n2 * K) * ES2);
                rr_mul_n(t, X, w);
            }
            if constexpr (K < 4) rr_lds_store(lds, (64u * K + n2) * RS + col * ES, t);
            else keep[K - 4] = t;
        })
        __syncthreads();
        if (k1 < 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) rr_lds_load(a2[j], lds, (64u * k1 + 8u * BR[j] + n2p) * RS + col * ES);
        }
        __syncthreads();
#pragma unroll
        for (int K = 4; K < 8; ++K) rr_lds_store(lds, (64u * (K - 4) + n2) * RS + col * ES, keep[K - 4]);
        __syncthreads();
        if (k1 >= 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) rr_lds_load(a2[j], lds, (64u * (k1 - 4u) + 8u * BR[j] + n2p) * RS + col * ES);
        }
        __syncthreads();   // the second exchange reuses the rows
    }
    auto o2 = dft8_rr<Q>(a2, w1, w2, w3);
    const uint32_t k1p = n2 & 7u;
    W a3[8];
    {
        // a wave's 64 rows of the half tile: (k1 & 1) 32 + 8 (K mod 4) + n2'; readers k1' < 4 first, then k1' >= 4
        const uint32_t wrow = (threadIdx.x >> 6) * 64u + (k1 & 1u) * 32u;
        W keep[4];
        BLZ_RR_FOR8(o2, {
            W t;
            if constexpr (K == 0) {
                t = rr_reduce2m(X);
            } else {
                WT w;
                rr_load_shoup<Q>(w, wp + (size_t)(8u * n2p * K) * ES2);
                rr_mul_n(t, X, w);
            }
            if constexpr (K < 4) rr_lds_store(lds, (wrow + 8u * K + n2p) * RS + col * ES, t);
            else keep[K - 4] = t;
        })
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (k1p < 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) rr_lds_load(a3[j], lds, (wrow + 8u * k1p + BR[j]) * RS + col * ES);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int K = 4; K < 8; ++K) rr_lds_store(lds, (wrow + 8u * (K - 4) + n2p) * RS + col * ES, keep[K - 4]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (k1p >= 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) rr_lds_load(a3[j], lds, (wrow + 8u * (k1p - 4u) + BR[j]) * RS + col * ES);
        }
    }
    // ---- step 2b: lane (k1, k1', col): 8-point DFTs over n2' (rows 64 k1 + 8 k1' + n2'); outputs k = k1 + 8 k1' + 64 k2'
    // leave with the inter-pass twiddle (passes 1, 2) or the closing factor (pass 3), which also brings them back
    // below 2m for the 32-byte word
    auto o3 = dft8_rr<Q>(a3, w1, w2, w3);
    const uint32_t kb = k1 + 8u * k1p;  // output row of x_(k2') is kb + 64 k2'
    // Boundary factors.  Between passes 1 and 2 every element owes w^(k2 (i0 + A i1)), between 2 and 3 w^(C i0 k1).
    // With the table tA (2^27 transforms) the first one is split: its column-independent part w^(A i1 k2) is READ
    // after pass 1 (index k2 i1 < 2^18: no stepping chain, 11 products less per lane), and w^(i0 k2), which does
    // not involve the index pass 2 transforms over, joins pass 2's own factor: w^(i0 (C k1 + k2)), still one
    // geometric sequence along the lane's rows.
    const bool split = T.tA != nullptr;
    constexpr bool tabB = PASS == 2 && TABB;
    W w, step;
    WT step_s;
    if (PASS == 1) {
        if (!split) {
            // x(i0, i1, k2 = row) *= w^(row (i0 + A i1)),  m = i0 + A i1 < 2^18
            const uint64_t m = col_base + col + ((uint64_t)fixed << g.logA);
            tw_pow_rr<Q>(w, T, (uint32_t)(kb * m));
            tw_pow_rr<Q>(step, T, (uint32_t)(64u * m));
        }
    } else if (PASS == 2) {
        // x(i0, k1 = row, k2) *= w^(C i0 row)   [split: * w^(i0 k2) as well; k2 = fixed]
        // With the table tB (2^27 transforms) the factor of every element is READ at the element's own index: 8 products per
        // lane instead of 8 + 7 (the stepping chain) + 2 (its first term from the 512-entry tables), for 4 GiB more traffic
        // on a pass the multiplier bounds.
        if constexpr (!tabB) {
            const uint64_t i0 = col_base + col;
            tw_pow_rr<Q>(w, T, (uint32_t)((((uint64_t)kb << g.logC) + (split ? fixed : 0)) * i0));
            rr_load_shoup<Q>(step_s, T.ts2 + (size_t)i0 * ES2);   // w^(64 C i0): a Shoup entry - w (Montgomery form) times a plain constant stays in Montgomery form
        }
    } else if (T.fin) {
        rr_load(w, T.fin);   // inverse transform: n^-1; a forward transform closes with the product-free reduction
    }
    BLZ_RR_FOR8(o3, {
        const uint32_t row = kb + 64u * K;
        W t;
        if (PASS == 3 && !T.fin) {
            t = rr_reduce2m(X);
        } else if (PASS == 1 && split) {   // w^(A i1 k2), read from the boundary table
            if constexpr (NTT_TA_SHOUP) {
                WT wa;
                rr_load_shoup<Q>(wa, T.tA + (size_t)(row * (uint32_t)fixed) * ES2);
                rr_mul_n(t, X, wa);
            } else {
                Frr<Q, 1, 1> wa;
                rr_load(wa, T.tA + (size_t)(row * (uint32_t)fixed) * ES);
                rr_mul_n(t, X, wa);
            }
        } else if constexpr (tabB) {
            Fp<Fr> tw;
            fp_load(tw, T.tB + (((uint64_t)tile * 8u + K) * NR_THREADS + threadIdx.x) * 8);
            Frr<Q, 1, 1> wa;
            rr_from_words<Q>(wa, tw.v);
            rr_mul_n(t, X, wa);
        } else {
            rr_mul_n(t, X, w);
        }
        Fp<Fr> y;
        rr_to_words<Q>(y.v, t);
        uint64_t oaddr;
        if (PASS == 3) {
            fp_csub_const<Fr, Fr::MOD>(y);   // < 2m -> canonical: the wire format
            oaddr = (col_base + col) + (uint64_t)C * fixed + (uint64_t)C * B * row;
            if (g.brout) oaddr = __brevll(oaddr) >> (64 - g.logn);
        } else {
            if (K != 7 && !(PASS == 1 && split) && !(PASS == 2 && tabB)) {   // twiddle x twiddle
                if (PASS == 2) rr_mul_shoup(w, w, step_s);
                else rr_mul(w, w, step);
            }
            oaddr = in_base + row * in_rstride + col;
        }
        if constexpr (((BLZ_NTT_NT >> (2 * (PASS - 1) + 1)) & 1) != 0) fp_store_nt(out + oaddr * 8, y);
        else fp_store(out + oaddr * 8, y);
    })
}

}  // namespace blz
