// Row-cooperative XYZZ group law for the one strictly sequential chain of an MSM: k_finish's Horner walk over the windows -
// ~255 dependent doublings, with a single point in flight.  The quad-cooperative law (ec_quad.hip.hpp) gives that chain one
// lane per PRODUCT of a formula round; every lane still runs a whole 14 x 14-limb Montgomery product on its own, ~470
// dependent instructions, and a doubling took ~5 us whatever the chip had free.  Here the 16 lanes of a DPP row hold ONE field
// element, a limb per lane (14 limbs of 28 bits: the loose reduced-radix budget of field_rr.hip.hpp, same Montgomery constant
// R = 2^392, so values move between the two forms limb for limb), and the four rows of the wave compute the four products of
// a formula round at the same time: a product is 14 rounds of (broadcast a limb of b, multiply-add, broadcast the low word of
// position 0, quotient digit, multiply-add, shift the accumulators down one lane) - ~190 instructions for all four products.
// Values are replicated over the four rows between rounds (a point is four registers); a round selects each row's operands by
// its row number and broadcasts each row's product to all rows with ds_bpermute.
//
// Bounds (per element): a product's operands have limbs < F 2^28 with (Fa Fb + 1) 14 + 1 <= 2^8 - the 64-bit accumulator of a
// position collects 14 caller products and 14 reduction products - and values Va Vb <= 2^11 (R / m); its result is < 2m with
// WEAKLY normalised limbs (<= 2^28: three carry rounds across the lanes; a last 0 / 1 carry may be pending).  Differences use
// the borrow-form multiples of m DOUBLED (2 KM[J-2] = 2^J m with every limb >= 2^29 - 2), which dominate a weakly normalised
// subtrahend limb by limb; the exact form - the pending carries resolved at once from generate / propagate ballots - is only
// needed by the zero test and the export (a ballot costs a VALU -> SALU -> VALU round trip on the one chain there is).
// The formulas are ec_quad.hip.hpp's quadrr_dbl / quadrr_add operation for operation (same multiples of m in the same places),
// except that Y3 = t - W Y + 4m stays the lazy < 6m it is (no one-digit quotient reduction); rowpt_export brings it below 2m
// for the 32-bit form.
#pragma once
#include "ec_rr.hip.hpp"

namespace blz {

template <class Q>
struct RowCtx {
    uint32_t li, row;      // limb index inside the row (0..15), row of the wave (0..3)
    uint32_t mod, one;     // this lane's limb of m and of R mod m (0 beyond the element's limbs)
    uint32_t km2, km5;     // ... of 4m and of 32m: twice the borrow forms of 2m and 16m (2 Q::KM[0], 2 Q::KM[3])
    uint32_t baddr[4];     // ds_bpermute byte address of this lane's limb in row K
};
template <class Q>
BLZ_DEV RowCtx<Q> row_ctx() {
    static_assert(Q::NL <= 14 && Q::B == 28 && !RR_TIGHT<Q>, "row arithmetic is written for the loose 28-bit budget");
    RowCtx<Q> c;
    const uint32_t lane = threadIdx.x & 63u;
    c.li = lane & 15u;
    c.row = lane >> 4;
    const bool in = c.li < (uint32_t)Q::NL;
    const uint32_t i = in ? c.li : 0u;
    c.mod = in ? Q::MOD[i] : 0u;
    c.one = in ? Q::ONE[i] : 0u;
    c.km2 = in ? 2u * Q::KM[0][i] : 0u;
    c.km5 = in ? 2u * Q::KM[3][i] : 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) c.baddr[k] = ((uint32_t)k * 16u + c.li) * 4u;
    return c;
}

// lane K of my row
template <int K>
BLZ_DEV uint32_t row_lane(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x150 + K, 0xf, 0xf, true);   // row_newbcast:K
}
BLZ_DEV uint32_t row_from_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }   // lane i <- i + 1 (row_shl:1), 0 into lane 15
BLZ_DEV uint32_t row_from_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); }   // lane i <- i - 1 (row_shr:1), 0 into lane 0
// row K's value on every row
template <int K, class Q>
BLZ_DEV uint32_t row_bcast(const RowCtx<Q>& c, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)c.baddr[K], (int)v); }
template <class Q>
BLZ_DEV uint32_t row_sel(const RowCtx<Q>& c, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3) {
    const uint32_t lo = c.row & 1u ? a1 : a0, hi = c.row & 1u ? a3 : a2;
    return c.row & 2u ? hi : lo;
}

// limbs in [0, 2^28] -> exactly normalised: the 0 / 1 carries left are resolved from the generate / propagate ballots (a row's
// lanes 14, 15 neither generate nor propagate, so a chain never leaves its row)
template <class Q>
BLZ_DEV uint32_t row_resolve(const RowCtx<Q>& c, uint32_t x) {
    const uint32_t low = x & Q::MASK;
    const bool top = c.li == (uint32_t)Q::NL - 1u;    // the top limb keeps what is left (the value bound keeps it small)
    const uint64_t G = __ballot((x >> Q::B) != 0u && !top);
    const uint64_t P = __ballot(low == Q::MASK && !top && c.li < (uint32_t)Q::NL);
    const uint64_t cin = (((G << 1) + P) ^ P);
    const uint32_t ci = (uint32_t)(cin >> (threadIdx.x & 63u)) & 1u;
    return top ? x + ci : (low + ci) & Q::MASK;
}
// lazy limbs (< 2^32) -> weakly normalised (<= 2^28)
template <class Q>
BLZ_DEV uint32_t row_norm(const RowCtx<Q>& c, uint32_t x) {
    const bool top = c.li == (uint32_t)Q::NL - 1u;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t cy = top ? 0u : x >> Q::B;
        x = (top ? x : x & Q::MASK) + row_from_prev(cy);
    }
    return x;
}

// Montgomery product of the row's element pairs (a, b): a b / R mod m, < 2m, weakly normalised
template <class Q, int K>
BLZ_DEV void row_mul_step(const RowCtx<Q>& c, uint64_t& T, uint32_t a, uint32_t b) {
    const uint32_t bk = row_lane<K>(b);
    T = (uint64_t)a * bk + T;
    const uint32_t t0 = row_lane<0>((uint32_t)T);
    const uint32_t q = (t0 * Q::N0) & Q::MASK;
    T = (uint64_t)q * c.mod + T;
    // position 0 is now 0 mod 2^B: everything moves down one position, what is left of position 0 joins the new position 0
    const uint64_t cy = T >> Q::B;
    const uint32_t lo = row_from_next((uint32_t)T), hi = row_from_next((uint32_t)(T >> 32));
    T = (((uint64_t)hi << 32) | lo) + (c.li == 0u ? cy : 0ull);
    if constexpr (K + 1 < Q::NL) row_mul_step<Q, K + 1>(c, T, a, b);
}
template <class Q>
BLZ_DEV uint32_t row_mul(const RowCtx<Q>& c, uint32_t a, uint32_t b) {
    uint64_t T = 0;
    row_mul_step<Q, 0>(c, T, a, b);
    // positions hold < 2^63: three carry rounds leave limbs <= 2^28
    const bool top = c.li == (uint32_t)Q::NL - 1u;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const uint64_t cy = top ? 0ull : T >> Q::B;
        const uint32_t lo = row_from_prev((uint32_t)cy), hi = row_from_prev((uint32_t)(cy >> 32));
        T = (top ? T : T & Q::MASK) + (((uint64_t)hi << 32) | lo);
    }
    return (uint32_t)T;
}

// a point, every coordinate replicated on the four rows
struct RowPt {
    uint32_t x, y, zz, zzz;
};
BLZ_DEV void rowpt_set_inf(RowPt& p) { p.x = p.y = p.zz = p.zzz = 0u; }
BLZ_DEV bool rowpt_is_inf(const RowPt& p) { return __ballot(p.zz != 0u) == 0ull; }
template <class Q>
BLZ_DEV void rowpt_load(const RowCtx<Q>& c, RowPt& p, const uint32_t* base, size_t idx) {
    constexpr int S = rr_stride<Q>();
    const uint32_t* q = base + idx * 4 * S;
    const bool in = c.li < (uint32_t)Q::NL;
    const uint32_t i = in ? c.li : 0u;
    p.x = in ? q[i] : 0u;
    p.y = in ? q[S + i] : 0u;
    p.zz = in ? q[2 * S + i] : 0u;
    p.zzz = in ? q[3 * S + i] : 0u;
}
// ... and back in the same layout, limbs as they are (weakly normalised): read by rowpt_load only - the quad law of
// ec_quad.hip.hpp wants limbs below 2^28 and coordinates below 2m
template <class Q>
BLZ_DEV void rowpt_store(const RowCtx<Q>& c, uint32_t* base, size_t idx, const RowPt& p) {
    constexpr int S = rr_stride<Q>();
    uint32_t* q = base + idx * 4 * S;
    if (c.row == 0u && c.li < (uint32_t)Q::NL) {
        q[c.li] = p.x;
        q[S + c.li] = p.y;
        q[2 * S + c.li] = p.zz;
        q[3 * S + c.li] = p.zzz;
    }
}

// x = 0 (mod m)?  x lazy, value < 2^11 m.  Exact: the product by one is < 2m and normalised - it is 0 or m
template <class Q>
BLZ_DEV bool row_is_zero(const RowCtx<Q>& c, uint32_t x) {
    const uint32_t t = row_resolve<Q>(c, row_mul<Q>(c, x, c.li == 0u ? 1u : 0u));
    const bool in = c.li < (uint32_t)Q::NL;
    return __ballot(in && t != 0u) == 0ull || __ballot(in && t != c.mod) == 0ull;
}
// cheap filter in front of it (field_rr.hip.hpp rr_maybe_equal): can d = a - b + 4m be 0 mod m?  d = k m for some 0 <= k < 8
// then, and k = d_0 m^-1 mod 2^28 whatever carries are pending above the low limb
template <class Q>
BLZ_DEV bool row_maybe_equal(const RowCtx<Q>& c, uint32_t d) {
    const uint32_t k = (row_lane<0>(d) * Q::MINV) & Q::MASK;
    return k < 32u || k > Q::MASK - 32u;
}

// p = 2p   (ec_quad.hip.hpp quadrr_dbl, loose budget)
template <class Q>
BLZ_DEV void rowpt_dbl(const RowCtx<Q>& c, RowPt& p) {
    if (rowpt_is_inf(p)) return;
    const uint32_t U = p.y + p.y;                                     // (2, 12)
    // round 1: V = U^2 | A = X^2
    uint32_t a = row_sel<Q>(c, U, p.x, U, p.x);
    uint32_t r = row_mul<Q>(c, a, a);
    const uint32_t V = row_bcast<0, Q>(c, r), A = row_bcast<1, Q>(c, r);
    const uint32_t M = A + A + A;                                     // (3, 6)
    // round 2: W = U V | S = X V | ZZ3 = V ZZ | MM = M^2
    a = row_sel<Q>(c, U, p.x, V, M);
    uint32_t b = row_sel<Q>(c, V, V, p.zz, M);
    r = row_mul<Q>(c, a, b);
    const uint32_t W = row_bcast<0, Q>(c, r), S = row_bcast<1, Q>(c, r), ZZ3 = row_bcast<2, Q>(c, r), MM = row_bcast<3, Q>(c, r);
    const uint32_t X3 = row_norm<Q>(c, MM + 2u * (c.km2 - S));        // M^2 - 2S + 8m: (1, 10)
    const uint32_t D = S + (c.km5 - X3);                              // S - X3 + 32m: (5, 34); M D: 3 x 5 + 1 = 16 <= 18
    // round 3: t = M D | WY = W Y | ZZZ3 = W ZZZ
    a = row_sel<Q>(c, M, W, W, W);
    b = row_sel<Q>(c, D, p.y, p.zzz, p.zzz);
    r = row_mul<Q>(c, a, b);
    const uint32_t t = row_bcast<0, Q>(c, r), WY = row_bcast<1, Q>(c, r), ZZZ3 = row_bcast<2, Q>(c, r);
    p.x = X3;
    p.y = row_norm<Q>(c, t + (c.km2 - WY));                           // M (S - X3) - W Y + 4m: (1, 6)
    p.zz = ZZ3;
    p.zzz = ZZZ3;
}

// acc += q   (quadrr_add)
template <class Q>
BLZ_DEV void rowpt_add(const RowCtx<Q>& c, RowPt& acc, const RowPt& q) {
    if (rowpt_is_inf(q)) return;
    if (rowpt_is_inf(acc)) { acc = q; return; }
    // round 1: U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1
    uint32_t a = row_sel<Q>(c, acc.x, q.x, acc.y, q.y);
    uint32_t b = row_sel<Q>(c, q.zz, acc.zz, q.zzz, acc.zzz);
    uint32_t r = row_mul<Q>(c, a, b);
    const uint32_t U1 = row_bcast<0, Q>(c, r), U2 = row_bcast<1, Q>(c, r), S1 = row_bcast<2, Q>(c, r), S2 = row_bcast<3, Q>(c, r);
    const uint32_t P = row_norm<Q>(c, U2 + (c.km2 - U1));             // U2 - U1 + 4m: (1, 6)
    const uint32_t R = row_norm<Q>(c, S2 + (c.km2 - S1));             // (1, 6)
    if (__builtin_expect(row_maybe_equal<Q>(c, P), 0)) {
        if (row_is_zero<Q>(c, P)) {   // same x: P + P or P - P
            if (row_is_zero<Q>(c, R)) { acc = q; rowpt_dbl<Q>(c, acc); }
            else rowpt_set_inf(acc);
            return;
        }
    }
    // round 2: PP = P^2 | RR = R^2 | Z12 = ZZ1 ZZ2 | Z123 = ZZZ1 ZZZ2
    a = row_sel<Q>(c, P, R, acc.zz, acc.zzz);
    b = row_sel<Q>(c, P, R, q.zz, q.zzz);
    r = row_mul<Q>(c, a, b);
    const uint32_t PP = row_bcast<0, Q>(c, r), RRv = row_bcast<1, Q>(c, r), Z12 = row_bcast<2, Q>(c, r), Z123 = row_bcast<3, Q>(c, r);
    // round 3: PPP = P PP | Q = U1 PP | ZZ3 = Z12 PP
    a = row_sel<Q>(c, P, U1, Z12, Z12);
    r = row_mul<Q>(c, a, PP);
    const uint32_t PPP = row_bcast<0, Q>(c, r), Qv = row_bcast<1, Q>(c, r), ZZ3 = row_bcast<2, Q>(c, r);
    const uint32_t X3 = row_norm<Q>(c, RRv + (c.km2 - PPP) + 2u * (c.km2 - Qv));   // R^2 - PPP - 2Q + 12m: (1, 14)
    const uint32_t D = row_norm<Q>(c, Qv + (c.km5 - X3));             // (1, 34): R D would be 1 x 5 ... kept normalised like P, R
    // round 4: t = R D | SP = S1 PPP | ZZZ3 = Z123 PPP
    a = row_sel<Q>(c, R, S1, Z123, Z123);
    b = row_sel<Q>(c, D, PPP, PPP, PPP);
    r = row_mul<Q>(c, a, b);
    const uint32_t t = row_bcast<0, Q>(c, r), SP = row_bcast<1, Q>(c, r), ZZZ3 = row_bcast<2, Q>(c, r);
    acc.x = X3;
    acc.y = row_norm<Q>(c, t + (c.km2 - SP));                         // (1, 6)
    acc.zz = ZZ3;
    acc.zzz = ZZZ3;
}

// ... in ec_rr.hip.hpp's accumulator form (exact limbs, Y below 2m by a product with R mod m: XYZZRR's bounds), for readers on
// either law
template <class Q>
BLZ_DEV void rowpt_store_strict(const RowCtx<Q>& c, uint32_t* base, size_t idx, const RowPt& p) {
    constexpr int S = rr_stride<Q>();
    const bool inf = rowpt_is_inf(p);
    const uint32_t y = inf ? 0u : row_mul<Q>(c, p.y, c.one);
    const uint32_t ex = row_resolve<Q>(c, inf ? 0u : p.x), ey = row_resolve<Q>(c, y), ezz = row_resolve<Q>(c, p.zz), ezzz = row_resolve<Q>(c, inf ? 0u : p.zzz);
    uint32_t* q = base + idx * 4 * S;
    if (c.row == 0u && c.li < (uint32_t)Q::NL) {
        q[c.li] = ex;
        q[S + c.li] = ey;
        q[2 * S + c.li] = ezz;
        q[3 * S + c.li] = ezzz;
    }
}

// the point as ec_rr.hip.hpp's accumulator, on lane 0 (via LDS: 4 x 16 dwords).  Y is brought below 2m by a product with R mod m.
template <class Q>
BLZ_DEV void rowpt_export(const RowCtx<Q>& c, const RowPt& p, uint32_t (*sh)[16], XYZZRR<Q>& out) {
    const bool inf = rowpt_is_inf(p);
    const uint32_t y = inf ? 0u : row_mul<Q>(c, p.y, c.one);
    const uint32_t ex = row_resolve<Q>(c, p.x), ey = row_resolve<Q>(c, y), ezz = row_resolve<Q>(c, p.zz), ezzz = row_resolve<Q>(c, p.zzz);
    if (c.row == 0u) {
        sh[0][c.li] = ex;
        sh[1][c.li] = ey;
        sh[2][c.li] = ezz;
        sh[3][c.li] = ezzz;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < Q::NL; ++i) {
        out.x.v[i] = sh[0][i];
        out.y.v[i] = sh[1][i];
        out.zz.v[i] = sh[2][i];
        out.zzz.v[i] = sh[3][i];
    }
}

}  // namespace blz
