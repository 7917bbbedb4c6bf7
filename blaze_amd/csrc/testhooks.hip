// Element-wise kernels over the device field / group primitives, so tests can compare each of them
// with the CPU oracle (include/blaze_hip.h "test hooks").  Not on the MSM/NTT product path.
#include "common.hpp"
#include "ec_rr.hip.hpp"
#include "ec_row.hip.hpp"

namespace blz {

template <class P>
__global__ __launch_bounds__(64) void k_test_field(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, uint32_t n) {
    uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    Fp<P> x, y, r;
    fp_load(x, a + (size_t)i * P::N);
    fp_load(y, b + (size_t)i * P::N);
    fp_to_mont(x, x);
    fp_to_mont(y, y);
    switch (op) {
        case 0: fp_mul(r, x, y); break;
        case 1: fp_add(r, x, y); break;
        case 2: fp_sub(r, x, y); break;
        case 3: fp_inv(r, x); break;
        case 5:  // x y + (x + y)(x - y), unreduced sums as operands
        case 6:  // x y - (x + y)(x - y)
            if constexpr (P::LAZY) {
                Fp<P> s, d;
                fp_add(s, x, y);
                fp_sub(d, x, y);
                if (op == 5) fp_mul2(r, x, y, s, d);
                else fp_mulsub2(r, x, y, s, d);
            } else {
                fp_zero(r);
            }
            break;
        case 10: case 11: case 12: case 13: case 14: case 15:
            // the reduced-radix twin (field_rr.hip.hpp): product, square, fused sum of products on lazy operands,
            // carry propagation, zero tests; operands go in through the wire-word conversion and come back
            // through the 32-bit Montgomery form, so both conversions are under test as well
            if constexpr (USE_RR<P>) {
                using Q = typename P::RR;
                fp_load(x, a + (size_t)i * P::N);
                fp_load(y, b + (size_t)i * P::N);
                Frr<Q, 1, 2> xr, yr, rr;
                rr_to_mont_from_words<Q>(xr, x.v);
                rr_to_mont_from_words<Q>(yr, y.v);
                if (op == 10) rr_mul(rr, xr, yr);
                else if (op == 11) rr_sqr(rr, xr);
                else if (op == 12) rr_mul2(rr, xr, yr, rr_norm(rr_add(xr, yr)), rr_sub<2>(xr, yr));   // x y + (x + y)(x - y)
                else if (op == 13) rr_mul(rr, rr_norm(rr_sub_twice<2>(rr_sub<2>(xr, yr), yr)), yr);  // (x - 3y) y
                else if (op == 15) {   // x - 3y + 12m brought below 2m by the quotient-digit reduction; then x (x - 3y)
                    const auto t = rr_reduce2m(rr_sub_twice<2>(rr_sub<2>(xr, yr), yr));
                    rr_mul(rr, t, xr);
                }
                else {  // 1 if x == y (exact test behind the cheap filter), else 0; Montgomery one / zero
                    const auto d = rr_sub<2>(xr, yr);
                    const bool eq = rr_maybe_equal(xr, yr) && rr_is_zero(d);
                    if (rr_is_zero(d) != eq) { rr_zero(rr); rr.v[0] = 7; }   // the filter must never hide an equality
                    else if (eq) rr_one(rr);
                    else rr_zero(rr);
                }
                rr_to_mont32_words<Q>(r.v, rr);
            } else {
                fp_zero(r);
            }
            break;
        default: fp_sqr(r, x); break;
    }
    fp_from_mont(r, r);
    fp_store(out + (size_t)i * P::N, r);
}

template <class F>
__global__ __launch_bounds__(64) void k_test_ec(int op, const uint32_t* p, const uint32_t* q, const uint8_t* inf_flags,
                                                uint32_t* out, uint8_t* out_inf, uint32_t n) {
    uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    Affine<F> P, Q;
    fp_load(P.x, p + (size_t)i * 2 * F::N);
    fp_load(P.y, p + (size_t)i * 2 * F::N + F::N);
    fp_load(Q.x, q + (size_t)i * 2 * F::N);
    fp_load(Q.y, q + (size_t)i * 2 * F::N + F::N);
    fp_to_mont(P.x, P.x); fp_to_mont(P.y, P.y); fp_to_mont(Q.x, Q.x); fp_to_mont(Q.y, Q.y);
    uint8_t fl = inf_flags[i];
    XYZZ<F> acc, qq;
    if (fl & 1) pt_set_inf(acc); else pt_from_affine(acc, P);
    if (fl & 2) pt_set_inf(qq); else pt_from_affine(qq, Q);
    // de-normalise the accumulator (scale by a non-trivial z) so the projective paths are exercised
    if (!(fl & 1)) {
        Fp<F> z, z2, z3;
        z = P.x; fp_add(z, z, Q.y);
        if (fp_is_zero(z)) fp_one(z);
        fp_sqr(z2, z); fp_mul(z3, z2, z);
        fp_mul(acc.x, acc.x, z2); fp_mul(acc.y, acc.y, z3); acc.zz = z2; acc.zzz = z3;
    }
    switch (op) {
        case 0: if (!(fl & 2)) pt_madd(acc, Q); break;
        case 1: { XYZZ<F> d; pt_dbl(d, acc); acc = d; } break;
        case 2: {
            if (!(fl & 2)) {  // de-normalise q too
                Fp<F> z, z2, z3;
                z = Q.x; fp_add(z, z, P.y);
                if (fp_is_zero(z)) fp_one(z);
                fp_sqr(z2, z); fp_mul(z3, z2, z);
                fp_mul(qq.x, qq.x, z2); fp_mul(qq.y, qq.y, z3); qq.zz = z2; qq.zzz = z3;
            }
            pt_add(acc, qq);
        } break;
        case 4: case 5:   // the reduced-radix mixed add of the bucket accumulation (ec_rr.hip.hpp); 5 subtracts
            if constexpr (USE_RR<F>) {
                if (!(fl & 2)) {
                    using QQ = typename F::RR;
                    XYZZRR<QQ> a;
                    ptrr_from_xyzz32<F>(a, acc);
                    AffineRR<QQ> q;
                    rr_from_mont32_words<QQ>(q.x, Q.x.v);
                    rr_from_mont32_words<QQ>(q.y, Q.y.v);
                    ptrr_madd<QQ, 2>(a, q, op == 5);
                    // a second, cancelling pair keeps the lazy ranges honest: (acc + Q) + Q - Q
                    ptrr_madd<QQ, 2>(a, q, op == 5);
                    ptrr_madd<QQ, 2>(a, q, op != 5);
                    ptrr_to_xyzz32<F>(acc, a);
                }
            } else {
                pt_set_inf(acc);
            }
            break;
        case 8: case 9:   // the full XYZZ add / doubling on the reduced radix (first bucket-reduce level): 8 = P + Q, 9 = 2 P
            if constexpr (USE_RR<F>) {
                using QQ = typename F::RR;
                if (op == 8 && !(fl & 2)) {  // de-normalise q too
                    Fp<F> z, z2, z3;
                    z = Q.x; fp_add(z, z, P.y);
                    if (fp_is_zero(z)) fp_one(z);
                    fp_sqr(z2, z); fp_mul(z3, z2, z);
                    fp_mul(qq.x, qq.x, z2); fp_mul(qq.y, qq.y, z3); qq.zz = z2; qq.zzz = z3;
                }
                XYZZRR<QQ> a, b;
                ptrr_from_xyzz32<F>(a, acc);
                ptrr_from_xyzz32<F>(b, qq);
                if (op == 8) ptrr_add<QQ, 2>(a, b);
                else a = ptrr_dbl_val<QQ, 2>(a);
                ptrr_to_xyzz32<F>(acc, a);
            } else {
                pt_set_inf(acc);
            }
            break;
        case 6: case 7:   // both operands affine (the first addition of a run, ptrr_aadd): 6 = P + Q, 7 = Q - P
            if constexpr (USE_RR<F>) {
                using QQ = typename F::RR;
                AffineRR<QQ> p, q;
                rr_from_mont32_words<QQ>(p.x, P.x.v);
                rr_from_mont32_words<QQ>(p.y, P.y.v);
                rr_from_mont32_words<QQ>(q.x, Q.x.v);
                rr_from_mont32_words<QQ>(q.y, Q.y.v);
                XYZZRR<QQ> a;
                ptrr_set_inf(a);
                if (fl == 0) ptrr_aadd<QQ, 2>(a, p, op == 7, q, false);
                else if (fl == 1) ptrr_madd<QQ, 2>(a, q, false);
                else if (fl == 2) ptrr_madd<QQ, 2>(a, p, op == 7);
                ptrr_to_xyzz32<F>(acc, a);
            } else {
                pt_set_inf(acc);
            }
            break;
        default: if (!(fl & 2)) { fp_neg(Q.y, Q.y); pt_madd(acc, Q); } break;
    }
    Affine<F> r;
    bool fin = pt_to_affine(r, acc);
    Fp<F> x, y;
    if (fin) { fp_from_mont(x, r.x); fp_from_mont(y, r.y); } else { fp_zero(x); fp_zero(y); }
    fp_store(out + (size_t)i * 2 * F::N, x);
    fp_store(out + (size_t)i * 2 * F::N + F::N, y);
    out_inf[i] = fin ? 0 : 1;
}

struct Tmp {
    std::vector<void*> ptrs;
    ~Tmp() { for (void* p : ptrs) (void)hipFree(p); }
    int alloc(void** p, size_t n) {
        BLZ_HIP(hipMalloc(p, n ? n : 16), BLZ_ERR_UNKNOWN);
        ptrs.push_back(*p);
        return BLZ_OK;
    }
};

// ops 20..24: the row-cooperative field arithmetic of ec_row.hip.hpp (a limb per lane, 16 lanes per element, four elements per
// wave).  Operands go in through the lane form's conversions and the results come back through them, so what is compared with
// the oracle is the value.  24 is a self-check of the carry machinery on SYNTHETIC limb patterns (random Montgomery forms
// practically never hold a run of 0xfffffff limbs): limbs drawn from {0, 1, MASK, MASK - 1, 2^28, 2^28 + 3, lazy ...} by the
// input bits, weak normalisation + exact resolve against a serial carry loop; the result is 1 (agree) or 0.
template <class P>
__global__ __launch_bounds__(64) void k_test_row(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, uint32_t n) {
    if constexpr (USE_RR<P>) {
        using Q = typename P::RR;
        if constexpr (!RR_TIGHT<Q> && Q::B == 28) {
            __shared__ uint32_t sh[2][4][16];
            const RowCtx<Q> c = row_ctx<Q>();
            // this row's element; the zero test (23) is wave-wide - the tail's rows are replicas - so it takes one element per wave
            const uint32_t e = op == 23 ? blockIdx.x : blockIdx.x * 4u + c.row;
            const uint32_t ee = e < n ? e : n - 1;
            Fp<P> x, y;
            fp_load(x, a + (size_t)ee * P::N);
            fp_load(y, b + (size_t)ee * P::N);
            uint32_t xl = 0, yl = 0;
            if (op == 24) {
                // limb li from 3 bits of the inputs
                const uint32_t bits = (x.v[c.li % P::N] >> (3u * (c.li / P::N))) ^ (y.v[(c.li * 5u) % P::N] >> 7);
                const uint32_t pat[8] = {0u, 1u, Q::MASK, Q::MASK - 1u, 1u << 28, (1u << 28) + 3u, 15u * (1u << 28) + Q::MASK, (7u << 28) + 1u};
                xl = c.li < (uint32_t)Q::NL ? pat[bits & 7u] : 0u;
                if (c.li == (uint32_t)Q::NL - 1u) xl &= 0xffffu;   // the top limb stays small (value bound)
            } else {
                Frr<Q, 1, 2> xr, yr;
                rr_to_mont_from_words<Q>(xr, x.v);
                rr_to_mont_from_words<Q>(yr, y.v);
                if ((threadIdx.x & 15u) == 0u) {
#pragma unroll
                    for (int i = 0; i < Q::NL; ++i) { sh[0][c.row][i] = xr.v[i]; sh[1][c.row][i] = yr.v[i]; }
                    sh[0][c.row][14] = sh[0][c.row][15] = sh[1][c.row][14] = sh[1][c.row][15] = 0u;
                }
                __syncthreads();
                xl = sh[0][c.row][c.li];
                yl = sh[1][c.row][c.li];
                __syncthreads();
            }
            uint32_t r = 0;
            if (op == 20) r = row_mul<Q>(c, xl, yl);
            else if (op == 21) r = row_mul<Q>(c, row_norm<Q>(c, xl + (c.km2 - yl) + 2u * (c.km2 - yl)), yl);   // (x - 3y + 12m) y
            else if (op == 22) {   // (x - y + 32m)(x + y): the 32m constant, a lazy operand on each side (F = 5 and 2: 11 <= 18)
                r = row_mul<Q>(c, xl + (c.km5 - yl), xl + yl);
            } else if (op == 23) {   // 1 if x == y else 0 (Montgomery one / zero), through the filter and the exact test
                const uint32_t d = row_norm<Q>(c, xl + (c.km2 - yl));
                const bool z = row_is_zero<Q>(c, d);
                const bool eq = row_maybe_equal<Q>(c, d) && z;
                r = z != eq ? (c.li == 0u ? 7u : 0u) : eq ? c.one : 0u;
            } else if (op == 24) {
                r = row_resolve<Q>(c, row_norm<Q>(c, xl));
            }
            if (op != 24) r = row_resolve<Q>(c, r);
            sh[0][c.row][c.li] = r;
            sh[1][c.row][c.li] = xl;
            __syncthreads();
            if (c.li == 0u && e < n && (op != 23 || c.row == 0u)) {
                Fp<P> o;
                if (op == 24) {
                    uint64_t carry = 0;
                    bool same = true;
                    for (int i = 0; i < Q::NL; ++i) {
                        const uint64_t t = (uint64_t)sh[1][c.row][i] + carry;
                        const uint32_t want = i + 1 < Q::NL ? (uint32_t)t & Q::MASK : (uint32_t)t;
                        carry = i + 1 < Q::NL ? t >> Q::B : 0;
                        same = same && want == sh[0][c.row][i];
                    }
                    fp_zero(o);
                    o.v[0] = same ? 1u : 0u;
                    fp_store(out + (size_t)e * P::N, o);
                } else {
                    Frr<Q, 1, 2> rr;
#pragma unroll
                    for (int i = 0; i < Q::NL; ++i) rr.v[i] = sh[0][c.row][i];
                    rr_to_mont32_words<Q>(o.v, rr);
                    fp_from_mont(o, o);
                    fp_store(out + (size_t)e * P::N, o);
                }
            }
        }
    }
}

template <class P>
int test_field_t(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
    Tmp tmp;
    size_t bytes = n * P::N * 4;
    void *da, *db, *dout;
    BLZ_TRY(tmp.alloc(&da, bytes)); BLZ_TRY(tmp.alloc(&db, bytes)); BLZ_TRY(tmp.alloc(&dout, bytes));
    BLZ_HIP(hipMemcpy(da, a, bytes, hipMemcpyHostToDevice), BLZ_ERR_WRITE);
    BLZ_HIP(hipMemcpy(db, b, bytes, hipMemcpyHostToDevice), BLZ_ERR_WRITE);
    if (op >= 20 && op <= 24) {
        BLZ_HIP(hipMemset(dout, 0, bytes), BLZ_ERR_UNKNOWN);   // (fields without the row arithmetic answer zero)
        hipLaunchKernelGGL(k_test_row<P>, dim3((unsigned)(op == 23 ? n : (n + 3) / 4)), dim3(64), 0, 0, op, (const uint32_t*)da, (const uint32_t*)db,
                           (uint32_t*)dout, (uint32_t)n);
    } else
        hipLaunchKernelGGL(k_test_field<P>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, op, (const uint32_t*)da,
                           (const uint32_t*)db, (uint32_t*)dout, (uint32_t)n);
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    BLZ_HIP(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost), BLZ_ERR_READ);
    return BLZ_OK;
}

template <class F>
int test_ec_t(int op, const uint8_t* p, const uint8_t* q, const uint8_t* fl, uint8_t* out, uint8_t* out_inf, size_t n) {
    Tmp tmp;
    size_t bytes = n * 2 * F::N * 4;
    void *dp, *dq, *dfl, *dout, *dinf;
    BLZ_TRY(tmp.alloc(&dp, bytes)); BLZ_TRY(tmp.alloc(&dq, bytes)); BLZ_TRY(tmp.alloc(&dfl, n));
    BLZ_TRY(tmp.alloc(&dout, bytes)); BLZ_TRY(tmp.alloc(&dinf, n));
    BLZ_HIP(hipMemcpy(dp, p, bytes, hipMemcpyHostToDevice), BLZ_ERR_WRITE);
    BLZ_HIP(hipMemcpy(dq, q, bytes, hipMemcpyHostToDevice), BLZ_ERR_WRITE);
    BLZ_HIP(hipMemcpy(dfl, fl, n, hipMemcpyHostToDevice), BLZ_ERR_WRITE);
    hipLaunchKernelGGL(k_test_ec<F>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, op, (const uint32_t*)dp,
                       (const uint32_t*)dq, (const uint8_t*)dfl, (uint32_t*)dout, (uint8_t*)dinf, (uint32_t)n);
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    BLZ_HIP(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost), BLZ_ERR_READ);
    BLZ_HIP(hipMemcpy(out_inf, dinf, n, hipMemcpyDeviceToHost), BLZ_ERR_READ);
    return BLZ_OK;
}

}  // namespace blz

using namespace blz;

extern "C" {

int blz_test_field_op(int device_id, int curve, int field, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
    BLZ_TRY(use_device(device_id));
    if (n == 0) return BLZ_OK;
    if (field == 0) {
        switch (curve) {
            case BLZ_BLS377: return test_field_t<Fq_BLS377>(op, a, b, out, n);
            case BLZ_BLS381: return test_field_t<Fq_BLS381>(op, a, b, out, n);
            case BLZ_BN254: return test_field_t<Fq_BN254>(op, a, b, out, n);
        }
    } else {
        switch (curve) {
            case BLZ_BLS377: return test_field_t<Fr_BLS377>(op, a, b, out, n);
            case BLZ_BLS381: return test_field_t<Fr_BLS381>(op, a, b, out, n);
            case BLZ_BN254: return test_field_t<Fr_BN254>(op, a, b, out, n);
        }
    }
    return fail(BLZ_ERR_INVALID_PARAM, "unknown curve %d", curve);
}

int blz_test_ec_op(int device_id, int curve, int op, const uint8_t* p, const uint8_t* q, const uint8_t* inf_flags,
                   uint8_t* out, uint8_t* out_inf, size_t n) {
    BLZ_TRY(use_device(device_id));
    if (n == 0) return BLZ_OK;
    switch (curve) {
        case BLZ_BLS377: return test_ec_t<Fq_BLS377>(op, p, q, inf_flags, out, out_inf, n);
        case BLZ_BLS381: return test_ec_t<Fq_BLS381>(op, p, q, inf_flags, out, out_inf, n);
        case BLZ_BN254: return test_ec_t<Fq_BN254>(op, p, q, inf_flags, out, out_inf, n);
    }
    return fail(BLZ_ERR_INVALID_PARAM, "unknown curve %d", curve);
}

}  // extern "C"
