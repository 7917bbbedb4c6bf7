// NTT kernels, templated on the scalar field (included by ntt_<field>.hip; one translation unit per
// field keeps the build parallel).  Design notes: ntt.hip.
#pragma once
#include "ntt_engine.hpp"
#include "field.hip.hpp"
#include "ntt_rr.hip.hpp"

namespace blz {

// The transform's root of unity w, Montgomery form, into wbase: the field generator's ROOT^(2^(TWO_ADICITY - logn)) - or the
// CALLER's (blz_ntt_new_ex3: `user` = 8 canonical words): any primitive 2^logn-th root, checked here (w^(n/2) == -1 is exactly
// "order 2^logn"); the inverse transform runs on w^-1.  *flag: 0 fine, 1 not a canonical field element, 2 not primitive.
template <class Fr>
__global__ void k_ntt_root(uint32_t* wbase, uint32_t* flag, const uint32_t* user, int logn, int inverse) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    using E = Fp<Fr>;
    E w;
    if (user) {
        E x;
        fp_load(x, user);
        bool lt = false;   // x < m ?
        for (int i = Fr::N - 1; i >= 0; --i) {
            if (x.v[i] != Fr::MOD[i]) { lt = x.v[i] < Fr::MOD[i]; break; }
        }
        if (!lt) { *flag = 1u; return; }
        fp_to_mont(w, x);
        E t = w, m1, one;
        for (int i = 0; i + 1 < logn; ++i) fp_sqr(t, t);
        fp_one(one);
        fp_neg(m1, one);
        if (!fp_eq(t, m1)) { *flag = 2u; return; }
        if (inverse) { E wi; fp_inv(wi, w); w = wi; }
        fp_reduce(w);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) w.v[i] = inverse ? Fr::ROOT_INV[i] : Fr::ROOT[i];
        for (int i = 0; i < Fr::TWO_ADICITY - logn; ++i) fp_sqr(w, w);
    }
    fp_store(wbase, w);
}

// out[j] = w^(j * mult), w = *wbase
template <class Fr>
__global__ void k_ntt_table(uint32_t* out, int count, const uint32_t* __restrict__ wbase, uint64_t mult) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    using E = Fp<Fr>;
    E w;
    fp_load(w, wbase);
    uint64_t e = (uint64_t)j * mult;
    E acc;
    fp_one(acc);
    for (int b = 63; b >= 0; --b) {
        fp_sqr(acc, acc);
        if ((e >> b) & 1) fp_mul(acc, acc, w);
    }
    fp_store(out + (size_t)j * 8, acc);
}

// out = (2^logn)^-1 in Montgomery form
template <class Fr>
__global__ void k_ntt_ninv(uint32_t* out, int logn) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    using E = Fp<Fr>;
    E two, acc;
    fp_one(two);
    fp_add(two, two, two);
    fp_one(acc);
    for (int i = 0; i < logn; ++i) fp_mul(acc, acc, two);
    E inv;
    fp_inv(inv, acc);
    fp_store(out, inv);
}

template <class E>
BLZ_DEV void lds_load(E& r, const uint32_t* lds, uint32_t dw) {  // dw: dword offset, multiple of 8
    const uint4* q = reinterpret_cast<const uint4*>(lds + dw);
    uint4 a = q[0], b = q[1];
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
}
template <class E>
BLZ_DEV void lds_store(uint32_t* lds, uint32_t dw, const E& r) {
    uint4* q = reinterpret_cast<uint4*>(lds + dw);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

// w^e for e < 2^27 from the three 512-entry tables
template <class E>
BLZ_DEV void tw_pow(E& r, const NttTables& T, uint32_t e) {
    E a, b;
    fp_load(r, T.t0 + (size_t)(e & 511u) * 8);
    uint32_t e1 = (e >> 9) & 511u, e2 = e >> 18;
    if (e1) { fp_load(a, T.t1 + (size_t)e1 * 8); fp_mul(r, r, a); }
    if (e2) { fp_load(b, T.t2 + (size_t)e2 * 8); fp_mul(r, r, b); }
}

// Data arithmetic of the transform.  Fields with head-room (P::LAZY: BLS12-377 / BN254 Fr) already
// multiply without a final subtraction; BLS12-381 Fr gets the wide-lazy forms of field.hip.hpp: data live
// in [0, 2m), twiddles stay canonical, only the last pass reduces to the wire format.
template <class E> struct NttOps;
template <class Fr>
struct NttOps<Fp<Fr>> {
    using E = Fp<Fr>;
    static BLZ_DEV void mul(E& r, const E& x, const E& w) {
        if constexpr (Fr::LAZY) fp_mul(r, x, w); else fp_mul_nr(r, x, w);
    }
    static BLZ_DEV void add(E& r, const E& a, const E& b) {
        if constexpr (Fr::LAZY) fp_add(r, a, b); else fp_add_wide(r, a, b);
    }
    static BLZ_DEV void sub(E& r, const E& a, const E& b) {
        if constexpr (Fr::LAZY) fp_sub(r, a, b); else fp_sub_wide(r, a, b);
    }
    static BLZ_DEV void canon(E& a) {
        if constexpr (Fr::LAZY) fp_reduce(a); else fp_canon_wide(a);
    }
    // A word off the wire is canonical by contract; any other 32-byte value is still a residue and is brought into
    // [0, 2m) by a product with one (x < 2^256 = R and one < m: the result is < 2m).  The top-limb test lets every
    // word >= m through to the product and almost no canonical one.
    static BLZ_DEV void wire_in(E& x) {
        if (__builtin_expect(x.v[Fr::N - 1] >= Fr::MOD[Fr::N - 1], 0)) {
            E one;
            fp_one(one);
            mul(x, x, one);
        }
    }
};

constexpr int NTT_THREADS = 1024;  // 4 waves per SIMD: the 128-147 KiB tile allows one block per CU

// LDS tile: element (row, col) at dword offset row * RS + col * 8 with RS = COLS * 8 + 8: the one
// element of padding per row spreads rows over the banks (a power-of-two row stride puts every row
// of a column on the same banks, and the bit-reversed row order of the load makes that the norm).
BLZ_DEV uint32_t ntt_rs(uint32_t cols) { return cols * 8u + 8u; }

// PASS 1..3 as in the header comment.  Tile = COLS columns x (1 << lr) rows.
// Inter-pass twiddles (the w^(i0 k2) factor of pass 2 is applied in pass 1: it does not depend on
// the index pass 2 transforms over):
//   after pass 1: x(i0, i1, k2) *= w^(k2 (i0 + A i1))      step along i0: w^k2
//   after pass 2: x(i0, k1, k2) *= w^(C i0 k1)             step along i0: w^(C k1)
// so a lane that owns a few adjacent columns of one row derives its twiddles by repeated
// multiplication from one table look-up.
template <class Fr, int PASS>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                          NttGeom g, NttTables T, int cols_log) {
    using E = Fp<Fr>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lr = PASS == 1 ? g.logC : PASS == 2 ? g.logB : g.logA;  // log radix of this pass
    const uint32_t radix = 1u << lr;
    const uint32_t COLS = 1u << cols_log;
    const uint32_t RS = ntt_rs(COLS);
    const uint32_t A = 1u << g.logA, B = 1u << g.logB, C = 1u << g.logC;
    uint64_t col_base, fixed, in_base, in_rstride, in_cstride;
    const uint64_t tile = blockIdx.x;
    if (PASS == 1) {  // rows i2 (stride AB), cols i0 (stride 1), fixed i1
        uint64_t tiles_per = A >> cols_log;
        fixed = tile / tiles_per;
        col_base = (tile % tiles_per) << cols_log;
        in_base = col_base + (uint64_t)A * fixed;
        in_rstride = (uint64_t)A * B;
        in_cstride = 1;
    } else if (PASS == 2) {  // rows i1 (stride A), cols i0, fixed k2
        uint64_t tiles_per = A >> cols_log;
        fixed = tile / tiles_per;
        col_base = (tile % tiles_per) << cols_log;
        in_base = col_base + (uint64_t)A * B * fixed;
        in_rstride = A;
        in_cstride = 1;
    } else {  // rows i0 (stride 1), cols k2 (stride AB), fixed k1
        uint64_t tiles_per = C >> cols_log;
        fixed = tile / tiles_per;
        col_base = (tile % tiles_per) << cols_log;
        in_base = (uint64_t)A * fixed + (uint64_t)A * B * col_base;
        in_rstride = 1;
        in_cstride = (uint64_t)A * B;
    }
    const uint32_t total = radix << cols_log;
    // ---- load, rows bit-reversed (decimation in time)
    for (uint32_t e = threadIdx.x; e < total; e += NTT_THREADS) {
        uint32_t row, col;
        if (PASS == 3) { row = e & (radix - 1); col = e >> lr; }   // contiguous along rows
        else { col = e & (COLS - 1); row = e >> cols_log; }         // contiguous along cols
        E x;
        uint64_t iaddr = in_base + row * in_rstride + col * in_cstride;
        if (PASS == g.wire_pass && g.brin) iaddr = __brevll(iaddr) >> (64 - g.logn);   // the caller's buffer is in bit-reversed order
        fp_load(x, in + iaddr * 8);
        if (PASS == g.wire_pass) NttOps<E>::wire_in(x);
        uint32_t rrow = lr ? (__brev(row) >> (32 - lr)) : 0;
        lds_store(lds, rrow * RS + col * 8, x);
    }
    __syncthreads();
    // ---- radix-2 stages
    const uint32_t* wp = T.wpass[PASS - 1];
    const uint32_t nbf = (radix >> 1) << cols_log;
    for (int s = 1; s <= lr; ++s) {
        const uint32_t half = 1u << (s - 1);
        for (uint32_t e = threadIdx.x; e < nbf; e += NTT_THREADS) {
            uint32_t col = e & (COLS - 1), b = e >> cols_log;
            uint32_t blk = b >> (s - 1), k = b & (half - 1);
            uint32_t u = (blk << s) + k, v = u + half;
            E xu, xv;
            lds_load(xu, lds, u * RS + col * 8);
            lds_load(xv, lds, v * RS + col * 8);
            uint32_t tw = k << (lr - s);
            if (tw) {
                E w;
                fp_load(w, wp + (size_t)tw * 8);
                NttOps<E>::mul(xv, xv, w);
            }
            E sum, dif;
            NttOps<E>::add(sum, xu, xv);
            NttOps<E>::sub(dif, xu, xv);
            lds_store(lds, u * RS + col * 8, sum);
            lds_store(lds, v * RS + col * 8, dif);
        }
        __syncthreads();
    }
    // ---- inter-pass twiddle + store: one lane owns CG adjacent columns of one row
    const uint32_t cg_log = cols_log < 2 ? cols_log : 2;
    const uint32_t CG = 1u << cg_log;
    const uint32_t ntask = total >> cg_log;
    for (uint32_t t = threadIdx.x; t < ntask; t += NTT_THREADS) {
        const uint32_t grp = t & ((COLS >> cg_log) - 1), row = t >> (cols_log - cg_log);
        const uint32_t col0 = grp << cg_log;
        E w, step;
        bool tw = false;
        if (PASS == 1) {
            // k2 = row, i1 = fixed, i0 = col_base + col0 + j
            uint32_t ex = (uint32_t)((uint64_t)row * (col_base + col0 + ((uint64_t)fixed << g.logA)));
            tw = row != 0;
            if (tw) { tw_pow(w, T, ex); fp_load(step, T.t0 + (size_t)row * 8); }
        } else if (PASS == 2) {
            // k1 = row, i0 = col_base + col0 + j
            uint32_t ex = (uint32_t)(((uint64_t)row * (col_base + col0)) << g.logC);
            tw = row != 0;
            if (tw) { tw_pow(w, T, ex); tw_pow(step, T, row << g.logC); }
        }
#pragma unroll 4
        for (uint32_t j = 0; j < CG; ++j) {
            E x;
            lds_load(x, lds, row * RS + (col0 + j) * 8);
            uint64_t oaddr;
            if (PASS == 3) {
                // element (k0 = row, k1 = fixed, k2 = col_base + col) -> natural address k2 + C k1 + CB k0
                if (T.ninv) { E s; fp_load(s, T.ninv); NttOps<E>::mul(x, x, s); }  // inverse transform: * n^-1
                NttOps<E>::canon(x);                                               // the wire format is canonical
                oaddr = (col_base + col0 + j) + (uint64_t)C * fixed + (uint64_t)C * B * row;
                if (g.brout) oaddr = __brevll(oaddr) >> (64 - g.logn);
            } else {
                if (tw) {
                    NttOps<E>::mul(x, x, w);
                    if (j + 1 < CG) fp_mul(w, w, step);   // twiddle x twiddle: stays canonical
                }
                oaddr = in_base + row * in_rstride + col0 + j;
            }
            fp_store(out + oaddr * 8, x);
        }
    }
}


// (The 512-point passes - 512 = 8 * 8 * 8 in registers, two LDS exchanges - live in ntt_rr.hip.hpp on the reduced-radix field;
// their 32-bit-limb predecessor k_ntt512 lost every A/B run since round 2 and was removed in round 4.)

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
template <class Fr>
int ntt_setup_t(hipStream_t st, NttTables& T, NttTablesRR& TR, const NttGeom& g, int inverse, const uint32_t* user_root, uint32_t* flag) {
    const int l = g.logn;
    if (l > Fr::TWO_ADICITY) return fail(BLZ_ERR_INVALID_PARAM, "log_size %d exceeds the field's two-adicity %d", l, Fr::TWO_ADICITY);
    hipLaunchKernelGGL(k_ntt_root<Fr>, dim3(1), dim3(64), 0, st, T.wbase, flag, user_root, l, inverse);
    if (inverse) hipLaunchKernelGGL(k_ntt_ninv<Fr>, dim3(1), dim3(64), 0, st, T.ninv, l);
    const uint64_t n = 1ull << l;
    int lrs[3] = {g.logC, g.logB, g.logA};
    for (int i = 0; i < 3; ++i) {
        int cnt = lrs[i] ? (1 << lrs[i]) : 1;  // the whole circle: the radix-8 kernel indexes exponents up to radix-1
        hipLaunchKernelGGL(k_ntt_table<Fr>, dim3(2), dim3(256), 0, st, T.wpass[i], cnt, (const uint32_t*)T.wbase, n >> lrs[i]);
    }
    hipLaunchKernelGGL(k_ntt_table<Fr>, dim3(2), dim3(256), 0, st, T.t0, 512, (const uint32_t*)T.wbase, (uint64_t)1);
    hipLaunchKernelGGL(k_ntt_table<Fr>, dim3(2), dim3(256), 0, st, T.t1, 512, (const uint32_t*)T.wbase, (uint64_t)512);
    hipLaunchKernelGGL(k_ntt_table<Fr>, dim3(2), dim3(256), 0, st, T.t2, 512, (const uint32_t*)T.wbase, (uint64_t)1 << 18);
    // the same tables in the reduced radix for the 512-point kernel (ntt_rr.hip.hpp)
    static_assert(rr_stride<typename Fr::RR>() == NTT_RR_ENTRY_DWORDS, "table entry size");
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(k_ntt_table_to_shoup<Fr>, dim3(2), dim3(256), 0, st, T.wpass[i], TR.wpass[i], lrs[i] ? (1 << lrs[i]) : 1);
    hipLaunchKernelGGL(k_ntt_table_to_rr<Fr>, dim3(2), dim3(256), 0, st, T.t0, TR.t0, 512);
    hipLaunchKernelGGL(k_ntt_table_to_rr<Fr>, dim3(2), dim3(256), 0, st, T.t1, TR.t1, 512);
    hipLaunchKernelGGL(k_ntt_table_to_rr<Fr>, dim3(2), dim3(256), 0, st, T.t2, TR.t2, 512);
    if (TR.fin) hipLaunchKernelGGL(k_ntt_fin_rr<Fr>, dim3(1), dim3(64), 0, st, (const uint32_t*)T.ninv, TR.fin);
    hipLaunchKernelGGL((k_ntt_table_rr_pow<Fr, true>), dim3(2), dim3(256), 0, st, TR.ts2, 512u, (const uint32_t*)T.wbase, (uint64_t)64 << g.logC);   // Shoup entries
    if (TR.tA)
        hipLaunchKernelGGL((k_ntt_table_rr_pow<Fr, NTT_TA_SHOUP>), dim3((unsigned)(NTT_RR_BOUNDARY_ENTRIES / 256)), dim3(256), 0, st, TR.tA,
                           (uint32_t)NTT_RR_BOUNDARY_ENTRIES, (const uint32_t*)T.wbase, (uint64_t)1 << g.logA);
    if (TR.tB)
        hipLaunchKernelGGL(k_ntt_table_b<Fr>, dim3((unsigned)(n / 256)), dim3(256), 0, st, TR.tB, g, TR);   // (after t0 / t1 / t2: same stream)
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

template <class Fr, int PASS>
int ntt_pass_t(hipStream_t st, const void* in, void* out, const NttGeom& g, const NttTables& T, const NttTablesRR& TR, int cl,
               bool force_generic) {
    int lr = PASS == 1 ? g.logC : PASS == 2 ? g.logB : g.logA;
    const int cols_avail = PASS == 3 ? g.logC : g.logA;  // extent of the tile's column index
    if (lr == 9 && cols_avail >= NR_COLS_LOG && !force_generic) {
        // the tile goes through the LDS in two halves: 256 rows x 4 columns x 40 bytes
        const size_t ldsr = (size_t)256 * NR_COLS * rr_stride<typename Fr::RR>() * 4;
        const uint64_t tilesr = (1ull << g.logn) >> (9 + NR_COLS_LOG);
        if (PASS == 2 && TR.tB) {
            BLZ_TRY(ensure_dynamic_lds((const void*)k_ntt512_rr<Fr, 2, true>, 160 * 1024));
            hipLaunchKernelGGL((k_ntt512_rr<Fr, 2, true>), dim3((unsigned)tilesr), dim3(NR_THREADS), ldsr, st, (const uint32_t*)in,
                               (uint32_t*)out, g, TR);
        } else {
            BLZ_TRY(ensure_dynamic_lds((const void*)k_ntt512_rr<Fr, PASS>, 160 * 1024));
            hipLaunchKernelGGL((k_ntt512_rr<Fr, PASS>), dim3((unsigned)tilesr), dim3(NR_THREADS), ldsr, st, (const uint32_t*)in,
                               (uint32_t*)out, g, TR);
        }
        BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
        return BLZ_OK;
    }
    size_t lds = ((size_t)4 << lr) * (((size_t)8 << cl) + 8);  // rows x (COLS*8 + 8) dwords
    uint64_t tiles = (1ull << g.logn) >> (lr + cl);
    BLZ_TRY(ensure_dynamic_lds((const void*)k_ntt_pass<Fr, PASS>, 160 * 1024));
    hipLaunchKernelGGL((k_ntt_pass<Fr, PASS>), dim3((unsigned)tiles), dim3(NTT_THREADS), lds, st, (const uint32_t*)in,
                       (uint32_t*)out, g, T, cl);
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

template <class Fr>
int ntt_pass_dispatch(int pass, hipStream_t st, const void* in, void* out, const NttGeom& g, const NttTables& T,
                      const NttTablesRR& TR, int cl, bool force_generic) {
    switch (pass) {
        case 1: return ntt_pass_t<Fr, 1>(st, in, out, g, T, TR, cl, force_generic);
        case 2: return ntt_pass_t<Fr, 2>(st, in, out, g, T, TR, cl, force_generic);
        default: return ntt_pass_t<Fr, 3>(st, in, out, g, T, TR, cl, force_generic);
    }
}

template <class Fr>
NttFieldOps make_ntt_ops() {
    NttFieldOps o;
    o.two_adicity = Fr::TWO_ADICITY;
    o.setup = &ntt_setup_t<Fr>;
    o.pass = &ntt_pass_dispatch<Fr>;
    return o;
}

}  // namespace blz
