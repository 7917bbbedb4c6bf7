// Host-side interface of the NTT device passes (per-field translation units: ntt_<field>.hip).
#pragma once
#include "common.hpp"

namespace blz {

struct NttTables {  // all Montgomery form, device memory
    uint32_t* wpass[3];  // wpass[p][j] = root_p^j, j < radix_p       (root_p = primitive radix_p-th root)
    uint32_t* t0;        // w^j            j < 512
    uint32_t* t1;        // w^(512 j)      j < 512
    uint32_t* t2;        // w^(2^18 j)     j < 512
    uint32_t* ninv;      // n^-1 (Montgomery) for the inverse transform, else nullptr
    uint32_t* wbase;     // w (Montgomery): the transform's primitive 2^logn-th root - the field generator's (ROOT^(2^(s - logn)))
                         // or the caller's (blz_ntt_new_ex3), inverted for the inverse transform; every table is powers of it
};

// reduced-radix twiddle tables of the 512-point kernel (ntt_rr.hip.hpp): entries of 10 dwords (27-bit limbs), Montgomery
// R_rr, < 2m
struct NttTablesRR {
    // wpass and tA hold SHOUP entries (canonical twiddle | floor(twiddle R_rr / m): 2 x 10 dwords, field_rr.hip.hpp
    // rr_mul_shoup), ts2 too (the step of pass 2's boundary twiddle: a constant); t0 / t1 / t2 / fin stay Montgomery (R_rr)
    uint32_t* wpass[3];
    uint32_t* t0;
    uint32_t* t1;
    uint32_t* t2;
    uint32_t* fin;   // closing factor of the last pass: n^-1 R_rr (inverse) or nullptr (forward: no product)
    uint32_t* tA;    // w^(A e), e < 2^18 (n / 512 entries, 10 MiB), or nullptr: the boundary factor after pass 1 that
                     // does not depend on the column, read instead of stepped (2^27 transforms only; ntt_rr.hip.hpp)
    uint32_t* ts2;   // w^(64 C i0), i0 < 512: the step of pass 2's boundary factor along a lane's rows (one per column)
    uint32_t* tB;    // pass 2's boundary factor of EVERY element, w^((C k1 + k2) i0) R_rr mod m as 8 words at the element's own index
                     // (n x 32 bytes: 4 GiB at 2^27), or nullptr: stepped along the lane's rows (ntt_rr.hip.hpp)
    uint32_t swz;    // 0: plain tile order; 1 + s: pass 1 walks its tiles in the channel-spreading order with 2^s adjacent column
                     // groups back to back (2^27 transforms; ntt_rr.hip.hpp)
};
constexpr size_t NTT_RR_BOUNDARY_ENTRIES = (size_t)1 << 18;
constexpr size_t NTT_RR_ENTRY_DWORDS = 10;
constexpr size_t NTT_RR_TABLE_BYTES = (4 * 512 * 2 + 3 * 512 + 1) * NTT_RR_ENTRY_DWORDS * 4;   // 4 Shoup tables (wpass x 3, ts2: 2 entries' worth each), 3 Montgomery, fin
constexpr size_t NTT_RR_BOUNDARY_BYTES = NTT_RR_BOUNDARY_ENTRIES * 2 * NTT_RR_ENTRY_DWORDS * 4;   // tA, Shoup entries

struct NttGeom {
    int logA, logB, logC, logn;
    int wire_pass;   // the pass that reads the caller's words (1, 2 or 3): the first one that runs
    int brin, brout; // blz_ntt_new_ex3: the caller's input / output buffers are in bit-reversed order (position p holds the element of
                     // index bitrev(p)): folded into the wire pass's loads / the last pass's stores
};

// per-field entry points.  Field ids follow enum blz_curve: the scalar field Fr of that curve.
struct NttFieldOps {
    int two_adicity;
    // fill the twiddle tables (device memory already carved into T) for a 2^logn transform
    // user_root: device pointer to the caller's root (8 canonical words) or nullptr; *flag (device u32): 1 the root is not a
    // canonical field element, 2 it is not a primitive 2^logn-th root of unity
    int (*setup)(hipStream_t st, NttTables& T, NttTablesRR& TR, const NttGeom& g, int inverse, const uint32_t* user_root, uint32_t* flag);
    // one of the three passes; cols_log is the tile width of the radix-2-in-LDS kernel
    int (*pass)(int pass, hipStream_t st, const void* in, void* out, const NttGeom& g, const NttTables& T, const NttTablesRR& TR,
                int cols_log, bool force_generic);
};
const NttFieldOps& ntt_ops_bls377();
const NttFieldOps& ntt_ops_bls381();
const NttFieldOps& ntt_ops_bn254();

}  // namespace blz
