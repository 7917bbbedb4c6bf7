// On-device expansion of the precompute table: for every base P the PRECOMPUTE_FACTOR = 8 points
// P, 2^32 P, ..., 2^224 P, contiguous, wire format x||y canonical LE - what precompute_base_*
// (tests/msm/mod.rs:115-135, :235-255, :360-380) builds on the host with 7 scalar multiplications per
// point.  At n = 2^26 the table is 48 GiB (BLS) / 32 GiB (BN254): it does not fit the pinned host
// memory of a typical box but fits the 288 GB of HBM, so it is built where it is used
// (SURVEY.md 8(f) rank 4).  One lane per base: 7 x (32 doublings + normalisation).
#include "common.hpp"
#include "ec.hip.hpp"

namespace blz {

template <class F>
__global__ __launch_bounds__(64) void k_precompute_bases(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const uint32_t* src = in + i * 2 * F::N;
    uint32_t* dst = out + i * 8 * 2 * F::N;
    Affine<F> a;
    fp_load(a.x, src);
    fp_load(a.y, src + F::N);
    fp_store(dst, a.x);            // j = 0: the base itself, bytes unchanged
    fp_store(dst + F::N, a.y);
    fp_to_mont(a.x, a.x);
    fp_to_mont(a.y, a.y);
    for (int j = 1; j < 8; ++j) {
        XYZZ<F> p;
        pt_mdbl(p, a);
        for (int d = 1; d < 32; ++d) {
            XYZZ<F> t;
            pt_dbl(t, p);
            p = t;
        }
        pt_to_affine(a, p);        // r-torsion points: never infinity
        Fp<F> x, y;
        fp_from_mont(x, a.x);
        fp_from_mont(y, a.y);
        fp_store(dst + (size_t)j * 2 * F::N, x);
        fp_store(dst + (size_t)j * 2 * F::N + F::N, y);
    }
}

template <class F>
int precompute_t(const void* d_in, void* d_out, uint64_t n) {
    if (n == 0) return BLZ_OK;
    hipLaunchKernelGGL(k_precompute_bases<F>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, (const uint32_t*)d_in,
                       (uint32_t*)d_out, n);
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    BLZ_TRY(sync_stream_bounded(0, "precompute expansion"));
    return BLZ_OK;
}

}  // namespace blz

using namespace blz;

extern "C" int blz_msm_precompute_bases_device(int device_id, int curve, const void* d_points, void* d_out, uint64_t n) {
    BLZ_TRY(use_device(device_id));
    if ((!d_points || !d_out) && n) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    switch (curve) {
        case BLZ_BLS377: return precompute_t<Fq_BLS377>(d_points, d_out, n);
        case BLZ_BLS381: return precompute_t<Fq_BLS381>(d_points, d_out, n);
        case BLZ_BN254: return precompute_t<Fq_BN254>(d_points, d_out, n);
    }
    return fail(BLZ_ERR_INVALID_PARAM, "unknown curve %d", curve);
}
