// Synthetic, device-resident inputs for bench.py and the full-size tests (include/blaze_hip.h
// "synthetic inputs").  Points have known discrete logs, P_i = (start+i+1) G, so the expected MSM
// result of any size is one scalar multiplication on the CPU (oracle: orc_index_weighted_sum);
// with pf = 8 element i carries B_{i,j} = 2^(32 j) P_i, the table tests/msm/mod.rs:360-380 builds.
#include "common.hpp"
#include "ec.hip.hpp"

namespace blz {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

// n x 32 B little-endian values in [0, m) of the 8-limb field P
template <class P>
__global__ __launch_bounds__(256) void k_synth_scalars(uint32_t* out, uint64_t n, uint64_t seed, uint64_t start) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Fp<P> v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint64_t r = splitmix64(seed * 0x2545f4914f6cdd1dull + 4 * (start + i) + k);
        v.v[2 * k] = (uint32_t)r;
        v.v[2 * k + 1] = (uint32_t)(r >> 32);
    }
    constexpr int topbits = P::BITS - 224;  // bits used in the top limb
    if constexpr (topbits < 32) v.v[7] &= (1u << topbits) - 1u;
    fp_csub_const<P, P::MOD>(v);  // value < 2^BITS < 2m
    fp_store(out + i * 8, v);
}

// table[j] = 2^(32 j) G, affine Montgomery
template <class F>
__global__ void k_synth_steps(uint32_t* table, int pf) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= pf) return;
    Affine<F> g;
#pragma unroll
    for (int i = 0; i < F::N; ++i) { g.x.v[i] = F::GX[i]; g.y.v[i] = F::GY[i]; }
    XYZZ<F> p;
    pt_from_affine(p, g);
    for (int d = 0; d < 32 * j; ++d) { XYZZ<F> t; pt_dbl(t, p); p = t; }
    Affine<F> a;
    pt_to_affine(a, p);
    fp_store(table + (size_t)j * 2 * F::N, a.x);
    fp_store(table + (size_t)j * 2 * F::N + F::N, a.y);
}

constexpr int SYNTH_K = 4;

// out[(i*pf + j)] = (start + i + 1) * table[j], wire format x||y canonical LE
template <class F>
__global__ __launch_bounds__(64) void k_synth_points(const uint32_t* __restrict__ table, uint32_t* __restrict__ out,
                                                     uint64_t n, int pf, uint64_t start) {
    uint64_t t = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    int j = blockIdx.y;
    uint64_t i0 = t * SYNTH_K;
    if (i0 >= n) return;
    Affine<F> A;
    fp_load(A.x, table + (size_t)j * 2 * F::N);
    fp_load(A.y, table + (size_t)j * 2 * F::N + F::N);
    uint64_t s = start + i0 + 1;
    XYZZ<F> pts[SYNTH_K];
    {
        XYZZ<F> acc;
        pt_set_inf(acc);
        int top = 63 - __builtin_clzll(s);
        for (int b = top; b >= 0; --b) {
            XYZZ<F> d;
            pt_dbl(d, acc);
            acc = d;
            if ((s >> b) & 1) pt_madd(acc, A);
        }
        pts[0] = acc;
    }
#pragma unroll
    for (int k = 1; k < SYNTH_K; ++k) {
        pts[k] = pts[k - 1];
        pt_madd(pts[k], A);
    }
    // batch inversion of the zzz coordinates (Montgomery's trick)
    Fp<F> c[SYNTH_K];
    c[0] = pts[0].zzz;
#pragma unroll
    for (int k = 1; k < SYNTH_K; ++k) fp_mul(c[k], c[k - 1], pts[k].zzz);
    Fp<F> inv;
    fp_inv(inv, c[SYNTH_K - 1]);
#pragma unroll
    for (int k = SYNTH_K - 1; k >= 0; --k) {
        Fp<F> iz3;
        if (k > 0) { fp_mul(iz3, inv, c[k - 1]); fp_mul(inv, inv, pts[k].zzz); }
        else iz3 = inv;
        if (i0 + k < n) {
            Fp<F> zi, x, y;
            fp_mul(zi, pts[k].zz, iz3);
            fp_mul(y, pts[k].y, iz3);
            fp_sqr(zi, zi);
            fp_mul(x, pts[k].x, zi);
            fp_from_mont(x, x);
            fp_from_mont(y, y);
            uint32_t* q = out + ((i0 + k) * pf + j) * 2 * F::N;
            fp_store(q, x);
            fp_store(q + F::N, y);
        }
    }
}

template <class F>
int synth_points_t(void* d_out, uint64_t n, int pf, uint64_t start) {
    void* table = nullptr;
    BLZ_HIP(hipMalloc(&table, 8 * 2 * F::N * 4), BLZ_ERR_UNKNOWN);
    hipLaunchKernelGGL(k_synth_steps<F>, dim3(1), dim3(64), 0, 0, (uint32_t*)table, pf);
    uint64_t threads = (n + SYNTH_K - 1) / SYNTH_K;
    if (threads) {
        hipLaunchKernelGGL(k_synth_points<F>, dim3((unsigned)((threads + 63) / 64), pf), dim3(64), 0, 0,
                           (const uint32_t*)table, (uint32_t*)d_out, n, pf, start);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(table);
    if (e != hipSuccess) return fail_hip(BLZ_ERR_UNKNOWN, "synth_points failed: %s", hipGetErrorString(e));
    return BLZ_OK;
}

template <class P>
int synth_scalars_t(void* d_out, uint64_t n, uint64_t seed, uint64_t start) {
    if (n) hipLaunchKernelGGL(k_synth_scalars<P>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (uint32_t*)d_out, n, seed, start);
    BLZ_HIP(hipGetLastError(), BLZ_ERR_UNKNOWN);
    BLZ_HIP(hipDeviceSynchronize(), BLZ_ERR_UNKNOWN);
    return BLZ_OK;
}

}  // namespace blz

using namespace blz;

extern "C" {

int blz_synth_scalars_at(int device_id, int curve, void* d_out, uint64_t n, uint64_t seed, uint64_t start) {
    BLZ_TRY(use_device(device_id));
    switch (curve) {
        case BLZ_BLS377: return synth_scalars_t<Fr_BLS377>(d_out, n, seed, start);
        case BLZ_BLS381: return synth_scalars_t<Fr_BLS381>(d_out, n, seed, start);
        case BLZ_BN254: return synth_scalars_t<Fr_BN254>(d_out, n, seed, start);
    }
    return fail(BLZ_ERR_INVALID_PARAM, "unknown curve %d", curve);
}

int blz_synth_scalars(int device_id, int curve, void* d_out, uint64_t n, uint64_t seed) {
    return blz_synth_scalars_at(device_id, curve, d_out, n, seed, 0);
}

int blz_synth_points(int device_id, int curve, void* d_out, uint64_t n, int pf, uint64_t start) {
    BLZ_TRY(use_device(device_id));
    if (pf != 1 && pf != 8) return fail(BLZ_ERR_INVALID_PARAM, "pf must be 1 or 8");
    switch (curve) {
        case BLZ_BLS377: return synth_points_t<Fq_BLS377>(d_out, n, pf, start);
        case BLZ_BLS381: return synth_points_t<Fq_BLS381>(d_out, n, pf, start);
        case BLZ_BN254: return synth_points_t<Fq_BN254>(d_out, n, pf, start);
    }
    return fail(BLZ_ERR_INVALID_PARAM, "unknown curve %d", curve);
}

int blz_synth_field_elements(int device_id, void* d_out, uint64_t n, uint64_t seed) {
    BLZ_TRY(use_device(device_id));
    return synth_scalars_t<Fr_BLS381>(d_out, n, seed, 0);
}

}  // extern "C"
