// Dev tool: compare fp_mul_ps against fp_mul_cios on the device for a few inputs, print limbs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../blaze_amd/csrc/field.cuh"
using namespace blz;
template <class P>
__global__ void k(uint32_t* out, uint32_t seed) {
    Fp<P> a, b, r1, r2;
    for (int i = 0; i < P::N; ++i) { a.v[i] = (threadIdx.x * 2654435761u + i * 40503u + seed) | 1u; b.v[i] = a.v[i] * 2246822519u + 7u; }
    a.v[P::N - 1] &= 0x00ffffffu; b.v[P::N - 1] &= 0x00ffffffu;
    fp_mul_cios(r1, a, b);
    fp_mul_ps(r2, a, b);
    for (int i = 0; i < P::N; ++i) { out[threadIdx.x * 4 * P::N + i] = a.v[i]; out[threadIdx.x * 4 * P::N + P::N + i] = b.v[i];
        out[threadIdx.x * 4 * P::N + 2 * P::N + i] = r1.v[i]; out[threadIdx.x * 4 * P::N + 3 * P::N + i] = r2.v[i]; }
}
template <class P> void run(const char* name) {
    uint32_t* d; hipMalloc(&d, 64 * 4 * P::N * 4);
    hipLaunchKernelGGL(k<P>, dim3(1), dim3(64), 0, 0, d, 12345u);
    uint32_t h[64 * 4 * 12]; hipMemcpy(h, d, 64 * 4 * P::N * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) {
        bool eq = true;
        for (int i = 0; i < P::N; ++i) eq &= h[t * 4 * P::N + 2 * P::N + i] == h[t * 4 * P::N + 3 * P::N + i];
        if (!eq && bad++ < 2) {
            printf("%s lane %d mismatch\n a=", name, t); for (int i = P::N - 1; i >= 0; --i) printf("%08x", h[t * 4 * P::N + i]);
            printf("\n b="); for (int i = P::N - 1; i >= 0; --i) printf("%08x", h[t * 4 * P::N + P::N + i]);
            printf("\n cios="); for (int i = P::N - 1; i >= 0; --i) printf("%08x ", h[t * 4 * P::N + 2 * P::N + i]);
            printf("\n ps  ="); for (int i = P::N - 1; i >= 0; --i) printf("%08x ", h[t * 4 * P::N + 3 * P::N + i]);
            printf("\n");
        }
    }
    printf("%s: %d/64 lanes differ\n", name, bad);
    hipFree(d);
}
int main() { run<Fq_BLS381>("Fq_BLS381"); run<Fq_BLS377>("Fq_BLS377"); run<Fq_BN254>("Fq_BN254"); run<Fr_BLS381>("Fr_BLS381"); return 0; }
