/* TEST INFRASTRUCTURE: sanitizer self-test of the C oracle.  Built with
 *   gcc -O1 -g -fsanitize=address,undefined blz_oracle.c selftest.c -lpthread
 * and run on the CPU (GPU AddressSanitizer is not available on the pool).  Exercises every
 * exported entry point on small inputs so that out-of-bounds accesses, signed overflow or
 * misaligned loads in the restatement would abort the run. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_point_bytes(int);
int orc_result_bytes(int);
int orc_decode_result(int, const uint8_t*, uint8_t*);
int orc_is_on_curve(int, const uint8_t*);
int orc_generator_mul(int, const uint8_t*, uint8_t*);
int orc_point_mul(int, const uint8_t*, const uint8_t*, uint8_t*);
int orc_point_add(int, const uint8_t*, const uint8_t*, int, uint8_t*);
int orc_precompute_base(int, const uint8_t*, int, uint8_t*);
int orc_msm_naive(int, const uint8_t*, const uint8_t*, uint64_t, int, uint8_t*);
int orc_input_generator(int, uint64_t, int, uint64_t, uint8_t*, uint8_t*, uint8_t*);
int orc_index_weighted_sum(int, const uint8_t*, uint64_t, uint64_t, uint8_t*);
int orc_msm_pippenger(int, const uint8_t*, const uint8_t*, uint64_t, int, int, int, uint8_t*);
int orc_omega(int, int, uint8_t*);
int orc_ntt(int, const uint8_t*, uint8_t*, int, int, int);
int orc_ntt_ex(int, const uint8_t*, uint8_t*, int, int, const uint8_t*, int);
int orc_bitrev_permute(const uint8_t*, uint8_t*, int, int);
int orc_ntt_eval_at(int, const uint8_t*, int, uint64_t, uint8_t*);
int orc_ntt_eval_at_mt(int, const uint8_t*, int, uint64_t, int, uint8_t*);
int orc_dft_naive(int, const uint8_t*, uint8_t*, int);
int orc_ntt_preprocess(const uint8_t*, uint8_t*, uint64_t);
int orc_ntt_postprocess(const uint8_t*, uint8_t*, uint64_t, uint64_t);

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "selftest failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(void) {
    for (int curve = 0; curve < 3; ++curve) {
        int pb = orc_point_bytes(curve), rb = orc_result_bytes(curve);
        for (int pf = 1; pf <= 8; pf += 7) {
            uint64_t n = 300;
            uint8_t* pts = malloc(n * pf * pb);
            uint8_t* sc = malloc(n * 32);
            uint8_t exp[144], got[144], got2[144], xy[96];
            CHECK(orc_input_generator(curve, n, pf, 77 + curve, pts, sc, exp) == 0);
            CHECK(orc_msm_pippenger(curve, pts, sc, n, pf, 3, 0, got) == 0);
            CHECK(memcmp(exp, got, rb) == 0);
            CHECK(orc_msm_naive(curve, pts, sc, 40, pf, got) == 0);
            CHECK(orc_msm_pippenger(curve, pts, sc, 40, pf, 2, 5, got2) == 0);
            CHECK(memcmp(got, got2, rb) == 0);
            CHECK((orc_decode_result(curve, got, xy) & 1) == 1);
            CHECK(orc_is_on_curve(curve, pts) == 1);
            free(pts);
            free(sc);
        }
        uint8_t k[32] = {5}, g5[96], g10[96], s[96], tab[8 * 96], c32[32];
        CHECK(orc_generator_mul(curve, k, g5) == 0);
        CHECK(orc_point_add(curve, g5, g5, 0, g10) == 0);
        k[0] = 2;
        CHECK(orc_point_mul(curve, g5, k, s) == 0);
        CHECK(memcmp(s, g10, pb) == 0);
        CHECK(orc_precompute_base(curve, g5, 8, tab) == 0);
        CHECK(memcmp(tab, g5, pb) == 0);
        uint8_t scal[64] = {0};
        scal[0] = 3; scal[32] = 4;
        CHECK(orc_index_weighted_sum(curve, scal, 2, 0, c32) == 0);
        CHECK(c32[0] == 3 * 1 + 4 * 2);
    }
    {
        enum { LOGN = 6, N = 1 << LOGN };
        uint8_t in[32 * N], out[32 * N], out2[32 * N], back[32 * N], w[32], e[32];
        memset(in, 0, sizeof(in));
        for (int i = 0; i < N; ++i) { in[32 * i] = (uint8_t)(i * 7 + 1); in[32 * i + 9] = (uint8_t)(i ^ 0x5a); }
        CHECK(orc_ntt(1, in, out, LOGN, 0, 2) == 0);
        CHECK(orc_dft_naive(1, in, out2, LOGN) == 0);
        CHECK(memcmp(out, out2, sizeof(out)) == 0);
        CHECK(orc_ntt(1, out, back, LOGN, 1, 1) == 0);
        CHECK(memcmp(back, in, sizeof(in)) == 0);
        CHECK(orc_ntt_eval_at(1, in, LOGN, 5, e) == 0);
        { uint8_t e2[32]; CHECK(orc_ntt_eval_at_mt(1, in, LOGN, 5, 3, e2) == 0); CHECK(memcmp(e, e2, 32) == 0); }
        CHECK(memcmp(e, out + 32 * 5, 32) == 0);
        CHECK(orc_omega(1, 27, w) == 0);
        /* the conventions of orc_ntt_ex: the default root given explicitly = the default transform; bit-reversed buffers are the
         * permuted buffers; a root of the wrong order is refused; the inverse of a convention undoes it */
        uint8_t w6[32], perm[32 * N], o3[32 * N], o4[32 * N], one[32] = {1};
        CHECK(orc_omega(1, LOGN, w6) == 0);
        CHECK(orc_ntt_ex(1, in, o3, LOGN, 0, w6, 2) == 0);
        CHECK(memcmp(o3, out, sizeof(out)) == 0);
        CHECK(orc_bitrev_permute(in, perm, LOGN, 2) == 0);
        CHECK(orc_ntt_ex(1, perm, o3, LOGN, 2 | 4, NULL, 1) == 0);       /* bit-reversed in and out */
        CHECK(orc_bitrev_permute(o3, o4, LOGN, 1) == 0);
        CHECK(memcmp(o4, out, sizeof(out)) == 0);
        CHECK(orc_ntt_ex(1, o3, o4, LOGN, 1 | 2 | 4, w6, 3) == 0);        /* its inverse */
        CHECK(memcmp(o4, perm, sizeof(perm)) == 0);
        CHECK(orc_ntt_ex(1, in, o3, LOGN, 0, one, 1) == -2);              /* 1 is not a primitive 2^6-th root */
    }
    {
        uint64_t n = 4096;
        uint8_t* a = calloc(n, 32);
        uint8_t* b = malloc(n * 32);
        uint8_t* c = malloc(n * 32);
        for (uint64_t i = 0; i < n; ++i) memcpy(a + 32 * i, &i, 8);
        CHECK(orc_ntt_preprocess(a, b, n) == 0);
        CHECK(orc_ntt_postprocess(b, c, n, 1) == 0);
        free(a); free(b); free(c);
    }
    printf("oracle selftest ok\n");
    return 0;
}
