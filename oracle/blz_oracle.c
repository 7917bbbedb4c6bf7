/* TEST INFRASTRUCTURE ONLY - plain-C CPU restatement of the reference's MSM/NTT check path.
 *
 * PARITY UNPINNED at byte level against the reference: the reference holds no golden vectors
 * for this path (inputs are thread_rng: tests/msm/mod.rs:66,186,311; NTT goldens are external
 * files: tests/integration_ntt.rs:15-18,78-85) and the arithmetic it checks against is the
 * third-party crate family arkworks 0.3.0 (ark-ec / ark-ff / ark-bls12-381 / ark-bls12-377 /
 * ark-bn254 = "0.3.0", Cargo.toml:14-19), absent from /root/reference and not buildable here
 * (no rustc/cargo, no network).  This file restates the published algorithms those call sites
 * use (short-Weierstrass group law, double-and-add scalar multiplication, canonical
 * little-endian encodings) and is pinned by: published curve constants and known answers
 * (2G of EIP-2537 / EIP-196, r*G = inf), agreement with the independent pure-Python
 * implementation oracle/pyref.py, and the committed vectors under tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Follows (reference file:line):
 *   tests/msm/mod.rs:297-358   input_generator_bls12_381 (and :52-113, :177-233 twins)
 *   tests/msm/mod.rs:360-380   precompute_base_*  (P, 2^32 P, ... 2^224 P, x||y LE canonical)
 *   tests/msm/mod.rs:382-420   result_check_*     (Z|Y|X, x = X/Z, y = Y/Z, on-curve, equality)
 *   tests/msm/mod.rs:88,208,333  acc += aff.mul(scalar)  (one double-and-add per element)
 *   src/ingo_msm/msm_cfg.rs:44-92   sizes 32 / 96(64) / 144(96)
 *   src/ingo_msm/msm_api.rs:155-220 pf=8: element i carries 8 contiguous bases
 *   src/ingo_ntt/ntt_data.rs:65-66,80-156  NTT shape and the 16-bank wire permutation
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

#define MAXL 6

typedef struct {
    int n;          /* 64-bit limbs */
    int nbytes;     /* wire bytes */
    u64 m[MAXL];    /* modulus */
    u64 n0;         /* -m^-1 mod 2^64 */
    u64 one[MAXL];  /* R mod m */
    u64 r2[MAXL];   /* R^2 mod m */
} fctx;

typedef struct {
    fctx fq, fr;
    u64 b[MAXL];             /* curve b, Montgomery */
    u64 gx[MAXL], gy[MAXL];  /* generator, Montgomery */
    int two_adicity;
    u64 root[MAXL];          /* 2^two_adicity-th primitive root in Fr, Montgomery */
} curve_t;

static curve_t g_curves[3];
static int g_init = 0;

/* ------------------------------------------------------------------------------------------
 * multi-precision helpers
 * ------------------------------------------------------------------------------------------ */
static int mp_cmp(const u64* a, const u64* b, int n) {
    for (int i = n - 1; i >= 0; --i) {
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    }
    return 0;
}
static u64 mp_add(u64* r, const u64* a, const u64* b, int n) {
    u128 c = 0;
    for (int i = 0; i < n; ++i) { c += (u128)a[i] + b[i]; r[i] = (u64)c; c >>= 64; }
    return (u64)c;
}
static u64 mp_sub(u64* r, const u64* a, const u64* b, int n) {
    u64 br = 0;
    for (int i = 0; i < n; ++i) {
        u128 d = (u128)a[i] - b[i] - br;
        r[i] = (u64)d;
        br = (u64)(d >> 64) & 1;
    }
    return br;
}
static int mp_is_zero(const u64* a, int n) {
    u64 o = 0;
    for (int i = 0; i < n; ++i) o |= a[i];
    return o == 0;
}
static void mp_from_hex(u64* r, int n, const char* hex) {
    memset(r, 0, sizeof(u64) * n);
    int len = (int)strlen(hex);
    for (int i = 0; i < len; ++i) {
        char ch = hex[len - 1 - i];
        u64 v = (ch >= '0' && ch <= '9') ? (u64)(ch - '0') : (ch >= 'a' && ch <= 'f') ? (u64)(ch - 'a' + 10) : (u64)(ch - 'A' + 10);
        if (i / 16 < n) r[i / 16] |= v << (4 * (i % 16));
    }
}

/* ------------------------------------------------------------------------------------------
 * prime field, Montgomery form (CIOS)
 * ------------------------------------------------------------------------------------------ */
static void f_add(const fctx* f, u64* r, const u64* a, const u64* b) {
    u64 t[MAXL];
    u64 c = mp_add(t, a, b, f->n);
    if (c || mp_cmp(t, f->m, f->n) >= 0) mp_sub(t, t, f->m, f->n);
    memcpy(r, t, sizeof(u64) * f->n);
}
static void f_sub(const fctx* f, u64* r, const u64* a, const u64* b) {
    u64 t[MAXL];
    if (mp_sub(t, a, b, f->n)) mp_add(t, t, f->m, f->n);
    memcpy(r, t, sizeof(u64) * f->n);
}
static void f_neg(const fctx* f, u64* r, const u64* a) {
    if (mp_is_zero(a, f->n)) { memset(r, 0, sizeof(u64) * f->n); return; }
    mp_sub(r, f->m, a, f->n);
}
static void f_mul(const fctx* f, u64* r, const u64* a, const u64* b) {
    const int n = f->n;
    u64 t[MAXL + 2];
    memset(t, 0, sizeof(t));
    for (int i = 0; i < n; ++i) {
        u128 c = 0;
        for (int j = 0; j < n; ++j) {
            c += (u128)a[j] * b[i] + t[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        c += t[n];
        t[n] = (u64)c;
        t[n + 1] = (u64)(c >> 64);
        u64 mm = t[0] * f->n0;
        c = (u128)mm * f->m[0] + t[0];
        c >>= 64;
        for (int j = 1; j < n; ++j) {
            c += (u128)mm * f->m[j] + t[j];
            t[j - 1] = (u64)c;
            c >>= 64;
        }
        c += t[n];
        t[n - 1] = (u64)c;
        t[n] = t[n + 1] + (u64)(c >> 64);
    }
    if (t[n] || mp_cmp(t, f->m, n) >= 0) mp_sub(t, t, f->m, n);
    memcpy(r, t, sizeof(u64) * n);
}
static void f_sqr(const fctx* f, u64* r, const u64* a) { f_mul(f, r, a, a); }
static void f_to_mont(const fctx* f, u64* r, const u64* a) { f_mul(f, r, a, f->r2); }
static void f_from_mont(const fctx* f, u64* r, const u64* a) {
    u64 o[MAXL] = {1, 0, 0, 0, 0, 0};
    f_mul(f, r, a, o);
}
/* r = a^e, e plain little-endian limbs */
static void f_pow(const fctx* f, u64* r, const u64* a, const u64* e, int en) {
    u64 acc[MAXL], base[MAXL];
    memcpy(acc, f->one, sizeof(acc));
    memcpy(base, a, sizeof(u64) * f->n);
    int top = en * 64 - 1;
    while (top >= 0 && !((e[top / 64] >> (top % 64)) & 1)) --top;
    for (int i = top; i >= 0; --i) {
        f_sqr(f, acc, acc);
        if ((e[i / 64] >> (i % 64)) & 1) f_mul(f, acc, acc, base);
    }
    memcpy(r, acc, sizeof(u64) * f->n);
}
static void f_inv(const fctx* f, u64* r, const u64* a) { /* Fermat: a^(m-2) */
    u64 e[MAXL], two[MAXL] = {2, 0, 0, 0, 0, 0};
    mp_sub(e, f->m, two, f->n);
    f_pow(f, r, a, e, f->n);
}
static void f_from_bytes(const fctx* f, u64* r, const uint8_t* b) { /* canonical LE -> Montgomery */
    u64 t[MAXL] = {0};
    for (int i = 0; i < f->nbytes; ++i) t[i / 8] |= (u64)b[i] << (8 * (i % 8));
    /* from_le_bytes_mod_order semantics (tests/msm/mod.rs:397-399) */
    while (mp_cmp(t, f->m, f->n) >= 0) mp_sub(t, t, f->m, f->n);
    f_to_mont(f, r, t);
}
static void f_to_bytes(const fctx* f, uint8_t* b, const u64* a) { /* Montgomery -> canonical LE */
    u64 t[MAXL];
    f_from_mont(f, t, a);
    for (int i = 0; i < f->nbytes; ++i) b[i] = (uint8_t)(t[i / 8] >> (8 * (i % 8)));
}
static void f_init(fctx* f, int n, int nbytes, const char* mhex) {
    memset(f, 0, sizeof(*f));
    f->n = n;
    f->nbytes = nbytes;
    mp_from_hex(f->m, n, mhex);
    u64 inv = 1; /* Newton: inv = m^-1 mod 2^64 */
    for (int i = 0; i < 6; ++i) inv *= 2 - f->m[0] * inv;
    f->n0 = (u64)0 - inv;
    /* one = 2^(64n) mod m by doubling 1, 64n times; r2 likewise 128n times */
    u64 t[MAXL] = {1, 0, 0, 0, 0, 0};
    for (int i = 0; i < 128 * n; ++i) {
        u64 c = mp_add(t, t, t, n);
        if (c || mp_cmp(t, f->m, n) >= 0) mp_sub(t, t, f->m, n);
        if (i == 64 * n - 1) memcpy(f->one, t, sizeof(u64) * n);
    }
    memcpy(f->r2, t, sizeof(u64) * n);
}

/* ------------------------------------------------------------------------------------------
 * G1, Jacobian (X:Y:Z), x = X/Z^2, y = Y/Z^3, Z = 0 <=> infinity.  a = 0 curves.
 * Complete behaviour (inf, P+P, P-P) is handled explicitly: the reference harness repeats a
 * 256-element tile (tests/msm/mod.rs:337-354) so equal/opposite operands are the common case.
 * ------------------------------------------------------------------------------------------ */
typedef struct { u64 x[MAXL], y[MAXL], z[MAXL]; } jac_t;
typedef struct { u64 x[MAXL], y[MAXL]; int inf; } aff_t;

static void j_set_inf(jac_t* p) { memset(p, 0, sizeof(*p)); }
static int j_is_inf(const curve_t* c, const jac_t* p) { return mp_is_zero(p->z, c->fq.n); }

static void j_dbl(const curve_t* c, jac_t* r, const jac_t* p) {
    const fctx* f = &c->fq;
    if (j_is_inf(c, p)) { j_set_inf(r); return; }
    u64 A[MAXL], B[MAXL], C[MAXL], D[MAXL], E[MAXL], F[MAXL], t[MAXL], z3[MAXL];
    f_sqr(f, A, p->x);
    f_sqr(f, B, p->y);
    f_sqr(f, C, B);
    f_add(f, t, p->x, B);
    f_sqr(f, t, t);
    f_sub(f, t, t, A);
    f_sub(f, t, t, C);
    f_add(f, D, t, t);
    f_add(f, E, A, A);
    f_add(f, E, E, A);
    f_sqr(f, F, E);
    f_mul(f, z3, p->y, p->z);
    f_add(f, z3, z3, z3);
    f_sub(f, t, F, D);
    f_sub(f, r->x, t, D);
    f_sub(f, t, D, r->x);
    f_mul(f, t, E, t);
    f_add(f, C, C, C);
    f_add(f, C, C, C);
    f_add(f, C, C, C);
    f_sub(f, r->y, t, C);
    memcpy(r->z, z3, sizeof(z3));
}

static void j_add(const curve_t* c, jac_t* r, const jac_t* p, const jac_t* q) {
    const fctx* f = &c->fq;
    if (j_is_inf(c, p)) { *r = *q; return; }
    if (j_is_inf(c, q)) { *r = *p; return; }
    u64 z1z1[MAXL], z2z2[MAXL], u1[MAXL], u2[MAXL], s1[MAXL], s2[MAXL], h[MAXL], rr[MAXL], t[MAXL];
    f_sqr(f, z1z1, p->z);
    f_sqr(f, z2z2, q->z);
    f_mul(f, u1, p->x, z2z2);
    f_mul(f, u2, q->x, z1z1);
    f_mul(f, s1, p->y, q->z);
    f_mul(f, s1, s1, z2z2);
    f_mul(f, s2, q->y, p->z);
    f_mul(f, s2, s2, z1z1);
    f_sub(f, h, u2, u1);
    f_sub(f, rr, s2, s1);
    if (mp_is_zero(h, f->n)) {
        if (mp_is_zero(rr, f->n)) { j_dbl(c, r, p); return; }
        j_set_inf(r);
        return;
    }
    u64 hh[MAXL], hhh[MAXL], v[MAXL], x3[MAXL], y3[MAXL], z3[MAXL];
    f_sqr(f, hh, h);
    f_mul(f, hhh, hh, h);
    f_mul(f, v, u1, hh);
    f_sqr(f, x3, rr);
    f_sub(f, x3, x3, hhh);
    f_sub(f, x3, x3, v);
    f_sub(f, x3, x3, v);
    f_sub(f, t, v, x3);
    f_mul(f, y3, rr, t);
    f_mul(f, t, s1, hhh);
    f_sub(f, y3, y3, t);
    f_mul(f, z3, p->z, q->z);
    f_mul(f, z3, z3, h);
    memcpy(r->x, x3, sizeof(x3));
    memcpy(r->y, y3, sizeof(y3));
    memcpy(r->z, z3, sizeof(z3));
}

static void j_from_aff(const curve_t* c, jac_t* r, const aff_t* a) {
    if (a->inf) { j_set_inf(r); return; }
    memcpy(r->x, a->x, sizeof(a->x));
    memcpy(r->y, a->y, sizeof(a->y));
    memcpy(r->z, c->fq.one, sizeof(r->z));
}
static void j_madd(const curve_t* c, jac_t* r, const jac_t* p, const aff_t* a) {
    jac_t q;
    j_from_aff(c, &q, a);
    j_add(c, r, p, &q);
}
static void j_to_aff(const curve_t* c, aff_t* a, const jac_t* p) {
    const fctx* f = &c->fq;
    memset(a, 0, sizeof(*a));
    if (j_is_inf(c, p)) { a->inf = 1; return; }
    u64 zi[MAXL], zi2[MAXL], zi3[MAXL];
    f_inv(f, zi, p->z);
    f_sqr(f, zi2, zi);
    f_mul(f, zi3, zi2, zi);
    f_mul(f, a->x, p->x, zi2);
    f_mul(f, a->y, p->y, zi3);
}
/* k: plain LE limbs (kn of them).  Double-and-add from the top bit, as ark-ec 0.3.0's
 * AffineCurve::mul does (mul_bits over BitIteratorBE). */
static void j_mul(const curve_t* c, jac_t* r, const aff_t* a, const u64* k, int kn) {
    jac_t acc;
    j_set_inf(&acc);
    int top = kn * 64 - 1;
    while (top >= 0 && !((k[top / 64] >> (top % 64)) & 1)) --top;
    for (int i = top; i >= 0; --i) {
        j_dbl(c, &acc, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) j_madd(c, &acc, &acc, a);
    }
    *r = acc;
}

static int aff_on_curve(const curve_t* c, const aff_t* a) {
    const fctx* f = &c->fq;
    if (a->inf) return 1;
    u64 l[MAXL], r[MAXL];
    f_sqr(f, l, a->y);
    f_sqr(f, r, a->x);
    f_mul(f, r, r, a->x);
    f_add(f, r, r, c->b);
    return mp_cmp(l, r, f->n) == 0;
}
static void aff_from_bytes(const curve_t* c, aff_t* a, const uint8_t* b) { /* x || y (mod.rs:363-366) */
    a->inf = 0;
    memset(a->x, 0, sizeof(a->x));
    memset(a->y, 0, sizeof(a->y));
    f_from_bytes(&c->fq, a->x, b);
    f_from_bytes(&c->fq, a->y, b + c->fq.nbytes);
}
static void aff_to_bytes(const curve_t* c, uint8_t* b, const aff_t* a) {
    f_to_bytes(&c->fq, b, a->x);
    f_to_bytes(&c->fq, b + c->fq.nbytes, a->y);
}
static void scalar_from_bytes(u64* k, const uint8_t* b) {
    for (int i = 0; i < 4; ++i) {
        k[i] = 0;
        for (int j = 0; j < 8; ++j) k[i] |= (u64)b[8 * i + j] << (8 * j);
    }
}

/* ------------------------------------------------------------------------------------------
 * curve table (published parameters)
 * ------------------------------------------------------------------------------------------ */
static void curve_init(curve_t* c, int fqn, int fqbytes, const char* q, const char* r, u64 b, const char* gx,
                       const char* gy, int two_adicity, u64 fr_gen) {
    f_init(&c->fq, fqn, fqbytes, q);
    f_init(&c->fr, 4, 32, r);
    u64 t[MAXL] = {b, 0, 0, 0, 0, 0};
    f_to_mont(&c->fq, c->b, t);
    mp_from_hex(t, fqn, gx);
    f_to_mont(&c->fq, c->gx, t);
    mp_from_hex(t, fqn, gy);
    f_to_mont(&c->fq, c->gy, t);
    c->two_adicity = two_adicity;
    /* root = gen^((r-1)/2^s) */
    u64 e[MAXL] = {0}, g[MAXL] = {fr_gen, 0, 0, 0, 0, 0}, gm[MAXL], onep[MAXL] = {1, 0, 0, 0, 0, 0};
    mp_sub(e, c->fr.m, onep, 4);
    for (int i = 0; i < two_adicity; ++i) { /* e >>= 1 */
        for (int j = 0; j < 4; ++j) e[j] = (e[j] >> 1) | (j < 3 ? e[j + 1] << 63 : 0);
    }
    f_to_mont(&c->fr, gm, g);
    f_pow(&c->fr, c->root, gm, e, 4);
}
static void ensure_init(void) {
    if (g_init) return;
    /* order of enum Curve: src/ingo_msm/msm_cfg.rs:4-8 */
    curve_init(&g_curves[0], 6, 48,
               "01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001",
               "12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001", 1,
               "008848defe740a67c8fc6225bf87ff5485951e2caa9d41bb188282c8bd37cb5cd5481512ffcd394eeab9b16eb21be9ef",
               "01914a69c5102eff1f674f5d30afeec4bd7fb348ca3e52d96d182ad44fb82305c2fe3d3634a9591afd82de55559c8ea6", 47, 22);
    curve_init(&g_curves[1], 6, 48,
               "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab",
               "73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001", 4,
               "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb",
               "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1", 32, 7);
    curve_init(&g_curves[2], 4, 32, "30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47",
               "30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001", 3, "1", "2", 28, 5);
    g_init = 1;
}
static const curve_t* get_curve(int id) {
    ensure_init();
    return (id >= 0 && id < 3) ? &g_curves[id] : NULL;
}

/* ------------------------------------------------------------------------------------------
 * exported API (ctypes).  All return 0 on success.
 * ------------------------------------------------------------------------------------------ */
int orc_point_bytes(int curve) { const curve_t* c = get_curve(curve); return c ? 2 * c->fq.nbytes : -1; }
int orc_result_bytes(int curve) { const curve_t* c = get_curve(curve); return c ? 3 * c->fq.nbytes : -1; }

/* result encoding of this build: Z=1 | y | x, infinity = Z=0 | Y=1 | X=0 */
static void encode_result(const curve_t* c, uint8_t* out, const jac_t* p) {
    aff_t a;
    j_to_aff(c, &a, p);
    int fb = c->fq.nbytes;
    memset(out, 0, 3 * fb);
    if (a.inf) { out[fb] = 1; return; }
    out[0] = 1;
    f_to_bytes(&c->fq, out + fb, a.y);
    f_to_bytes(&c->fq, out + 2 * fb, a.x);
}

/* tests/msm/mod.rs:397-405.  out_xy = affine x||y canonical; returns flags: bit0 on_curve, bit1 infinity */
int orc_decode_result(int curve, const uint8_t* res, uint8_t* out_xy) {
    const curve_t* c = get_curve(curve);
    if (!c) return -1;
    const fctx* f = &c->fq;
    int fb = f->nbytes;
    u64 X[MAXL], Y[MAXL], Z[MAXL], zi[MAXL];
    f_from_bytes(f, Z, res);
    f_from_bytes(f, Y, res + fb);
    f_from_bytes(f, X, res + 2 * fb);
    memset(out_xy, 0, 2 * fb);
    if (mp_is_zero(Z, f->n)) return 2 | 1;
    f_inv(f, zi, Z);
    aff_t a;
    memset(&a, 0, sizeof(a));
    f_mul(f, a.x, X, zi);
    f_mul(f, a.y, Y, zi);
    aff_to_bytes(c, out_xy, &a);
    return aff_on_curve(c, &a) ? 1 : 0;
}

int orc_is_on_curve(int curve, const uint8_t* xy) {
    const curve_t* c = get_curve(curve);
    if (!c) return -1;
    aff_t a;
    aff_from_bytes(c, &a, xy);
    return aff_on_curve(c, &a);
}

/* k*G as x||y (k: 32-byte LE).  Returns 1 if infinity. */
int orc_generator_mul(int curve, const uint8_t* k32, uint8_t* out_xy) {
    const curve_t* c = get_curve(curve);
    if (!c) return -1;
    aff_t g;
    memset(&g, 0, sizeof(g));
    memcpy(g.x, c->gx, sizeof(g.x));
    memcpy(g.y, c->gy, sizeof(g.y));
    u64 k[4];
    scalar_from_bytes(k, k32);
    jac_t r;
    j_mul(c, &r, &g, k, 4);
    aff_t a;
    j_to_aff(c, &a, &r);
    memset(out_xy, 0, 2 * c->fq.nbytes);
    if (a.inf) return 1;
    aff_to_bytes(c, out_xy, &a);
    return 0;
}

/* k*P as x||y. Returns 1 if infinity. */
int orc_point_mul(int curve, const uint8_t* xy, const uint8_t* k32, uint8_t* out_xy) {
    const curve_t* c = get_curve(curve);
    if (!c) return -1;
    aff_t p;
    aff_from_bytes(c, &p, xy);
    u64 k[4];
    scalar_from_bytes(k, k32);
    jac_t r;
    j_mul(c, &r, &p, k, 4);
    aff_t a;
    j_to_aff(c, &a, &r);
    memset(out_xy, 0, 2 * c->fq.nbytes);
    if (a.inf) return 1;
    aff_to_bytes(c, out_xy, &a);
    return 0;
}

/* P+Q (affine bytes in/out); flags in: bit0 P inf, bit1 Q inf.  Returns 1 if result infinity. */
int orc_point_add(int curve, const uint8_t* p_xy, const uint8_t* q_xy, int inf_flags, uint8_t* out_xy) {
    const curve_t* c = get_curve(curve);
    if (!c) return -1;
    aff_t p, q;
    aff_from_bytes(c, &p, p_xy);
    aff_from_bytes(c, &q, q_xy);
    p.inf = inf_flags & 1;
    q.inf = (inf_flags >> 1) & 1;
    jac_t jp, r;
    j_from_aff(c, &jp, &p);
    j_madd(c, &r, &jp, &q);
    aff_t a;
    j_to_aff(c, &a, &r);
    memset(out_xy, 0, 2 * c->fq.nbytes);
    if (a.inf) return 1;
    aff_to_bytes(c, out_xy, &a);
    return 0;
}

/* tests/msm/mod.rs:360-380: P, 2^32 P, ..., 2^(32(pf-1)) P contiguous, x||y each. */
int orc_precompute_base(int curve, const uint8_t* xy, int pf, uint8_t* out) {
    const curve_t* c = get_curve(curve);
    if (!c) return -1;
    int pb = 2 * c->fq.nbytes;
    aff_t p;
    aff_from_bytes(c, &p, xy);
    memcpy(out, xy, pb);
    for (int i = 1; i < pf; ++i) {
        /* coeff = Fr::from(2^(32 i)) (mod.rs:370): 2^(32i) < r for i < 8 on all three curves */
        u64 k[4] = {0, 0, 0, 0};
        k[(32 * i) / 64] = (u64)1 << ((32 * i) % 64);
        jac_t r;
        j_mul(c, &r, &p, k, 4);
        aff_t a;
        j_to_aff(c, &a, &r);
        aff_to_bytes(c, out + i * pb, &a);
    }
    return 0;
}

/* Device task semantics with the reference's CPU structure: one double-and-add
 * per element, accumulated in order (tests/msm/mod.rs:326-335).  pf = 1 or 8. */
int orc_msm_naive(int curve, const uint8_t* points, const uint8_t* scalars, u64 n, int pf, uint8_t* out_result) {
    const curve_t* c = get_curve(curve);
    if (!c) return -1;
    int pb = 2 * c->fq.nbytes;
    jac_t acc, t;
    j_set_inf(&acc);
    for (u64 i = 0; i < n; ++i) {
        u64 k[4];
        scalar_from_bytes(k, scalars + 32 * i);
        if (pf == 1) {
            aff_t p;
            aff_from_bytes(c, &p, points + pb * i);
            j_mul(c, &t, &p, k, 4);
            j_add(c, &acc, &acc, &t);
        } else {
            for (int j = 0; j < pf; ++j) {
                u64 kj[1] = {(k[j / 2] >> (32 * (j % 2))) & 0xffffffffu};
                aff_t p;
                aff_from_bytes(c, &p, points + pb * (i * pf + j));
                j_mul(c, &t, &p, kj, 1);
                j_add(c, &acc, &acc, &t);
            }
        }
    }
    encode_result(c, out_result, &acc);
    return 0;
}

/* ---- seeded harness generator (tests/msm/mod.rs:297-358) ---- */
static u64 splitmix64(u64* s) {
    u64 z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static void rand_scalar(const curve_t* c, u64* st, u64* k) { /* uniform in [0, r) by rejection */
    int topbits = 0;
    u64 top = c->fr.m[3];
    while (top) { ++topbits; top >>= 1; }
    for (;;) {
        for (int i = 0; i < 4; ++i) k[i] = splitmix64(st);
        if (topbits < 64) k[3] &= (((u64)1 << topbits) - 1);
        if (mp_cmp(k, c->fr.m, 4) < 0) return;
    }
}
/* Generates the <=256-element tile and repeats it like the reference generator.
 * points: n*pf*pb bytes, scalars: n*32 bytes, expected: result bytes (Z=1|y|x) of
 * floor(n/256)*S_256 + S_(n%256)  (mod.rs:388-395), computed through the running sums. */
static int input_generator_impl(int curve, u64 n, int pf, u64 seed, uint8_t* points, uint8_t* scalars, uint8_t* expected,
                                int tile_only) {
    const curve_t* c = get_curve(curve);
    if (!c) return -1;
    int pb = 2 * c->fq.nbytes;
    u64 st = seed;
    u64 tile = n > 256 ? 256 : n;
    jac_t acc;
    j_set_inf(&acc);
    jac_t* running = (jac_t*)malloc(sizeof(jac_t) * (tile ? tile : 1));
    aff_t g;
    memset(&g, 0, sizeof(g));
    memcpy(g.x, c->gx, sizeof(g.x));
    memcpy(g.y, c->gy, sizeof(g.y));
    for (u64 i = 0; i < tile; ++i) {
        u64 k[4], s[4];
        do { rand_scalar(c, &st, k); } while (mp_is_zero(k, 4));
        jac_t pj;
        j_mul(c, &pj, &g, k, 4); /* random r-torsion point (G1Projective::rand, mod.rs:327) */
        aff_t p;
        j_to_aff(c, &p, &pj);
        uint8_t xy[96];
        aff_to_bytes(c, xy, &p);
        orc_precompute_base(curve, xy, pf, points + i * pf * pb);
        rand_scalar(c, &st, s);
        for (int b = 0; b < 32; ++b) scalars[32 * i + b] = (uint8_t)(s[b / 8] >> (8 * (b % 8)));
        jac_t t;
        j_mul(c, &t, &p, s, 4);
        j_add(c, &acc, &acc, &t);
        running[i] = acc;
    }
    if (n > 256) {
        u64 mult = n / 256, rest = n % 256;
        if (!tile_only) {
            for (u64 m = 1; m < mult; ++m) {
                memcpy(points + m * 256 * pf * pb, points, 256 * (size_t)pf * pb);
                memcpy(scalars + m * 256 * 32, scalars, 256 * 32);
            }
            memcpy(points + mult * 256 * pf * pb, points, rest * (size_t)pf * pb);
            memcpy(scalars + mult * 256 * 32, scalars, rest * 32);
        }
        aff_t s256;
        j_to_aff(c, &s256, &running[255]);
        u64 km[1] = {mult};
        jac_t e;
        if (s256.inf) j_set_inf(&e); else j_mul(c, &e, &s256, km, 1);
        if (rest) j_add(c, &e, &e, &running[rest - 1]);
        acc = e;
    }
    encode_result(c, expected, &acc);
    free(running);
    return 0;
}
int orc_input_generator(int curve, u64 n, int pf, u64 seed, uint8_t* points, uint8_t* scalars, uint8_t* expected) {
    return input_generator_impl(curve, n, pf, seed, points, scalars, expected, 0);
}
/* Same generator, but only the (<= 256-element) tile is written: points = min(n,256)*pf*pb bytes, scalars =
 * min(n,256)*32 bytes; `expected` is still the result for all n elements.  For the reference's largest shape
 * (n = 2^26, pf = 8: 48 GiB of points, tests/integration_msm.rs:386-467) the caller repeats the tile itself. */
int orc_input_tile(int curve, u64 n, int pf, u64 seed, uint8_t* points, uint8_t* scalars, uint8_t* expected) {
    return input_generator_impl(curve, n, pf, seed, points, scalars, expected, 1);
}

/* ------------------------------------------------------------------------------------------
 * Linearity helper for full-size checks: the build's synthetic point set is P_i = (start+i+1)*G
 * so  sum s_i P_i = (sum s_i (start+i+1) mod r) G.  Returns the 32-byte LE canonical coefficient.
 * For pf=8 bases B_{i,j} = 2^(32 j) P_i the same coefficient applies (sum_j s_{i,j} 2^(32j) = s_i).
 * ------------------------------------------------------------------------------------------ */
static void index_weighted_sum_range(const curve_t* c, const uint8_t* scalars, u64 lo, u64 hi, u64 start, u64* acc) {
    const fctx* f = &c->fr;
    memset(acc, 0, 32);
    for (u64 i = lo; i < hi; ++i) {
        u64 s[4], sm[MAXL] = {0}, w[MAXL] = {start + i + 1, 0, 0, 0, 0, 0}, wm[MAXL], p[MAXL];
        scalar_from_bytes(s, scalars + 32 * i);
        while (mp_cmp(s, f->m, 4) >= 0) mp_sub(s, s, f->m, 4);
        memcpy(sm, s, sizeof(s));
        f_to_mont(f, wm, w);
        f_mul(f, p, sm, wm); /* = s*w plain (one operand Montgomery) */
        f_add(f, acc, acc, p);
    }
}
typedef struct { const curve_t* c; const uint8_t* scalars; u64 lo, hi, start; u64 acc[4]; } iws_job;
static void* iws_worker(void* arg) {
    iws_job* J = (iws_job*)arg;
    index_weighted_sum_range(J->c, J->scalars, J->lo, J->hi, J->start, J->acc);
    return NULL;
}
/* threaded form (full-size checks of bench.py: 2^26 scalars) */
int orc_index_weighted_sum_mt(int curve, const uint8_t* scalars, u64 n, u64 start, int threads, uint8_t* out32) {
    const curve_t* c = get_curve(curve);
    if (!c) return -1;
    if (threads < 1) threads = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    iws_job* jobs = (iws_job*)malloc(sizeof(iws_job) * threads);
    for (int t = 0; t < threads; ++t) {
        jobs[t].c = c; jobs[t].scalars = scalars; jobs[t].lo = n * t / threads; jobs[t].hi = n * (t + 1) / threads; jobs[t].start = start;
        pthread_create(&th[t], NULL, iws_worker, &jobs[t]);
    }
    u64 acc[4] = {0, 0, 0, 0};
    for (int t = 0; t < threads; ++t) { pthread_join(th[t], NULL); f_add(&c->fr, acc, acc, jobs[t].acc); }
    for (int b = 0; b < 32; ++b) out32[b] = (uint8_t)(acc[b / 8] >> (8 * (b % 8)));
    free(th); free(jobs);
    return 0;
}
int orc_index_weighted_sum(int curve, const uint8_t* scalars, u64 n, u64 start, uint8_t* out32) {
    const curve_t* c = get_curve(curve);
    if (!c) return -1;
    const fctx* f = &c->fr;
    u64 acc[4] = {0, 0, 0, 0};
    for (u64 i = 0; i < n; ++i) {
        u64 s[4], sm[MAXL] = {0}, w[MAXL] = {start + i + 1, 0, 0, 0, 0, 0}, wm[MAXL], p[MAXL];
        scalar_from_bytes(s, scalars + 32 * i);
        while (mp_cmp(s, f->m, 4) >= 0) mp_sub(s, s, f->m, 4);
        memcpy(sm, s, sizeof(s));
        f_to_mont(f, wm, w);
        f_mul(f, p, sm, wm); /* = s*w plain (one operand Montgomery) */
        f_add(f, acc, acc, p);
    }
    for (int b = 0; b < 32; ++b) out32[b] = (uint8_t)(acc[b / 8] >> (8 * (b % 8)));
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * CPU Pippenger (baseline only; pthreads).  Signed c-bit windows, Jacobian buckets.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const curve_t* c;
    u64 n;
    int pf, cbits, nwin, sbits, nchunks;
    const uint8_t* points; const uint8_t* scalars;
    aff_t* aff;      /* decoded points (n*pf) */
    int32_t* digits; /* [npts][nwin] signed window digits */
    jac_t* win_sum;  /* [nwin][nchunks] */
    int next_task, nthreads;
    pthread_mutex_t mu;
} pip_t;
typedef struct { pip_t* P; int tid; } pip_arg;

/* round 1: decode the points and recode the scalars, one contiguous slice of the elements per thread */
static void* pip_decode_worker(void* arg) {
    pip_t* P = ((pip_arg*)arg)->P;
    const int tid = ((pip_arg*)arg)->tid;
    const curve_t* c = P->c;
    const int pb = 2 * c->fq.nbytes, nwin = P->nwin, cbits = P->cbits;
    u64 npts = P->n * P->pf;
    u64 lo = npts * tid / P->nthreads, hi = npts * (tid + 1) / P->nthreads;
    for (u64 i = lo; i < hi; ++i) {
        aff_from_bytes(c, &P->aff[i], P->points + pb * i);
        u64 k[5] = {0, 0, 0, 0, 0};
        if (P->pf == 1) scalar_from_bytes(k, P->scalars + 32 * i);
        else { const uint8_t* sp = P->scalars + 4 * i; k[0] = (u64)sp[0] | ((u64)sp[1] << 8) | ((u64)sp[2] << 16) | ((u64)sp[3] << 24); }
        int carry = 0;
        for (int w = 0; w < nwin; ++w) {
            int bit = w * cbits;
            u64 v = (k[bit / 64] >> (bit % 64));
            if (bit % 64 + cbits > 64) v |= k[bit / 64 + 1] << (64 - bit % 64);
            int64_t d = (int64_t)(v & (((u64)1 << cbits) - 1)) + carry;
            carry = 0;
            if (d > ((int64_t)1 << (cbits - 1))) { d -= (int64_t)1 << cbits; carry = 1; }
            P->digits[i * nwin + w] = (int32_t)d;
        }
    }
    return NULL;
}
/* round 2: tasks (window, chunk of the elements) from a shared counter: bucket sums of the chunk, then
 * sum_b b * S_b by running sums */
static void* pip_worker(void* arg) {
    pip_t* P = ((pip_arg*)arg)->P;
    const curve_t* c = P->c;
    u64 npts = P->n * P->pf;
    u64 nb = (u64)1 << (P->cbits - 1);
    jac_t* buckets = (jac_t*)malloc(sizeof(jac_t) * (nb + 1));
    for (;;) {
        pthread_mutex_lock(&P->mu);
        int task = P->next_task++;
        pthread_mutex_unlock(&P->mu);
        if (task >= P->nwin * P->nchunks) break;
        const int w = task / P->nchunks, ch = task % P->nchunks;
        u64 lo = npts * ch / P->nchunks, hi = npts * (ch + 1) / P->nchunks;
        memset(buckets, 0, sizeof(jac_t) * (nb + 1));
        for (u64 i = lo; i < hi; ++i) {
            int32_t d = P->digits[i * P->nwin + w];
            if (d == 0) continue;
            if (d > 0) {
                j_madd(c, &buckets[d], &buckets[d], &P->aff[i]);
            } else {
                aff_t m = P->aff[i];
                f_neg(&c->fq, m.y, m.y);
                j_madd(c, &buckets[-d], &buckets[-d], &m);
            }
        }
        jac_t run, sum;
        j_set_inf(&run);
        j_set_inf(&sum);
        for (u64 b = nb; b >= 1; --b) {
            j_add(c, &run, &run, &buckets[b]);
            j_add(c, &sum, &sum, &run);
        }
        P->win_sum[(size_t)w * P->nchunks + ch] = sum;
    }
    free(buckets);
    return NULL;
}

int orc_msm_pippenger(int curve, const uint8_t* points, const uint8_t* scalars, u64 n, int pf, int threads,
                      int cbits, uint8_t* out_result) {
    const curve_t* c = get_curve(curve);
    if (!c) return -1;
    u64 npts = n * pf;
    int sbits = pf == 1 ? 256 : 32;
    if (cbits <= 0) {
        cbits = 4;
        while (((u64)1 << (cbits + 4)) < npts && cbits < 16) ++cbits;
    }
    if (threads < 1) threads = 1;
    int nwin = (sbits + 1 + cbits - 1) / cbits;
    pip_t P;
    memset(&P, 0, sizeof(P));
    P.c = c; P.n = n; P.pf = pf; P.cbits = cbits; P.nwin = nwin; P.sbits = sbits; P.nthreads = threads;
    P.points = points; P.scalars = scalars;
    /* enough (window, chunk) tasks to keep every thread busy, but chunks long enough that the bucket reduce
     * (2 * 2^(c-1) additions per task) stays small beside the accumulation */
    P.nchunks = 1;
    while (P.nchunks * nwin < 3 * threads && npts / (P.nchunks * 2) >= ((u64)8 << cbits)) P.nchunks *= 2;
    P.aff = (aff_t*)malloc(sizeof(aff_t) * (npts ? npts : 1));
    P.digits = (int32_t*)malloc(sizeof(int32_t) * (npts ? npts : 1) * nwin);
    P.win_sum = (jac_t*)malloc(sizeof(jac_t) * nwin * P.nchunks);
    pthread_mutex_init(&P.mu, NULL);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    pip_arg* args = (pip_arg*)malloc(sizeof(pip_arg) * threads);
    for (int t = 0; t < threads; ++t) { args[t].P = &P; args[t].tid = t; pthread_create(&th[t], NULL, pip_decode_worker, &args[t]); }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, pip_worker, &args[t]);
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    jac_t acc;
    j_set_inf(&acc);
    for (int w = nwin - 1; w >= 0; --w) {
        for (int d = 0; d < cbits; ++d) j_dbl(c, &acc, &acc);
        for (int ch = 0; ch < P.nchunks; ++ch) j_add(c, &acc, &acc, &P.win_sum[(size_t)w * P.nchunks + ch]);
    }
    encode_result(c, out_result, &acc);
    free(th); free(args); free(P.win_sum); free(P.aff); free(P.digits);
    pthread_mutex_destroy(&P.mu);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * NTT over Fr: forward X[k] = sum_j x[j] w^(jk), natural order in/out, w = root^(2^(s-logn)).
 * (field/direction/order are this build's definition: the reference states none)
 * ------------------------------------------------------------------------------------------ */
int orc_omega(int curve, int logn, uint8_t* out32) {
    const curve_t* c = get_curve(curve);
    if (!c || logn > c->two_adicity) return -1;
    u64 w[MAXL];
    memcpy(w, c->root, sizeof(w));
    for (int i = 0; i < c->two_adicity - logn; ++i) f_sqr(&c->fr, w, w);
    f_to_bytes(&c->fr, out32, w);
    return 0;
}

typedef struct { const curve_t* c; u64* a; u64 n; u64 len; const u64* tw; int tid, nth; } ntt_job;
static void* ntt_stage_worker(void* arg) {
    ntt_job* J = (ntt_job*)arg;
    const fctx* f = &J->c->fr;
    u64 half = J->len / 2, nb = J->n / 2, step = J->n / J->len;
    u64 t0 = nb * J->tid / J->nth, t1 = nb * (J->tid + 1) / J->nth;
    for (u64 t = t0; t < t1; ++t) {
        u64 blk = t / half, k = t % half;
        u64* u = J->a + 4 * (blk * J->len + k);
        u64* v = u + 4 * half;
        u64 x[MAXL], s[MAXL], d[MAXL];
        f_mul(f, x, v, J->tw + 4 * (k * step));
        f_add(f, s, u, x);
        f_sub(f, d, u, x);
        memcpy(u, s, 32);
        memcpy(v, d, 32);
    }
    return NULL;
}
/* the three element-wise sweeps of orc_ntt (bit-reversed load, twiddle table, scaled store), split over threads */
typedef struct { const curve_t* c; int phase, logn, tid, nth; u64 n; const uint8_t* in; uint8_t* out; u64* a; u64* tw;
                 const u64* w; const u64* ninv; int inverse; int brin, brout; } ntt_sweep;
static u64 bitrev_u64(u64 i, int logn) {
    u64 rev = 0;
    for (int b = 0; b < logn; ++b) rev |= ((i >> b) & 1) << (logn - 1 - b);
    return rev;
}
static void* ntt_sweep_worker(void* arg) {
    ntt_sweep* S = (ntt_sweep*)arg;
    const fctx* f = &S->c->fr;
    if (S->phase == 0) {
        u64 lo = S->n * S->tid / S->nth, hi = S->n * (S->tid + 1) / S->nth;
        for (u64 i = lo; i < hi; ++i) {
            /* the decimation-in-time stages want a[bitrev(i)] = x[i]; a buffer in bit-reversed order holds x[bitrev(p)] at p */
            u64 rev = S->brin ? i : bitrev_u64(i, S->logn);
            u64 t[MAXL];
            f_from_bytes(f, t, S->in + 32 * i);
            memcpy(S->a + 4 * rev, t, 32);
        }
    } else if (S->phase == 1) {
        u64 h = S->n / 2, lo = h * S->tid / S->nth, hi = h * (S->tid + 1) / S->nth;
        if (lo < hi) {
            /* cur = w^lo by square-and-multiply, then step */
            u64 cur[MAXL], base[MAXL];
            memcpy(cur, f->one, sizeof(cur));
            memcpy(base, S->w, sizeof(base));
            for (u64 e = lo; e; e >>= 1) {
                if (e & 1) f_mul(f, cur, cur, base);
                f_sqr(f, base, base);
            }
            for (u64 i = lo; i < hi; ++i) { memcpy(S->tw + 4 * i, cur, 32); f_mul(f, cur, cur, S->w); }
        }
    } else {
        u64 lo = S->n * S->tid / S->nth, hi = S->n * (S->tid + 1) / S->nth;
        for (u64 i = lo; i < hi; ++i) {
            u64 t[MAXL];
            memcpy(t, S->a + 4 * i, 32);
            if (S->inverse) f_mul(f, t, t, S->ninv);
            f_to_bytes(f, S->out + 32 * (S->brout ? bitrev_u64(i, S->logn) : i), t);
        }
    }
    return NULL;
}
/* The transform under a caller-chosen convention (the device's blz_ntt_new_ex3; the reference states none: ntt_api.rs:8-23):
 * flags bit 0 inverse (w^-1, times n^-1), bit 1 input in bit-reversed order, bit 2 output in bit-reversed order; root32 (nullable):
 * a primitive 2^logn-th root of unity, canonical LE, instead of the generator's.  -2: root^(n/2) != -1. */
int orc_ntt_ex(int curve, const uint8_t* in, uint8_t* out, int logn, int flags, const uint8_t* root32, int threads);
int orc_ntt(int curve, const uint8_t* in, uint8_t* out, int logn, int inverse, int threads) {
    return orc_ntt_ex(curve, in, out, logn, inverse ? 1 : 0, NULL, threads);
}
int orc_ntt_ex(int curve, const uint8_t* in, uint8_t* out, int logn, int flags, const uint8_t* root32, int threads) {
    const curve_t* c = get_curve(curve);
    if (!c || logn > c->two_adicity || logn < 1) return -1;
    const fctx* f = &c->fr;
    const int inverse = flags & 1, brin = (flags >> 1) & 1, brout = (flags >> 2) & 1;
    u64 n = (u64)1 << logn;
    if (threads < 1) threads = 1;
    u64* a = (u64*)malloc(32 * n);
    u64* tw = (u64*)malloc(32 * (n / 2 ? n / 2 : 1));
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    ntt_job* jobs = (ntt_job*)malloc(sizeof(ntt_job) * threads);
    ntt_sweep* sw = (ntt_sweep*)malloc(sizeof(ntt_sweep) * threads);
    u64 w[MAXL];
    if (root32) {
        u64 t[MAXL], m1[MAXL], zero[MAXL];
        f_from_bytes(f, w, root32);
        memcpy(t, w, sizeof(t));
        for (int i = 0; i + 1 < logn; ++i) f_sqr(f, t, t);
        memset(zero, 0, sizeof(zero));
        f_sub(f, m1, zero, f->one);
        if (memcmp(t, m1, 8 * (size_t)f->n) != 0) { free(a); free(tw); free(th); free(jobs); free(sw); return -2; }
    } else {
        memcpy(w, c->root, sizeof(w));
        for (int i = 0; i < c->two_adicity - logn; ++i) f_sqr(f, w, w);
    }
    if (inverse) f_inv(f, w, w);
    u64 ninv[MAXL];
    memset(ninv, 0, sizeof(ninv));
    if (inverse) {
        u64 nn[MAXL] = {n, 0, 0, 0, 0, 0};
        f_to_mont(f, ninv, nn);
        f_inv(f, ninv, ninv);
    }
    for (int phase = 0; phase < 2; ++phase) {   /* x[bitrev(i)] into a; tw[i] = w^i */
        for (int t = 0; t < threads; ++t) {
            sw[t] = (ntt_sweep){c, phase, logn, t, threads, n, in, out, a, tw, w, ninv, inverse, brin, brout};
            pthread_create(&th[t], NULL, ntt_sweep_worker, &sw[t]);
        }
        for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    }
    for (u64 len = 2; len <= n; len <<= 1) {
        for (int t = 0; t < threads; ++t) {
            jobs[t] = (ntt_job){c, a, n, len, tw, t, threads};
            pthread_create(&th[t], NULL, ntt_stage_worker, &jobs[t]);
        }
        for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    }
    for (int t = 0; t < threads; ++t) {
        sw[t] = (ntt_sweep){c, 2, logn, t, threads, n, in, out, a, tw, w, ninv, inverse, brin, brout};
        pthread_create(&th[t], NULL, ntt_sweep_worker, &sw[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    free(a); free(tw); free(th); free(jobs); free(sw);
    return 0;
}
/* out[bitrev(i)] = in[i] over logn bits (32-byte elements): the order blz_ntt_new_ex3's BITREV flags speak of */
typedef struct { const uint8_t* in; uint8_t* out; int logn, tid, nth; } brp_job;
static void* brp_worker(void* arg) {
    brp_job* J = (brp_job*)arg;
    u64 n = (u64)1 << J->logn, lo = n * J->tid / J->nth, hi = n * (J->tid + 1) / J->nth;
    for (u64 i = lo; i < hi; ++i) memcpy(J->out + 32 * bitrev_u64(i, J->logn), J->in + 32 * i, 32);
    return NULL;
}
int orc_bitrev_permute(const uint8_t* in, uint8_t* out, int logn, int threads) {
    if (logn < 0 || logn > 40 || in == out) return -1;
    if (threads < 1) threads = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    brp_job* jobs = (brp_job*)malloc(sizeof(brp_job) * threads);
    for (int t = 0; t < threads; ++t) {
        jobs[t] = (brp_job){in, out, logn, t, threads};
        pthread_create(&th[t], NULL, brp_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs);
    return 0;
}

/* X[k] = sum_i x[i] (w^k)^i by Horner: one output coefficient of a transform of any size, O(n).
 * Used to spot-check full-size (2^27) device transforms. */
int orc_ntt_eval_at(int curve, const uint8_t* in, int logn, u64 k, uint8_t* out32) {
    const curve_t* c = get_curve(curve);
    if (!c || logn > c->two_adicity) return -1;
    const fctx* f = &c->fr;
    u64 n = (u64)1 << logn;
    u64 w[MAXL], wk[MAXL], e[1] = {k};
    memcpy(w, c->root, sizeof(w));
    for (int i = 0; i < c->two_adicity - logn; ++i) f_sqr(f, w, w);
    f_pow(f, wk, w, e, 1);
    u64 acc[MAXL] = {0};
    for (u64 i = n; i-- > 0;) {
        u64 t[MAXL];
        f_mul(f, acc, acc, wk);
        f_from_bytes(f, t, in + 32 * i);
        f_add(f, acc, acc, t);
    }
    f_to_bytes(f, out32, acc);
    return 0;
}

/* the same coefficient with the sum cut into `threads` chunks: X[k] = sum_t (w^k)^(lo_t) * Horner(chunk t) */
typedef struct { const curve_t* c; const uint8_t* in; u64 lo, hi; const u64* wk; u64 part[MAXL]; } eval_job;
static void* eval_worker(void* arg) {
    eval_job* J = (eval_job*)arg;
    const fctx* f = &J->c->fr;
    u64 acc[MAXL] = {0};
    for (u64 i = J->hi; i-- > J->lo;) {
        u64 t[MAXL];
        f_mul(f, acc, acc, J->wk);
        f_from_bytes(f, t, J->in + 32 * i);
        f_add(f, acc, acc, t);
    }
    /* times (w^k)^lo */
    u64 sh[MAXL], e[1] = {J->lo};
    f_pow(f, sh, J->wk, e, 1);
    f_mul(f, acc, acc, sh);
    memcpy(J->part, acc, sizeof(acc));
    return NULL;
}
int orc_ntt_eval_at_mt(int curve, const uint8_t* in, int logn, u64 k, int threads, uint8_t* out32) {
    const curve_t* c = get_curve(curve);
    if (!c || logn > c->two_adicity) return -1;
    const fctx* f = &c->fr;
    u64 n = (u64)1 << logn;
    if (threads < 1) threads = 1;
    if ((u64)threads > n) threads = (int)n;
    u64 w[MAXL], wk[MAXL], e[1] = {k};
    memcpy(w, c->root, sizeof(w));
    for (int i = 0; i < c->two_adicity - logn; ++i) f_sqr(f, w, w);
    f_pow(f, wk, w, e, 1);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    eval_job* jobs = (eval_job*)malloc(sizeof(eval_job) * threads);
    for (int t = 0; t < threads; ++t) {
        jobs[t] = (eval_job){c, in, n * t / threads, n * (t + 1) / threads, wk, {0}};
        pthread_create(&th[t], NULL, eval_worker, &jobs[t]);
    }
    u64 acc[MAXL] = {0};
    for (int t = 0; t < threads; ++t) {
        pthread_join(th[t], NULL);
        f_add(f, acc, acc, jobs[t].part);
    }
    f_to_bytes(f, out32, acc);
    free(th); free(jobs);
    return 0;
}
/* O(n^2) DFT for tiny n (independent structure from orc_ntt) */
int orc_dft_naive(int curve, const uint8_t* in, uint8_t* out, int logn) {
    const curve_t* c = get_curve(curve);
    if (!c || logn > 12) return -1;
    const fctx* f = &c->fr;
    u64 n = (u64)1 << logn;
    u64 w[MAXL];
    memcpy(w, c->root, sizeof(w));
    for (int i = 0; i < c->two_adicity - logn; ++i) f_sqr(f, w, w);
    u64* x = (u64*)malloc(32 * n);
    u64* pw = (u64*)malloc(32 * n);
    u64 cur[MAXL];
    memcpy(cur, f->one, sizeof(cur));
    for (u64 i = 0; i < n; ++i) {
        u64 t[MAXL];
        f_from_bytes(f, t, in + 32 * i);
        memcpy(x + 4 * i, t, 32);
        memcpy(pw + 4 * i, cur, 32);
        f_mul(f, cur, cur, w);
    }
    for (u64 k = 0; k < n; ++k) {
        u64 acc[MAXL] = {0};
        for (u64 j = 0; j < n; ++j) {
            u64 t[MAXL];
            f_mul(f, t, x + 4 * j, pw + 4 * ((j * k) % n));
            f_add(f, acc, acc, t);
        }
        f_to_bytes(f, out + 32 * k, acc);
    }
    free(x); free(pw);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * NTT 16-bank wire permutation, restated from the closed forms of the loops at
 * src/ingo_ntt/ntt_data.rs:80-111 (preprocess) and :113-156 (postprocess) on a scaled shape:
 * groups G (reference 512), everything else as the reference.  Element size 32 B.
 * banks: 16 contiguous arrays of n/16 elements each.
 * ------------------------------------------------------------------------------------------ */
int orc_ntt_preprocess(const uint8_t* in, uint8_t* banks, u64 n) {
    if (n % (512 * 2) != 0) return -1;
    u64 per_bank = n / 16;
    u64 off[16] = {0};
    u64 addr = 0;
    u64 nblocks = n / 512;
    for (u64 blk = 0; blk < nblocks; ++blk) {
        int core = (int)(blk % 2);
        for (int row = 0; row < 64; ++row) {
            for (int b = 0; b < 8; ++b) {
                int bank = core * 8 + b;
                memcpy(banks + (bank * per_bank + off[bank]) * 32, in + ((u64)(b + row * 8) + addr) * 32, 32);
                off[bank]++;
            }
        }
        addr += 512;
    }
    return 0;
}
int orc_ntt_postprocess(const uint8_t* banks, uint8_t* out, u64 n, u64 groups) {
    /* reference: groups = 512, blocks per group = 2*16*8 = 256, subNTT stride 1024 */
    u64 per_bank = n / 16;
    u64 blocks_per_group = n / 512 / 2 / groups;
    u64 off[16] = {0};
    for (u64 group = 0; group < groups; ++group) {
        u64 block = 0;
        for (u64 bb = 0; bb < blocks_per_group; ++bb) {
            for (int icore = 0; icore < 2; ++icore) {
                u64 isub = (icore ? groups : 0) + group + 2 * groups * block;
                u64 i = 0;
                for (int row = 0; row < 64; ++row) {
                    int base = ((group % 2 == 0) ? icore : 1 - icore) * 8;
                    for (int b = 0; b < 8; ++b) {
                        int bank = base + b;
                        u64 a = 512 * isub + i;
                        memcpy(out + a * 32, banks + (bank * per_bank + off[bank]) * 32, 32);
                        off[bank]++;
                        i++;
                    }
                }
            }
            block++;
        }
    }
    return 0;
}
