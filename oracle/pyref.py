"""TEST INFRASTRUCTURE ONLY - independent pure-Python big-int restatement of the
reference's CPU check path (tests/msm/mod.rs) and of the wire formats.

PARITY UNPINNED at byte level against the reference: the reference holds no golden
vectors for MSM or NTT (inputs are thread_rng, tests/msm/mod.rs:66,186,311; NTT
goldens are external files, tests/integration_ntt.rs:15-18) and its arithmetic
lives in arkworks 0.3.0 (Cargo.toml:14-19), which is not vendored and cannot be
built here (no rustc).  What pins this file instead: published curve constants
and known answers (EIP-2537 / EIP-196 doubling of the generator, r*G = inf),
and byte-for-byte agreement with the independent C restatement in
oracle/blz_oracle.c on seeded vectors (tests/test_oracle.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It is never on the product path.

Follows:
  tests/msm/mod.rs:297-358  input_generator_*  (generation + 256-tile repeat)
  tests/msm/mod.rs:360-380  precompute_base_*  (P, 2^32 P, ..., 2^224 P)
  tests/msm/mod.rs:382-420  result_check_*     (Z|Y|X decode, x=X/Z, y=Y/Z)
  src/ingo_msm/msm_cfg.rs:44-92                (sizes)
"""
from __future__ import annotations

import random

# --------------------------------------------------------------------------------------
# Curve table.  Values are the published parameters of the three curves (arkworks 0.3.0
# ark-bls12-381 / ark-bls12-377 / ark-bn254 use the same ones); SURVEY.md appendix B.
# --------------------------------------------------------------------------------------
CURVES = {
    "BLS377": dict(
        id=0,
        q=0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001,
        r=0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001,
        b=1,
        gx=0x008848DEFE740A67C8FC6225BF87FF5485951E2CAA9D41BB188282C8BD37CB5CD5481512FFCD394EEAB9B16EB21BE9EF,
        gy=0x01914A69C5102EFF1F674F5D30AFEEC4BD7FB348CA3E52D96D182AD44FB82305C2FE3D3634A9591AFD82DE55559C8EA6,
        fq_bytes=48,
        two_adicity=47,
        fr_gen=22,
    ),
    "BLS381": dict(
        id=1,
        q=0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
        r=0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
        b=4,
        gx=0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
        gy=0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
        fq_bytes=48,
        two_adicity=32,
        fr_gen=7,
    ),
    "BN254": dict(
        id=2,
        q=0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47,
        r=0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001,
        b=3,
        gx=1,
        gy=2,
        fq_bytes=32,
        two_adicity=28,
        fr_gen=5,
    ),
}
CURVE_BY_ID = {v["id"]: k for k, v in CURVES.items()}
SCALAR_BYTES = 32
LARGE_PARAM = 256  # tests/msm/mod.rs:19
PRECOMPUTE_FACTOR = 8  # src/ingo_msm/msm_api.rs:40


def point_bytes(curve: str) -> int:  # src/ingo_msm/msm_cfg.rs:47,57,67,77,87
    return 2 * CURVES[curve]["fq_bytes"]


def result_bytes(curve: str) -> int:  # src/ingo_msm/msm_cfg.rs:46,56,66,76,86
    return 3 * CURVES[curve]["fq_bytes"]


# --------------------------------------------------------------------------------------
# Affine group law over y^2 = x^3 + b, None = point at infinity.
# --------------------------------------------------------------------------------------
def is_on_curve(curve: str, P) -> bool:
    if P is None:
        return True
    c = CURVES[curve]
    x, y = P
    return (y * y - x * x * x - c["b"]) % c["q"] == 0


def neg(curve: str, P):
    if P is None:
        return None
    return (P[0], (-P[1]) % CURVES[curve]["q"])


def add(curve: str, P, Q):
    q = CURVES[curve]["q"]
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % q == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, q) % q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, q) % q
    x3 = (lam * lam - x1 - x2) % q
    y3 = (lam * (x1 - x3) - y1) % q
    return (x3, y3)


def mul(curve: str, P, k: int):
    """Double-and-add, the structure of ark-ec's `AffineCurve::mul`
    (tests/msm/mod.rs:88,208,333 call it once per element)."""
    R = None
    Q = P
    while k:
        if k & 1:
            R = add(curve, R, Q)
        Q = add(curve, Q, Q)
        k >>= 1
    return R


def generator(curve: str):
    c = CURVES[curve]
    return (c["gx"], c["gy"])


# --------------------------------------------------------------------------------------
# Wire formats.
# --------------------------------------------------------------------------------------
def enc_fq(curve: str, v: int) -> bytes:
    return int(v).to_bytes(CURVES[curve]["fq_bytes"], "little")


def enc_scalar(s: int) -> bytes:  # tests/msm/mod.rs:331-332
    return int(s).to_bytes(SCALAR_BYTES, "little")


def enc_point(curve: str, P) -> bytes:  # tests/msm/mod.rs:363-366  (x || y, no infinity)
    assert P is not None
    return enc_fq(curve, P[0]) + enc_fq(curve, P[1])


def dec_point(curve: str, buf: bytes):
    fb = CURVES[curve]["fq_bytes"]
    return (int.from_bytes(buf[:fb], "little"), int.from_bytes(buf[fb : 2 * fb], "little"))


def precompute_base(curve: str, P, pf: int) -> bytes:
    """tests/msm/mod.rs:360-380: P, 2^32 P, ..., 2^(32(pf-1)) P, contiguous."""
    out = enc_point(curve, P)
    r = CURVES[curve]["r"]
    for i in range(1, pf):
        out += enc_point(curve, mul(curve, P, pow(2, 32 * i, r)))
    return out


def dec_result(curve: str, buf: bytes):
    """tests/msm/mod.rs:397-405: Z=[0..fb] Y=[fb..2fb] X=[2fb..3fb], x=X/Z, y=Y/Z.
    Z == 0 -> infinity (the reference would panic on `inverse().unwrap()`)."""
    c = CURVES[curve]
    fb, q = c["fq_bytes"], c["q"]
    Z = int.from_bytes(buf[0:fb], "little") % q
    Y = int.from_bytes(buf[fb : 2 * fb], "little") % q
    X = int.from_bytes(buf[2 * fb : 3 * fb], "little") % q
    if Z == 0:
        return None
    zi = pow(Z, -1, q)
    return (X * zi % q, Y * zi % q)


def enc_result(curve: str, P) -> bytes:
    """Canonical device output of this build: Z=1 | y | x; infinity = Z=0 | Y=1 | X=0."""
    if P is None:
        return enc_fq(curve, 0) + enc_fq(curve, 1) + enc_fq(curve, 0)
    return enc_fq(curve, 1) + enc_fq(curve, P[1]) + enc_fq(curve, P[0])


# --------------------------------------------------------------------------------------
# MSM semantics of the device task.
# --------------------------------------------------------------------------------------
def msm_naive(curve: str, points: bytes, scalars: bytes, n: int, pf: int = 1):
    """R = sum_i s_i P_i (pf=1) or sum_i sum_j s_{i,j} B_{i,j} with s_{i,j} the j-th
    32-bit LE chunk of s_i and B_{i,j} the j-th of the pf bases sent for element i."""
    pb = point_bytes(curve)
    R = None
    for i in range(n):
        s = int.from_bytes(scalars[32 * i : 32 * i + 32], "little")
        if pf == 1:
            P = dec_point(curve, points[pb * i : pb * (i + 1)])
            R = add(curve, R, mul(curve, P, s))
        else:
            for j in range(pf):
                sj = (s >> (32 * j)) & 0xFFFFFFFF
                B = dec_point(curve, points[pb * (i * pf + j) : pb * (i * pf + j + 1)])
                R = add(curve, R, mul(curve, B, sj))
    return R


def rand_point(curve: str, rng: random.Random):
    """A random point of the prime-order subgroup: k*G, k uniform (the reference uses
    G1Projective::rand, tests/msm/mod.rs:327; both yield r-torsion points)."""
    r = CURVES[curve]["r"]
    return mul(curve, generator(curve), rng.randrange(1, r))


def input_generator(curve: str, nof_elements: int, pf: int, seed: int):
    """tests/msm/mod.rs:297-358 restated.  Returns (bases, scalars, results) where
    results[i] = running sum after element i of the <=256-element tile.
    Harness quirk kept out: the reference truncates with rest*96*8 regardless of pf and
    point size (mod.rs:101,221,346); here the tail is the first `rest` elements exactly."""
    rng = random.Random(seed)
    r = CURVES[curve]["r"]
    nof = min(nof_elements, LARGE_PARAM) if nof_elements > LARGE_PARAM else nof_elements
    bases, scalars, results = b"", b"", []
    acc = None
    per = point_bytes(curve) * pf
    for _ in range(nof):
        P = rand_point(curve, rng)
        bases += precompute_base(curve, P, pf)
        s = rng.randrange(0, r)
        scalars += enc_scalar(s)
        acc = add(curve, acc, mul(curve, P, s))
        results.append(acc)
    if nof_elements > LARGE_PARAM:
        mult, rest = divmod(nof_elements, LARGE_PARAM)
        bases = bases * mult + bases[: rest * per]
        scalars = scalars * mult + scalars[: rest * 32]
    return bases, scalars, results


def expected_from_results(curve: str, results, nof_elements: int):
    """tests/msm/mod.rs:388-395: expected = floor(n/256)*S_256 + S_(n%256)
    (with the n==256 off-by-one of the reference checker NOT reproduced)."""
    if nof_elements <= LARGE_PARAM:
        return results[nof_elements - 1] if nof_elements else None
    mult, rest = divmod(nof_elements, LARGE_PARAM)
    e = mul(curve, results[LARGE_PARAM - 1], mult)
    if rest:
        e = add(curve, e, results[rest - 1])
    return e


# --------------------------------------------------------------------------------------
# NTT over Fr (direction/order are this build's definition, SURVEY.md a13: forward
# transform X[k] = sum_j x[j] w^(jk), natural order in and out, w = omega(2^logn)).
# --------------------------------------------------------------------------------------
def omega(curve: str, logn: int) -> int:
    c = CURVES[curve]
    r = c["r"]
    root = pow(c["fr_gen"], (r - 1) >> c["two_adicity"], r)
    return pow(root, 1 << (c["two_adicity"] - logn), r)


def dft_naive(curve: str, xs):
    r = CURVES[curve]["r"]
    n = len(xs)
    logn = n.bit_length() - 1
    w = omega(curve, logn)
    return [sum(x * pow(w, j * k, r) for j, x in enumerate(xs)) % r for k in range(n)]


def ntt(curve: str, xs, inverse: bool = False):
    r = CURVES[curve]["r"]
    n = len(xs)
    logn = n.bit_length() - 1
    a = list(xs)
    j = 0
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j |= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    w_n = omega(curve, logn)
    if inverse:
        w_n = pow(w_n, -1, r)
    length = 2
    while length <= n:
        wl = pow(w_n, n // length, r)
        for s in range(0, n, length):
            w = 1
            for k in range(length // 2):
                u, v = a[s + k], a[s + k + length // 2] * w % r
                a[s + k], a[s + k + length // 2] = (u + v) % r, (u - v) % r
                w = w * wl % r
        length <<= 1
    if inverse:
        ninv = pow(n, -1, r)
        a = [x * ninv % r for x in a]
    return a
