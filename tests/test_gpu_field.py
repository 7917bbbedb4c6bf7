"""Element-wise parity of the device field / group primitives against the CPU oracle
(bit-exact: integer arithmetic)."""
import ctypes as C
import random

import pytest

import blaze_amd
from oracle import pyref

pytestmark = pytest.mark.gpu

CURVES = ["BLS377", "BLS381", "BN254"]


def _edge_values(m):
    return [0, 1, 2, m - 1, m - 2, (m - 1) // 2, (m + 1) // 2, (1 << (m.bit_length() - 1)), m // 3]


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("field", [0, 1])
def test_field_ops(gpu, curve, field):
    c = pyref.CURVES[curve]
    m = c["q"] if field == 0 else c["r"]
    nb = c["fq_bytes"] if field == 0 else 32
    rng = random.Random(1234 + field)
    edge = _edge_values(m)
    a = [x for x in edge for _ in edge] + [rng.randrange(m) for _ in range(4096)]
    b = [y for _ in edge for y in edge] + [rng.randrange(m) for _ in range(4096)]
    n = len(a)
    ab = b"".join(x.to_bytes(nb, "little") for x in a)
    bb = b"".join(x.to_bytes(nb, "little") for x in b)
    ops = {0: lambda x, y: x * y % m, 1: lambda x, y: (x + y) % m, 2: lambda x, y: (x - y) % m,
           3: lambda x, y: pow(x, -1, m) if x else 0, 4: lambda x, y: x * x % m}
    if field == 0 or curve != "BLS381":   # lazy-range fields have the fused sum of two products (ec.hip.hpp's Y3)
        ops[5] = lambda x, y: (x * y + (x + y) * (x - y)) % m
        ops[6] = lambda x, y: (x * y - (x + y) * (x - y)) % m
    if True:   # reduced-radix twins: the BLS base fields (14 x 28 bits); BN254's base field and every scalar field (9 x 29)
        ops[10] = lambda x, y: x * y % m
        ops[11] = lambda x, y: x * x % m
        ops[12] = lambda x, y: (x * y + (x + y) * (x - y)) % m
        ops[13] = lambda x, y: (x - 3 * y) * y % m
        ops[14] = lambda x, y: 1 if x == y else 0
        ops[15] = lambda x, y: (x - 3 * y) * x % m
    if field == 0 and curve != "BN254":   # the row-cooperative arithmetic of the tail (ec_row.hip.hpp): the BLS base fields
        ops[20] = lambda x, y: x * y % m
        ops[21] = lambda x, y: (x - 3 * y) * y % m
        ops[22] = lambda x, y: (x - y) * (x + y) % m
        ops[23] = lambda x, y: 1 if x == y else 0
        ops[24] = lambda x, y: 1            # carry machinery self-check on synthetic limb patterns
    for op, fn in ops.items():
        cnt = n if op != 3 else 200  # inversion is slow on one lane; sample
        out = C.create_string_buffer(cnt * nb)
        rc = blaze_amd.aux().blz_test_field_op(0, pyref.CURVES[curve]["id"], field, op, ab, bb, C.cast(out, C.c_void_p), cnt)
        assert rc == 0, gpu.blz_last_error_message()
        got = [int.from_bytes(out.raw[i * nb:(i + 1) * nb], "little") for i in range(cnt)]
        exp = [fn(a[i], b[i]) for i in range(cnt)]
        bad = [i for i in range(cnt) if got[i] != exp[i]]
        assert not bad, f"{curve} field={field} op={op}: {len(bad)} mismatches, first at {bad[0]}: a={a[bad[0]]:#x} b={b[bad[0]]:#x}"


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("field", [0, 1])
def test_field_inversion_edge_representations(gpu, curve, field):
    """fp_inv works on the plain integer A = x R mod m behind a Montgomery-form input (Kaliski's almost inverse: A^-1 2^k with
    n <= k <= 2n, the power of two taken off in the closing Montgomery products - by one product when 2W - k <= n, by three when
    A is so close to a power of two that k stays within a few bits of n).  The random inputs of test_field_ops never get near
    those representations: here x is chosen so that A is 1, 2, 3, every power of two, 2^j +- 1, m - 1, m - 2^j, (m +- 1) / 2."""
    c = pyref.CURVES[curve]
    m = c["q"] if field == 0 else c["r"]
    nb = c["fq_bytes"] if field == 0 else 32
    R = 1 << (8 * nb)
    Rinv = pow(R, -1, m)
    bits = m.bit_length()
    A = [1, 2, 3, m - 1, m - 2, (m - 1) // 2, (m + 1) // 2]
    for j in range(1, bits):
        for v in (1 << j, (1 << j) - 1, (1 << j) + 1, m - (1 << j)):
            if 0 < v < m:
                A.append(v)
    xs = [a * Rinv % m for a in A]
    ab = b"".join(x.to_bytes(nb, "little") for x in xs)
    out = C.create_string_buffer(len(xs) * nb)
    rc = blaze_amd.aux().blz_test_field_op(0, c["id"], field, 3, ab, ab, C.cast(out, C.c_void_p), len(xs))
    assert rc == 0, gpu.blz_last_error_message()
    got = [int.from_bytes(out.raw[i * nb:(i + 1) * nb], "little") for i in range(len(xs))]
    bad = [i for i, x in enumerate(xs) if got[i] != pow(x, -1, m)]
    assert not bad, f"{curve} field={field}: {len(bad)} wrong inverses, first for A={A[bad[0]]:#x}"


@pytest.mark.parametrize("curve", CURVES)
def test_ec_ops(gpu, curve):
    c = pyref.CURVES[curve]
    cid = c["id"]
    rng = random.Random(99)
    G = pyref.generator(curve)
    base = [pyref.mul(curve, G, rng.randrange(1, c["r"])) for _ in range(24)]
    P, Q, fl = [], [], []
    for i in range(24):
        for j in range(24):
            P.append(base[i]); Q.append(base[j]); fl.append(0)   # includes P == Q (doubling)
    for i in range(24):
        P.append(base[i]); Q.append(pyref.neg(curve, base[i])); fl.append(0)  # P + (-P)
        P.append(base[i]); Q.append(base[(i + 1) % 24]); fl.append(1)        # inf + Q
        P.append(base[i]); Q.append(base[(i + 1) % 24]); fl.append(2)        # P + inf
        P.append(base[i]); Q.append(base[i]); fl.append(3)                   # inf + inf
    n = len(P)
    pb = b"".join(pyref.enc_point(curve, p) for p in P)
    qb = b"".join(pyref.enc_point(curve, q) for q in Q)
    flb = bytes(fl)
    sz = 2 * c["fq_bytes"]

    def expect(op, p, q, f):
        p = None if f & 1 else p
        q = None if f & 2 else q
        if op == 0 or op == 2:
            return pyref.add(curve, p, q)
        if op == 1:
            return pyref.add(curve, p, p)
        if op in (4, 6, 8):
            return pyref.add(curve, p, q)
        if op == 9:
            return pyref.add(curve, p, p)
        if op == 7:
            return pyref.add(curve, q, None if p is None else pyref.neg(curve, p))
        return pyref.add(curve, p, pyref.neg(curve, q))

    for op in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9):
        out = C.create_string_buffer(n * sz)
        oinf = C.create_string_buffer(n)
        rc = blaze_amd.aux().blz_test_ec_op(0, cid, op, pb, qb, flb, C.cast(out, C.c_void_p), C.cast(oinf, C.c_void_p), n)
        assert rc == 0, gpu.blz_last_error_message()
        for i in range(n):
            e = expect(op, P[i], Q[i], fl[i])
            if e is None:
                assert oinf.raw[i] == 1, f"{curve} op={op} case {i}: expected infinity"
            else:
                assert oinf.raw[i] == 0, f"{curve} op={op} case {i}: unexpected infinity"
                assert out.raw[i * sz:(i + 1) * sz] == pyref.enc_point(curve, e), f"{curve} op={op} case {i} fl={fl[i]}"
