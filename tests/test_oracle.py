"""The CPU oracle (test infrastructure) against: published known answers, the committed golden
vectors minted by the independent pure-Python reference, and its own alternative algorithms."""
import json
import os
import random

import pytest

from oracle import pyref

HERE = os.path.dirname(os.path.abspath(__file__))
CURVES = ["BLS377", "BLS381", "BN254"]


def _golden(name):
    with open(os.path.join(HERE, "golden", name)) as f:
        return json.load(f)


def test_known_answers(orc):
    kat = _golden("kat.json")
    for curve, key in (("BLS381", "BLS381"), ("BN254", "BN254"), ("BLS377", "BLS377")):
        two_g = orc.generator_mul(curve, 2)
        fb = pyref.CURVES[curve]["fq_bytes"]
        g = orc.generator_mul(curve, 1)
        assert int.from_bytes(g[:fb], "little") == int(kat[f"{key}_G_x"], 16)
        assert int.from_bytes(g[fb:], "little") == int(kat[f"{key}_G_y"], 16)
        assert orc.generator_mul(curve, pyref.CURVES[curve]["r"]) is None          # r G = infinity
        assert pyref.enc_point(curve, pyref.mul(curve, pyref.generator(curve), 2)) == two_g   # both implementations
        assert int.from_bytes(two_g[:fb], "little") == int(kat[f"{key}_2G_x"], 16)
        assert int.from_bytes(two_g[fb:], "little") == int(kat[f"{key}_2G_y"], 16)
    assert orc.omega("BLS381", 27) == int(kat["BLS381_omega_2_27"], 16)
    w = orc.omega("BLS381", 27)
    r = pyref.CURVES["BLS381"]["r"]
    assert pow(w, 1 << 27, r) == 1 and pow(w, 1 << 26, r) == r - 1


@pytest.mark.parametrize("curve", CURVES)
def test_group_order_and_generator(orc, curve):
    c = pyref.CURVES[curve]
    assert orc.is_on_curve(curve, pyref.enc_point(curve, pyref.generator(curve)))
    assert orc.generator_mul(curve, c["r"]) is None                      # r*G = inf
    g = orc.generator_mul(curve, 1)
    m = orc.generator_mul(curve, c["r"] - 1)                             # (r-1)*G = -G
    assert m == pyref.enc_point(curve, pyref.neg(curve, pyref.dec_point(curve, g)))
    assert orc.point_add(curve, g, m) is None
    assert orc.point_add(curve, g, g) == orc.generator_mul(curve, 2)


def test_sizes(orc):
    # src/ingo_msm/msm_cfg.rs:44-92
    assert (orc.point_bytes("BLS381"), orc.result_bytes("BLS381")) == (96, 144)
    assert (orc.point_bytes("BLS377"), orc.result_bytes("BLS377")) == (96, 144)
    assert (orc.point_bytes("BN254"), orc.result_bytes("BN254")) == (64, 96)


def test_golden_msm_vectors(orc):
    for v in _golden("msm_vectors.json"):
        pts, sc = bytes.fromhex(v["points"]), bytes.fromhex(v["scalars"])
        exp = bytes.fromhex(v["result"])
        assert orc.msm_naive(v["curve"], pts, sc, v["n"], v["pf"]) == exp, v["name"]
        assert orc.msm_pippenger(v["curve"], pts, sc, v["n"], v["pf"], threads=2) == exp, v["name"]
        xy, on = orc.decode_result(v["curve"], exp)
        assert on


@pytest.mark.parametrize("curve", CURVES)
def test_precompute_equivalence(orc, curve):
    """pf=8 result == pf=1 result on the same (P, s): tests/msm/mod.rs:360-380 table semantics."""
    p1, s1, e1 = orc.input_generator(curve, 40, 1, 5)
    p8, s8, e8 = orc.input_generator(curve, 40, 8, 5)
    assert s1 == s8 and e1 == e8
    pb = orc.point_bytes(curve)
    assert bytes(p8[: pb]) == bytes(p1[: pb])
    assert orc.msm_pippenger(curve, p8, s8, 40, 8, threads=4) == e8


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("n", [255, 256, 257, 1024])
def test_harness_tiling(orc, curve, n):
    """256-tile repeat of the reference generator: expected = floor(n/256) S_256 + S_(n%256)."""
    pts, sc, exp = orc.input_generator(curve, n, 1, 9)
    assert orc.msm_pippenger(curve, pts, sc, n, 1, threads=8) == exp
    if n >= 512:
        pb = orc.point_bytes(curve)
        assert bytes(pts[: 256 * pb]) == bytes(pts[256 * pb: 512 * pb])


def test_linearity_helper(orc):
    curve = "BLS381"
    r = pyref.CURVES[curve]["r"]
    rng = random.Random(3)
    ss = [rng.randrange(r) for _ in range(50)]
    sc = b"".join(s.to_bytes(32, "little") for s in ss)
    assert orc.index_weighted_sum(curve, sc, 50, 10) == sum(s * (10 + i + 1) for i, s in enumerate(ss)) % r


def test_golden_ntt_vectors(orc):
    for v in _golden("ntt_vectors.json"):
        data = bytes.fromhex(v["input"])
        out = bytes.fromhex(v["output"])
        assert bytes(orc.ntt("BLS381", data, v["logn"])) == out
        if v["logn"] <= 6:
            assert bytes(orc.dft_naive("BLS381", data, v["logn"])) == out
        assert bytes(orc.ntt("BLS381", out, v["logn"], inverse=True)) == data
        assert orc.ntt_eval_at("BLS381", data, v["logn"], 1) == int.from_bytes(out[32:64], "little") or v["logn"] == 0


def test_ntt_threads_and_delta(orc):
    logn = 12
    n = 1 << logn
    delta = (1).to_bytes(32, "little") + b"\0" * (32 * (n - 1))
    assert bytes(orc.ntt("BLS381", delta, logn, threads=4)) == (1).to_bytes(32, "little") * n


def test_ntt_bank_permutation_roundtrip(orc):
    """NTTBanks::preprocess / postprocess (ntt_data.rs:80-156) at a scaled shape: postprocess after
    preprocess leaves a transpose of 512-element blocks, checked by closed form."""
    groups = 4
    n = 512 * 2 * groups * 2  # blocks_per_group = 2
    data = bytearray(32 * n)
    for i in range(n):
        data[32 * i: 32 * i + 4] = i.to_bytes(4, "little")
    banks = orc.ntt_preprocess(data, n)
    assert len(banks) == 32 * n
    out = orc.ntt_postprocess(banks, n, groups)
    got = [int.from_bytes(out[32 * i: 32 * i + 4], "little") for i in range(n)]
    assert sorted(got) == list(range(n))           # a permutation
    for a in range(0, n, 512):                     # whole 512-blocks move together, order kept
        assert got[a + 1] == got[a] + 1 and got[a + 511] == got[a] + 511 and got[a] % 512 == 0


def test_oracle_under_sanitizers():
    """ASan + UBSan build of the C restatement, every entry point on small inputs (CPU only: GPU
    sanitizers are not available on the pool)."""
    import subprocess

    odir = os.path.join(os.path.dirname(HERE), "oracle")
    subprocess.check_call(["make", "-C", odir, "selftest_asan"], stdout=subprocess.DEVNULL)
    p = subprocess.run([os.path.join(odir, "selftest_asan")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "oracle selftest ok" in p.stdout


def test_ntt_conventions_against_naive_sums(orc):
    """orc_ntt_ex (a caller's root, bit-reversed buffers, inverse) against the definition evaluated with Python integers."""
    import random

    rng = random.Random(1)
    for curve in ("BLS381", "BLS377", "BN254"):
        r = pyref.CURVES[curve]["r"]
        for logn in (1, 3, 6):
            n = 1 << logn
            xs = [rng.randrange(r) for _ in range(n)]
            data = b"".join(x.to_bytes(32, "little") for x in xs)
            w = orc.omega(curve, logn)
            br = lambda i: int(format(i, f"0{logn}b")[::-1], 2)   # noqa: E731
            for t in (1, 3, 5):
                wt = pow(w, t, r)
                exp = [sum(xs[i] * pow(wt, i * k, r) for i in range(n)) % r for k in range(n)]
                got = orc.ntt(curve, data, logn, root=wt)
                assert bytes(got) == b"".join(e.to_bytes(32, "little") for e in exp), (curve, logn, t)
                din = b"".join(xs[br(p)].to_bytes(32, "little") for p in range(n))
                got2 = orc.ntt(curve, din, logn, root=wt, bitrev_in=True, bitrev_out=True)
                assert bytes(got2) == b"".join(exp[br(p)].to_bytes(32, "little") for p in range(n))
                assert bytes(orc.bitrev_permute(data, logn, 2)) == din
                assert bytes(orc.ntt(curve, bytes(got), logn, inverse=True, root=wt)) == data
