#!/usr/bin/env python3
"""Mint the committed golden vectors with the pure-Python big-int reference (oracle/pyref.py), which
is independent of both the C oracle and the HIP code.  The reference repository holds no vectors
for this path, so these are minted here from the published curve parameters.

    python tests/golden/make_golden.py      # rewrites tests/golden/*.json deterministically
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyref  # noqa: E402


def msm_vectors():
    out = []
    for curve in ("BLS377", "BLS381", "BN254"):
        r = pyref.CURVES[curve]["r"]
        G = pyref.generator(curve)
        for pf in (1, 8):
            for n in (1, 2, 5):
                pts, sc, results = pyref.input_generator(curve, n, pf, seed=1000 * n + pf)
                exp = pyref.expected_from_results(curve, results, n)
                out.append(dict(name=f"random_n{n}", curve=curve, pf=pf, n=n, points=pts.hex(), scalars=sc.hex(),
                                result=pyref.enc_result(curve, exp).hex()))
        # edge scalars 0, 1, r-1 on the generator and on 2G; duplicate and negated points
        rng = random.Random(7)
        P = pyref.mul(curve, G, rng.randrange(1, r))
        cases = {
            "scalar_zero": ([G], [0]),
            "scalar_one": ([G], [1]),
            "scalar_r_minus_1": ([G], [r - 1]),
            "all_zero_scalars": ([G, P], [0, 0]),
            "duplicate_points_same_scalar": ([P, P, P, P], [5, 5, 5, 5]),
            "negated_pair_cancels": ([P, pyref.neg(curve, P)], [12345, 12345]),
            "negated_pair_plus_one": ([P, pyref.neg(curve, P), G], [r - 3, r - 3, 2]),
            "same_point_many_windows": ([P] * 6, [(1 << 200) + 3, (1 << 200) + 3, 3, 3, 1 << 100, 1 << 100]),
        }
        for name, (pl, sl) in cases.items():
            pts = b"".join(pyref.enc_point(curve, p) for p in pl)
            sc = b"".join(pyref.enc_scalar(s) for s in sl)
            e = pyref.msm_naive(curve, pts, sc, len(pl), 1)
            out.append(dict(name=name, curve=curve, pf=1, n=len(pl), points=pts.hex(), scalars=sc.hex(),
                            result=pyref.enc_result(curve, e).hex()))
    return out


def kat_vectors():
    """Known answers that do not depend on any implementation here: the doubled generators published
    in EIP-2537 (BLS12-381 G1 add vector) and EIP-196 (alt_bn128), and r*G = infinity."""
    return dict(
        BLS381_2G_x="0572cbea904d67468808c8eb50a9450c9721db309128012543902d0ac358a62ae28f75bb8f1c7c42c39a8c5529bf0f4e",
        BLS381_2G_y="166a9d8cabc673a322fda673779d8e3822ba3ecb8670e461f73bb9021d5fd76a4c56d9d4cd16bd1bba86881979749d28",
        BN254_2G_x="030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3",
        BN254_2G_y="15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4",
        BLS381_omega_2_27="23397a9300f8f98bece8ea224f31d25db94f1101b1d7a628e2d0a7869f0319ed",
    )


def ntt_vectors():
    out = []
    r = pyref.CURVES["BLS381"]["r"]
    rng = random.Random(11)
    for logn in (1, 3, 6, 10):
        n = 1 << logn
        xs = [rng.randrange(r) for _ in range(n)]
        if logn == 3:
            xs = [1] + [0] * (n - 1)  # delta -> all ones
        ys = pyref.ntt("BLS381", xs) if logn > 6 else pyref.dft_naive("BLS381", xs)
        out.append(dict(logn=logn, input=b"".join(x.to_bytes(32, "little") for x in xs).hex(),
                        output=b"".join(y.to_bytes(32, "little") for y in ys).hex()))
    return out


def main():
    with open(os.path.join(HERE, "msm_vectors.json"), "w") as f:
        json.dump(msm_vectors(), f, indent=0)
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat_vectors(), f, indent=1)
    with open(os.path.join(HERE, "ntt_vectors.json"), "w") as f:
        json.dump(ntt_vectors(), f, indent=0)
    print("golden vectors written")


if __name__ == "__main__":
    main()
