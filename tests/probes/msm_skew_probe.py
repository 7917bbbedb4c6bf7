#!/usr/bin/env python3
"""Dev tool: scalar DISTRIBUTIONS a bucket method likes least - all equal (one bucket per window holds every entry), all r - 1, all
2^256 - 1 (non-canonical), one non-zero among zeros, low words only (upper windows empty), byte patterns that sit on the edges of the
signed-digit recoding (0x80.., 0x7f.., 0xff.., 0x55.., 0xaa..), half equal / half random - at 2^logn points on the three curves, pf = 1
(arena bases, two in flight) and pf = 8 (exact path and checked-table plan), every result checked through linearity.
    python3 tests/probes/msm_skew_probe.py [logn] [logn_pf8]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np  # noqa: E402

import blaze_amd  # noqa: E402
import oracle  # noqa: E402
from blaze_amd import DeviceBuffer  # noqa: E402
from blaze_amd.ingo_msm import MSMInput, MSMParams, PointMemoryType  # noqa: E402
from gpu_util import msm_client, synth  # noqa: E402
from oracle import pyref  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
logn8 = int(sys.argv[2]) if len(sys.argv) > 2 else logn - 3
L = blaze_amd.lib()
bad, t0 = 0, time.time()


def patterns(n, r, rng):
    def rep(v):
        return np.tile(np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.uint8), n).reshape(n, 32)

    rnd = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    rnd[:, 31] &= 0x0F
    k = int.from_bytes(rng.integers(0, 256, size=31, dtype=np.uint8).tobytes(), "little")
    out = [("all equal", rep(k)), ("all r - 1", rep(r - 1)), ("all 2^256 - 1", rep((1 << 256) - 1)), ("all zero", rep(0))]
    one = rep(0).copy()
    one[n // 3] = rnd[0]
    out.append(("one non-zero", one))
    low = rnd.copy()
    low[:, 4:] = 0
    out.append(("low 32 bits only", low))
    top = rep(0).copy()
    top[:, 28:] = rnd[:, 28:]
    out.append(("top 32 bits only", top))
    for b in (0x80, 0x7F, 0xFF, 0x55, 0xAA, 0x01):
        out.append((f"bytes 0x{b:02x}", np.full((n, 32), b, dtype=np.uint8)))
    half = rnd.copy()
    half[::2] = rep(k)[::2]
    out.append(("half equal, half random", half))
    few = rnd.copy()
    few[:] = rnd[rng.integers(0, 7, size=n)]
    out.append(("seven distinct values", few))
    return out


for curve in ("BLS381", "BLS377", "BN254"):
    r = pyref.CURVES[curve]["r"]
    for pf, lg in ((1, logn), (8, logn8)):
        n = 1 << lg
        blaze_amd._lib.check(L.blz_arena_release(0))
        dp, _ds = synth(curve, n, pf=pf, seed=3)
        _ds.free()
        clients = [("arena", msm_client(curve, pf, PointMemoryType.HBM))]
        clients[0][1].load_data_to_hbm(dp, 0, 0)
        if pf == 8:
            plan = msm_client(curve, 8, PointMemoryType.HBM)
            plan.set_precompute_plan(True)
            clients.append(("plan", plan))
        dp.free()
        prm = MSMParams(n, (0, 0))
        rng = np.random.default_rng(lg * 10 + pf)
        for name, sc in patterns(n, r, rng):
            scb = sc.tobytes()
            k = oracle.index_weighted_sum(curve, scb, n, 0, threads=8)
            exp = oracle.result_from_affine(curve, oracle.generator_mul(curve, k))
            d = DeviceBuffer(0, len(scb))
            d.upload(scb)
            for cname, cl in clients:
                t1 = time.perf_counter()
                outs = []
                cl.initialize(prm); cl.start_process(); cl.set_data(MSMInput(None, d, prm))
                cl.initialize(prm); cl.start_process(); cl.set_data(MSMInput(None, scb, prm))   # (the second one from host memory)
                for _ in range(2):
                    cl.wait_result(); outs.append(cl.result().result)
                ms = (time.perf_counter() - t1) * 1e3
                ok = all(o == exp for o in outs)
                bad += not ok
                print(f"{curve} pf={pf} 2^{lg} {cname:5s} {name:26s}: {'ok' if ok else 'MISMATCH'} ({ms:.1f} ms for two tasks)", flush=True)
            d.free()
        for _, cl in clients:
            cl.close()
blaze_amd._lib.check(L.blz_arena_release(0))
print(f"mismatches: {bad} ({time.time() - t0:.0f} s)")
sys.exit(1 if bad else 0)
