#!/usr/bin/env python3
"""Dev tool: long streams of identical tasks, two in flight, every result compared with the first one (which the oracle
checks by linearity): device-resident 2^24 and 2^26 (hidden sort), DMA-mode 2^22 (pieces), host-scalar HBM flow 2^24.
    python tests/probes/soak.py [scale]   (scale 1 = about a minute)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import blaze_amd
import oracle
from blaze_amd.ingo_msm import MSMInput, MSMParams, PointMemoryType
from gpu_util import msm_client, synth

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
bad = 0


def expect(curve, sc_bytes, n):
    k = oracle.index_weighted_sum(curve, sc_bytes, n, 0, threads=16)
    return oracle.result_from_affine(curve, oracle.generator_mul(curve, k))


def run(label, cl, prm, pts, sc, count, want):
    global bad
    t0 = time.perf_counter()
    pending = 0
    wrong = 0
    for _ in range(count):
        cl.initialize(prm); cl.start_process(); cl.set_data(MSMInput(pts, sc, prm))
        pending += 1
        if pending == 2:
            cl.wait_result(); wrong += cl.result().result != want; pending -= 1
    while pending:
        cl.wait_result(); wrong += cl.result().result != want; pending -= 1
    bad += wrong
    print(f"{label}: {count} tasks, {(time.perf_counter() - t0) / count * 1e3:.2f} ms each, wrong {wrong}", flush=True)


for curve, logn, count in (("BLS381", 24, int(300 * scale)), ("BLS381", 26, int(40 * scale)), ("BN254", 24, int(300 * scale))):
    n = 1 << logn
    dp, ds = synth(curve, n, seed=3)
    want = expect(curve, bytes(ds.download()), n)
    cl = msm_client(curve, 1)
    run(f"{curve} 2^{logn} resident", cl, MSMParams(n, None), dp, ds, count, want)
    cl.close(); dp.free(); ds.free()
n = 1 << 22
dp, ds = synth("BLS381", n, seed=4)
pts, sc = bytes(dp.download()), bytes(ds.download())
want = expect("BLS381", sc, n)
cl = msm_client("BLS381", 1)
run("BLS381 2^22 DMA", cl, MSMParams(n, None), pts, sc, int(300 * scale), want)
cl.close()
blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
cl = msm_client("BLS381", 1, PointMemoryType.HBM)
cl.load_data_to_hbm(dp, 0, 0)
for k in range(int(40 * scale)):   # lone tasks: the piecewise HBM flow
    run("BLS381 2^22 HBM flow, host scalars, lone" if k == 0 else "  ...", cl, MSMParams(n, (0, 0)), None, sc, 1, want) if k < 1 else None
    cl.initialize(MSMParams(n, (0, 0))); cl.start_process(); cl.set_data(MSMInput(None, sc, MSMParams(n, (0, 0)))); cl.wait_result()
    bad += cl.result().result != want
cl.close(); dp.free(); ds.free()
print("mismatches:", bad)
