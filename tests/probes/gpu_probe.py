#!/usr/bin/env python3
"""Quick GPU-side probe: small MSMs against the oracle + timing of synthetic sizes.  Dev tool."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from oracle import pyref
import blaze_amd
from blaze_amd import DeviceBuffer
from blaze_amd.driver_client import DriverClient
from blaze_amd.ingo_msm import *

def run_msm(cl, pts, sc, n, hbm=None):
    params = MSMParams(n, hbm)
    cl.initialize(params); cl.start_process(); cl.set_data(MSMInput(pts, sc, params)); cl.wait_result()
    return cl.result().result

def main():
    dc = DriverClient(0)
    sizes = [int(x) for x in sys.argv[1:]] or [1, 2, 5, 300, 1000]
    for curve in ["BLS381", "BLS377", "BN254"]:
        for pf in (1, 8):
            cl = MSMClient(MSMInit(PointMemoryType.DMA, pf == 8, Curve[curve]), dc)
            for n in sizes:
                pts, sc, exp = oracle.input_generator(curve, n, pf, 42 + n)
                t = time.time()
                got = run_msm(cl, pts, sc, n)
                dt = time.time() - t
                ok = got == exp
                print(f"{curve} pf={pf} n={n}: {'OK' if ok else 'MISMATCH'} {dt*1e3:.1f} ms  {cl.get_api()}", flush=True)
                if not ok:
                    print(" got", got.hex()); print(" exp", exp.hex())
            cl.close()
    # synthetic sizes, timing
    curve = "BLS381"
    cl = MSMClient(MSMInit(PointMemoryType.DMA, False, Curve[curve]), dc)
    for logn in (16, 20, 22):
        n = 1 << logn
        dp = DeviceBuffer(0, n * 96); ds = DeviceBuffer(0, n * 32)
        t = time.time()
        blaze_amd.aux().blz_synth_points(0, 1, dp.ptr, n, 1, 0)
        blaze_amd.aux().blz_synth_scalars(0, 1, ds.ptr, n, 7)
        tg = time.time() - t
        for rep in range(2):
            t = time.time()
            got = run_msm(cl, dp, ds, n)
            dt = time.time() - t
        sc = ds.download()
        k = oracle.index_weighted_sum(curve, sc, n, 0)
        exp = oracle.result_from_affine(curve, oracle.generator_mul(curve, k))
        print(f"synth 2^{logn}: gen {tg:.2f}s msm {dt*1e3:.1f} ms {'OK' if got == exp else 'MISMATCH'} {cl.get_api()}", flush=True)
        dp.free(); ds.free()

if __name__ == "__main__":
    main()
