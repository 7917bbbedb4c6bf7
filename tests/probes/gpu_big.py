#!/usr/bin/env python3
"""Dev tool: time synthetic MSMs at large sizes and check them through the linearity oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
import blaze_amd
from blaze_amd import DeviceBuffer
from blaze_amd.driver_client import DriverClient
from blaze_amd.ingo_msm import *

def main():
    curve = os.environ.get("CURVE", "BLS381")
    pf = int(os.environ.get("PF", "1"))
    check = int(os.environ.get("CHECK", "1"))
    reps = int(os.environ.get("REPS", "2"))
    dc = DriverClient(0)
    cid = int(Curve[curve])
    ps = 64 if curve == "BN254" else 96
    cl = MSMClient(MSMInit(PointMemoryType.DMA, pf == 8, Curve[curve]), dc)
    for logn in [int(x) for x in sys.argv[1:]]:
        n = 1 << logn
        dp = DeviceBuffer(0, n * ps * pf); ds = DeviceBuffer(0, n * 32)
        t = time.time()
        blaze_amd.aux().blz_synth_points(0, cid, dp.ptr, n, pf, 0)
        blaze_amd.aux().blz_synth_scalars(0, cid, ds.ptr, n, 7)
        tg = time.time() - t
        for rep in range(reps):
            params = MSMParams(n, None)
            t = time.time()
            cl.initialize(params); cl.start_process(); cl.set_data(MSMInput(dp, ds, params)); cl.wait_result()
            got = cl.result().result
            dt = time.time() - t
            print(f"  rep {rep}: wall {dt*1e3:.1f} ms {cl.get_api()}", flush=True)
        status = "unchecked"
        if check:
            sc = ds.download()
            k = oracle.index_weighted_sum(curve, sc, n, 0)
            exp = oracle.result_from_affine(curve, oracle.generator_mul(curve, k))
            status = 'OK' if got == exp else 'MISMATCH'
        print(f"{curve} pf={pf} 2^{logn}: gen {tg:.2f}s msm {dt*1e3:.1f} ms {status}", flush=True)
        dp.free(); ds.free()

if __name__ == "__main__":
    main()
