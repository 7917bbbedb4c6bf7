#!/usr/bin/env python3
"""Dev tool: the reference harness's repeated 256-tile at large n (tests/integration_msm.rs:385-467
uses 2^26): every bucket that is hit at all holds n/256 equal points."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from blaze_amd.driver_client import DriverClient
from blaze_amd.ingo_msm import *

curve = os.environ.get("CURVE", "BLS381"); pf = int(os.environ.get("PF", "1"))
cl = MSMClient(MSMInit(PointMemoryType.DMA, pf == 8, Curve[curve]), DriverClient(0))
for logn in [int(x) for x in sys.argv[1:]]:
    n = 1 << logn
    t = time.time(); pts, sc, exp = oracle.input_generator(curve, n, pf, 5); tg = time.time() - t
    params = MSMParams(n, None)
    for rep in range(2):
        t0 = time.time(); cl.initialize(params); cl.start_process()
        t1 = time.time(); cl.set_data(MSMInput(pts, sc, params))
        t2 = time.time(); cl.wait_result(); got = cl.result().result
        t3 = time.time()
        print(f"  set_data {1e3*(t2-t1):.0f} ms wait_result {1e3*(t3-t2):.0f} ms full {1e3*(t3-t0):.0f} ms {cl.get_api()}", flush=True)
    print(f"{curve} pf={pf} harness 2^{logn}: gen {tg:.1f}s {'OK' if got == exp else 'MISMATCH'}", flush=True)
