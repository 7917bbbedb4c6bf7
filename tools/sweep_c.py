#!/usr/bin/env python3
"""Dev tool: sweep the window size c (BLAZE_MSM_PLAN=c=..: uniform windows) per problem size against the planner's own choice
(mixed widths); prints the throughput-relevant part of the device pipeline (sort + accumulation + bucket reduce; the
finish step hides under the next task) and the full latency."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blaze_amd
from blaze_amd import DeviceBuffer
from blaze_amd.driver_client import DriverClient
from blaze_amd.ingo_msm import *
dc = DriverClient(0)
for logn in [int(x) for x in sys.argv[1:]]:
    n = 1 << logn
    dp = DeviceBuffer(0, n * 96); ds = DeviceBuffer(0, n * 32)
    blaze_amd.aux().blz_synth_points(0, 1, dp.ptr, n, 1, 0); blaze_amd.aux().blz_synth_scalars(0, 1, ds.ptr, n, 7)
    row = []
    for c in [0] + list(range(max(8, logn - 7), min(23, logn - 2) + 1)):
        if c:
            os.environ["BLAZE_MSM_PLAN"] = f"c={c}"
        else:
            os.environ.pop("BLAZE_MSM_PLAN", None)
        cl = MSMClient(MSMInit(PointMemoryType.DMA, False, Curve.BLS381), dc)
        best = (1e9, 1e9)
        for rep in range(3):
            p = MSMParams(n, None); cl.initialize(p); cl.start_process(); cl.set_data(MSMInput(dp, ds, p)); cl.wait_result(); cl.result()
            a = cl.get_api()
            best = min(best, (a["sort_ms"] + a["phase1_accumulate_ms"] + a["phase2_reduce_ms"], a["total_ms"]))
        row.append(("plan" if c == 0 else c, round(best[0], 2), round(best[1], 2)))
        cl.close()
    print(f"2^{logn}:", row, "best", min(row, key=lambda t: t[1]), flush=True)
    dp.free(); ds.free()
