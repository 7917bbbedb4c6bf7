#!/usr/bin/env python3
"""Dev tool: sweep the window size c (BLAZE_MSM_C) per problem size and print the device pipeline time."""
import sys, os, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blaze_amd
from blaze_amd import DeviceBuffer
from blaze_amd.driver_client import DriverClient
from blaze_amd.ingo_msm import *
dc = DriverClient(0)
for logn in [int(x) for x in sys.argv[1:]]:
    n = 1 << logn
    dp = DeviceBuffer(0, n * 96); ds = DeviceBuffer(0, n * 32)
    blaze_amd.lib().blz_synth_points(0, 1, dp.ptr, n, 1, 0); blaze_amd.lib().blz_synth_scalars(0, 1, ds.ptr, n, 7)
    row = []
    for c in range(max(8, logn - 7), min(23, logn - 2) + 1):
        os.environ["BLAZE_MSM_C"] = str(c)
        cl = MSMClient(MSMInit(PointMemoryType.DMA, False, Curve.BLS381), dc)
        best = 1e9
        for rep in range(3):
            p = MSMParams(n, None); cl.initialize(p); cl.start_process(); cl.set_data(MSMInput(dp, ds, p)); cl.wait_result(); cl.result()
            best = min(best, cl.get_api()["total_ms"])
        row.append((c, round(best, 2)))
        cl.close()
    print(f"2^{logn}:", row, "best", min(row, key=lambda t: t[1]), flush=True)
    dp.free(); ds.free()
