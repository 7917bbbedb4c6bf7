#!/usr/bin/env python3
"""Dev tool: do two independent MSM pipelines on one GPU overlap (one task's digit sort under the other's bucket
accumulation)?  Two clients over the same arena points, tasks interleaved; compares MSM/s with one client."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import blaze_amd
from blaze_amd import DeviceBuffer
from blaze_amd._lib import check
from blaze_amd.driver_client import DriverClient
from blaze_amd.ingo_msm import Curve, MSMClient, MSMInit, MSMInput, MSMParams, PointMemoryType

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 26
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = blaze_amd.lib()
n = 1 << logn
cid = int(Curve.BLS381)
d_pts = DeviceBuffer(0, n * 96)
d_sc = DeviceBuffer(0, n * 32)
check(blaze_amd.aux().blz_synth_points(0, cid, d_pts.ptr, n, 1, 0))
check(blaze_amd.aux().blz_synth_scalars_at(0, cid, d_sc.ptr, n, 0xB1A2E, 0))
L.blz_arena_release(0)
cur = Curve.BLS381
clients = [MSMClient(MSMInit(PointMemoryType.HBM, False, cur), DriverClient(0)) for _ in range(2)]
clients[0].load_data_to_hbm(d_pts, 0, 0)
params = MSMParams(n, (0, 0))

def submit(c):
    c.initialize(params); c.start_process(); c.set_data(MSMInput(None, d_sc, params))

def collect(c):
    c.wait_result(); return c.result().result

def run(cs, k, depth):
    pend = []
    out = None
    for i in range(k):
        c = cs[i % len(cs)]
        submit(c); pend.append(c)
        if len(pend) >= depth:
            out = collect(pend.pop(0))
    while pend:
        out = collect(pend.pop(0))
    return out

for name, cs, depth in (("one client, 2 in flight", clients[:1], 2), ("two clients, 2 in flight", clients, 2),
                        ("two clients, 4 in flight", clients, 4), ("one client, 2 in flight", clients[:1], 2)):
    run(cs, 2, depth); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = run(cs, steps, depth); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {dt / steps * 1e3:.2f} ms/MSM  {steps / dt:.3f} MSM/s  result[:8]={bytes(r)[:8].hex()}", flush=True)
