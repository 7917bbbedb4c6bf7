#!/usr/bin/env python3
"""Dev tool: can a small-footprint memory-bound kernel run UNDER the bucket accumulation (2 waves x 193 VGPRs per SIMD
leave 112 VGPRs free)?  Runs the MSM loop alone, then with device-to-device tensor copies looping on a second
stream, and reports MSM ms/step and the copy bandwidth in both situations."""
import os, sys, time, threading
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
c = MSMClient(MSMInit(PointMemoryType.HBM, False, Curve.BLS381), DriverClient(0))
c.load_data_to_hbm(d_pts, 0, 0)
params = MSMParams(n, (0, 0))

def run(k):
    pend = 0
    for _ in range(k):
        c.initialize(params); c.start_process(); c.set_data(MSMInput(None, d_sc, params)); pend += 1
        if pend >= 2:
            c.wait_result(); c.result(); pend -= 1
    while pend:
        c.wait_result(); c.result(); pend -= 1

src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0")   # 1 GiB
dst = torch.empty_like(src)
side = torch.cuda.Stream()
stop = False
copied = [0]

def copier():
    with torch.cuda.stream(side):
        while not stop:
            for _ in range(4):
                dst.copy_(src)
            side.synchronize()
            copied[0] += 4

run(2); torch.cuda.synchronize()
t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"MSM alone: {dt / steps * 1e3:.2f} ms/step", flush=True)
with torch.cuda.stream(side):
    t0 = time.perf_counter()
    for _ in range(16): dst.copy_(src)
    side.synchronize(); d = time.perf_counter() - t0
print(f"copy alone: {16 * 2 * src.numel() / d / 1e9:.0f} GB/s (read + write)", flush=True)
th = threading.Thread(target=copier); th.start()
time.sleep(0.2)
c0 = copied[0]; t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0; c1 = copied[0]
stop = True; th.join()
print(f"MSM with copies underneath: {dt / steps * 1e3:.2f} ms/step; copies meanwhile: {(c1 - c0) * 2 * src.numel() / dt / 1e9:.0f} GB/s", flush=True)
