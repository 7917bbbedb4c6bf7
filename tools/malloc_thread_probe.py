#!/usr/bin/env python3
"""Dev tool: does a big hipMalloc on another host thread stall this thread's task stream?  (A fresh 80 GiB allocation takes
0.2 ms on some boxes and 2.5 - 3 s on others.)"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import blaze_amd
from blaze_amd import DeviceBuffer
from blaze_amd.ingo_msm import MSMInput, MSMParams, PointMemoryType
from gpu_util import msm_client, synth

n = 1 << 24
dp, ds = synth("BLS381", n)
cl = msm_client("BLS381", 1)
p = MSMParams(n, None)
def task():
    t = time.perf_counter(); cl.initialize(p); cl.start_process(); cl.set_data(MSMInput(dp, ds, p)); cl.wait_result(); cl.result()
    return (time.perf_counter() - t) * 1e3
for _ in range(3): task()
# make the next big allocation a slow one: allocate, touch nothing, free, allocate again (the driver scrubs what was freed)
b = DeviceBuffer(0, 80 << 30); b.free()
res = {}
def alloc():
    t = time.perf_counter(); res["buf"] = DeviceBuffer(0, 80 << 30); res["ms"] = (time.perf_counter() - t) * 1e3
th = threading.Thread(target=alloc); th.start()
lat = []
while th.is_alive() or len(lat) < 5:
    lat.append(round(task(), 1))
th.join()
print("80 GiB hipMalloc on a second thread took %.1f ms; task latencies meanwhile (2^24 MSM, ~36 ms alone): %s" % (res["ms"], lat[:40]))
