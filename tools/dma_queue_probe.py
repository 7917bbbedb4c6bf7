#!/usr/bin/env python3
"""Dev tool: DMA-mode tasks (host points + scalars) from a handle opened after PRE_CLIENTS other handles were opened and
(PRE_CLOSE=1) closed: do the copy stream's waits still run beside the main stream's kernels, wherever the runtime put the streams?
    PRE_CLIENTS=k PRE_CLOSE=1 python tools/dma_queue_probe.py [logn]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import blaze_amd
from blaze_amd.ingo_msm import MSMInput, MSMParams
from gpu_util import msm_client, synth

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << logn
dp, ds = synth("BLS381", n)
pts, sc = bytes(dp.download()), bytes(ds.download())
dp.free(); ds.free()
pre = [msm_client("BLS381", 1) for _ in range(int(os.environ.get("PRE_CLIENTS", "0")))]
if os.environ.get("PRE_CLOSE", "0") == "1":
    for c_ in pre:
        c_.close()
cl = msm_client("BLS381", 1)
p = MSMParams(n, None)
def submit():
    cl.initialize(p); cl.start_process(); cl.set_data(MSMInput(pts, sc, p))
def collect():
    cl.wait_result(); return cl.result().result
submit(); r0 = collect()
lone = []
for _ in range(4):
    t = time.perf_counter(); submit(); assert collect() == r0; lone.append((time.perf_counter() - t) * 1e3)
done = []
submit()
for _ in range(8):
    submit(); assert collect() == r0; done.append(time.perf_counter())
collect()
gaps = [(b - a) * 1e3 for a, b in zip(done, done[1:])]
print(f"2^{logn} DMA pre={os.environ.get('PRE_CLIENTS', '0')} close={os.environ.get('PRE_CLOSE', '0')}: lone {min(lone):.2f} ms, two in flight {statistics.median(gaps):.2f} ms per MSM")
