#!/usr/bin/env python3
"""Dev tool: the reference's HBM flow (bases in the arena, scalars from pageable host memory with every task), one task at a
time and two in flight.  usage: [BLAZE_MSM_PIECES=k] python tools/hbm_flow_probe.py [logn]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import blaze_amd
from blaze_amd.ingo_msm import MSMInput, MSMParams, PointMemoryType
from gpu_util import msm_client, synth

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << logn
dp, ds = synth("BLS381", n)
sc = ds.download()
blaze_amd.lib().blz_arena_release(0)
cl = msm_client("BLS381", 1, PointMemoryType.HBM)
cl.load_data_to_hbm(dp, 0, 0)
p = MSMParams(n, (0, 0))
rows = []
for rep in range(5):
    t0 = time.perf_counter(); cl.initialize(p); cl.start_process()
    cl.set_data(MSMInput(None, sc, p)); t1 = time.perf_counter()
    cl.wait_result(); r = cl.result().result; t2 = time.perf_counter()
    rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3))
best = min(rows[1:], key=lambda x: x[2])
done = []
def submit():
    cl.initialize(p); cl.start_process(); cl.set_data(MSMInput(None, sc, p))
def collect():
    cl.wait_result(); assert cl.result().result == r; done.append(time.perf_counter())
submit()
for _ in range(7):
    submit(); collect()
collect()
gaps = sorted((b - a) * 1e3 for a, b in zip(done, done[1:]))
print(json.dumps({"config": f"2^{logn} BLS12-381, bases in the arena, host scalars", "lone_task_set_data_ms": round(best[0], 2),
                  "lone_task_wait_result_ms": round(best[1], 2), "lone_task_full_ms": round(best[2], 2),
                  "two_in_flight_steady_ms_per_msm": round(gaps[len(gaps) // 2], 2)}))
