#!/usr/bin/env python3
"""Dev tool: config 3 (2^26 BN254 elements, precompute factor 8: 2^29 bases = 32 GiB resident in the arena, scalars-only
set_data with device-resident scalars), a stream of tasks two in flight and lone tasks.
    [PC_PLAN=1] [BLAZE_MSM_PIECES=k] python tools/config3_probe.py [logn] [curve]
PC_PLAN=1: the handle opts in to the checked-table plan (blz_msm_set_precompute_plan): 4n even bases, 64-bit chunks."""
import os, sys, time, json, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import blaze_amd
from blaze_amd.ingo_msm import MSMInput, MSMParams, PointMemoryType
from gpu_util import msm_client, synth

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 26
curve = sys.argv[2] if len(sys.argv) > 2 else "BN254"
n = 1 << logn
dp, ds = synth(curve, n, pf=8)
blaze_amd.lib().blz_arena_release(0)
# (PRE_CLIENTS=k: k other handles created - and, PRE_CLOSE=1, closed - first: where do this handle's streams land among the
# process's hardware queues?)
pre = [msm_client("BLS381", 1) for _ in range(int(os.environ.get("PRE_CLIENTS", "0")))]
if os.environ.get("PRE_CLOSE", "0") == "1":
    for c_ in pre:
        c_.close()
cl = msm_client(curve, 8, PointMemoryType.HBM)
cl.load_data_to_hbm(dp, 0, 0)
dp.free()
plan = os.environ.get("PC_PLAN", "0") == "1"
if plan:
    cl.set_precompute_plan(True)
    assert cl.prepare_precompute_plan(n, (0, 0))
p = MSMParams(n, (0, 0))
def submit():
    cl.initialize(p); cl.start_process(); cl.set_data(MSMInput(None, ds, p))
def collect():
    cl.wait_result(); return cl.result().result, cl.get_api()
submit(); r0, _ = collect()
lone = []
for _ in range(3):
    t = time.perf_counter(); submit(); r, a = collect(); lone.append((time.perf_counter() - t) * 1e3); assert r == r0
done, apis = [], []
submit()
for _ in range(6):
    submit(); r, a = collect(); assert r == r0; done.append(time.perf_counter()); apis.append(a)
r, a = collect(); done.append(time.perf_counter()); apis.append(a)
gaps = [(b - a_) * 1e3 for a_, b in zip(done, done[1:])]
# bench.py's config 3 key: bursts of four tasks, two in flight, from an idle handle (the ramp and the last tail included)
bursts = []
for _ in range(3):
    t = time.perf_counter()
    submit(); submit(); collect(); submit(); collect(); submit(); collect(); collect()
    bursts.append((time.perf_counter() - t) * 1e3 / 4)
    hidden = blaze_amd.lib().blz_msm_last_sort_hidden
print(json.dumps({"config": f"2^{logn} {curve} pf=8, pieces={os.environ.get('BLAZE_MSM_PIECES', 'auto')}, {'checked-table plan' if plan else 'exact path'}", "plan_info": cl.precompute_plan_info(), "lone_ms": round(min(lone), 2),
                  "two_in_flight_ms_per_msm": round(statistics.median(gaps), 2), "gaps": [round(g, 1) for g in gaps], "burst_of_4_ms_per_msm": [round(b, 1) for b in bursts], "accumulate_kernel_ms": round(apis[-2]["accumulate_kernel_ms"], 2),
                  "phases": {k: round(v, 2) for k, v in apis[-2].items() if k.endswith("_ms")}}))
