#!/usr/bin/env python3
"""Dev tool: the CPU baseline at the FULL size of the headline workload, once (bench.py times a ~20 s prefix and scales it linearly:
this is the anchor for that scaling).  The oracle's threaded Pippenger (test infrastructure, here the thing timed as the baseline -
never the product) over all 2^logn BLS12-381 elements of bench.py's synthetic input on the box's usable host threads, result checked by
linearity.  One JSON line.      python3 tests/probes/cpu_baseline_full.py [logn] [cbits]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import blaze_amd  # noqa: E402
import oracle  # noqa: E402
from blaze_amd import DeviceBuffer  # noqa: E402
from blaze_amd._lib import check  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 26
cbits = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = 1 << logn
CURVE, cid = "BLS381", 1
try:
    cores = len(os.sched_getaffinity(0))
except AttributeError:
    cores = os.cpu_count() or 1
try:   # cgroup quota (bench.py host_threads())
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max":
        cores = max(1, min(cores, int(int(q) / int(p))))
except Exception:   # noqa: BLE001
    pass
d_pts, d_sc = DeviceBuffer(0, n * 96), DeviceBuffer(0, n * 32)
check(blaze_amd.aux().blz_synth_points(0, cid, d_pts.ptr, n, 1, 0))
check(blaze_amd.aux().blz_synth_scalars_at(0, cid, d_sc.ptr, n, 0xB1A2E, 0))     # bench.py's inputs
pts, sc = d_pts.download(), d_sc.download()
d_pts.free(); d_sc.free()
recs = []
for ln in sorted({min(logn, 23), logn}):
    ns = 1 << ln
    t0 = time.perf_counter()
    got = oracle.msm_pippenger(CURVE, pts[: ns * 96], sc[: ns * 32], ns, 1, threads=cores, cbits=cbits)
    dt = time.perf_counter() - t0
    kk = oracle.index_weighted_sum(CURVE, sc[: ns * 32], ns, 0, threads=min(64, cores))
    ok = got == oracle.result_from_affine(CURVE, oracle.generator_mul(CURVE, kk))
    recs.append({"log_n": ln, "seconds": round(dt, 2), "msm_per_s": round(1.0 / dt, 6), "result_ok": bool(ok)})
full = recs[-1]
line = {"what": f"oracle Pippenger (C, {cbits}-bit signed windows, pthreads) over ALL 2^{logn} BLS12-381 elements of bench.py's synthetic input",
        "cores": cores, "kind": "port", "value": full["msm_per_s"], "unit": "MSM/s", "seconds": full["seconds"], "result_ok": full["result_ok"], "runs": recs}
if len(recs) > 1:
    scaled = recs[0]["msm_per_s"] / (1 << (logn - recs[0]["log_n"]))
    line["prefix_scaled_linearly"] = {"from_log_n": recs[0]["log_n"], "msm_per_s": round(scaled, 6), "ratio_measured_over_scaled": round(full["msm_per_s"] / scaled, 3)}
print(json.dumps(line), flush=True)
