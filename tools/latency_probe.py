"""One MSM at a time (the reference's integration tests time single tasks): median wall time of set_data -> result over
resident inputs.  python3 tools/latency_probe.py logn [reps]"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blaze_amd  # noqa: E402
from blaze_amd import DeviceBuffer  # noqa: E402
from blaze_amd._lib import check  # noqa: E402
from blaze_amd.driver_client import DriverClient  # noqa: E402
from blaze_amd.ingo_msm import Curve, MSMClient, MSMInit, MSMInput, MSMParams, PointMemoryType  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
curve = os.environ.get("CURVE", "BLS381")
L = blaze_amd.lib()
n = 1 << logn
cid = int(Curve[curve])
dp = DeviceBuffer(0, n * int(L.blz_point_size(cid)))
ds = DeviceBuffer(0, n * 32)
check(blaze_amd.aux().blz_synth_points(0, cid, dp.ptr, n, 1, 0))
check(blaze_amd.aux().blz_synth_scalars(0, cid, ds.ptr, n, 7))
L.blz_arena_release(0)
cl = MSMClient(MSMInit(PointMemoryType.HBM, False, Curve[curve]), DriverClient(0))
cl.load_data_to_hbm(dp, 0, 0)
params = MSMParams(n, (0, 0))
table = os.environ.get("TABLE", "0") != "0"     # TABLE=1: the resident-base window table (opt-in), paid up front
if table:
    cl.set_window_table(2)
    print("window table ready:", cl.prepare_window_table(n, (0, 0)), cl.window_table_info() if hasattr(cl, "window_table_info") else "")
ts, dev, calls = [], [], []
inp = MSMInput(None, ds, params)
for k in range(reps + 2):
    t0 = time.perf_counter()
    cl.initialize(params); t1 = time.perf_counter()
    cl.start_process(); t2 = time.perf_counter()
    cl.set_data(inp); t3 = time.perf_counter()
    cl.wait_result(); t4 = time.perf_counter()
    cl.result(); t5 = time.perf_counter()
    if k >= 2:
        ts.append((t5 - t0) * 1e3)
        calls.append([(b - a) * 1e3 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))])
        dev.append(cl.get_api()["total_ms"])
a = cl.get_api()
print(f"{curve} 2^{logn}{' (window table)' if table else ''} one at a time: wall {statistics.median(ts):.3f} ms, device pipeline {statistics.median(dev):.3f} ms "
      f"[calls: " + " ".join(f"{n} {statistics.median(c[i] for c in calls):.3f}" for i, n in enumerate(("initialize", "start", "set_data", "wait", "result"))) + "] "
      f"(sort {a['sort_ms']:.2f}, accumulate {a['accumulate_kernel_ms']:.2f}, reduce {a['phase2_reduce_ms']:.2f}, finish {a['phase3_final_ms']:.2f})")
