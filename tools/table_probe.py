"""Window-table tasks alone (for rocprofv3 --kernel-trace --stats): 2^LOGN BLS12-381 bases in the arena, K tasks, two in
flight.  python3 tools/table_probe.py [logn] [steps] [table 0/1]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blaze_amd  # noqa: E402
from blaze_amd import DeviceBuffer  # noqa: E402
from blaze_amd._lib import check  # noqa: E402
from blaze_amd.driver_client import DriverClient  # noqa: E402
from blaze_amd.ingo_msm import Curve, MSMClient, MSMInit, MSMInput, MSMParams, PointMemoryType  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 26
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
table = int(sys.argv[3]) if len(sys.argv) > 3 else 1
curve = os.environ.get("CURVE", "BLS381")
L = blaze_amd.lib()
n = 1 << logn
cid = int(Curve[curve])
ps = int(L.blz_point_size(cid))
dp = DeviceBuffer(0, n * ps)
ds = DeviceBuffer(0, n * 32)
check(blaze_amd.aux().blz_synth_points(0, cid, dp.ptr, n, 1, 0))
check(blaze_amd.aux().blz_synth_scalars(0, cid, ds.ptr, n, 7))
L.blz_arena_release(0)
cl = MSMClient(MSMInit(PointMemoryType.HBM, False, Curve[curve]), DriverClient(0))
cl.set_window_table(bool(table))
cl.load_data_to_hbm(dp, 0, 0)
params = MSMParams(n, (0, 0))


def submit():
    cl.initialize(params); cl.start_process(); cl.set_data(MSMInput(None, ds, params))


def collect():
    cl.wait_result()
    r = cl.result().result
    return r, cl.get_api()


# the table is built beside the tasks: a stream of tasks, two in flight, from the moment the bases are loaded; per collected task:
# ms since the first submission, the interval since the previous result, whether the NEXT task was launched off the table
t_start = time.perf_counter()
submit()
r0, _ = collect()
print(f"first task (plain path, build enqueued): {(time.perf_counter() - t_start) * 1e3:.1f} ms")
pend, last, seen_table, extra = 0, time.perf_counter(), False, 0
ntasks = 0
while extra < steps and time.perf_counter() - t_start < 120:
    ntasks += 1
    if table and ntasks == 7 and not seen_table:     # six tasks paid their chunks; now the host would rather have the table
        t_p = time.perf_counter()
        cl.prepare_window_table(n, (0, 0), -1)
        print(f"prepare_window_table(wait): {(time.perf_counter() - t_p) * 1e3:.1f} ms")
    submit(); pend += 1
    used = cl.window_table_info()["bytes"] > 0
    if pend >= 2:
        r, a = collect(); pend -= 1
        now = time.perf_counter()
        assert r == r0
        print(f"  t = {(now - t_start) * 1e3:8.1f} ms  interval {(now - last) * 1e3:7.1f} ms  accumulate {a['accumulate_kernel_ms']:.1f}  windows {int(a['windows'])}  next launched off the table: {used}")
        last = now
    if used:
        if not seen_table:
            print(f"table in use {(time.perf_counter() - t_start) * 1e3:.1f} ms after the first submission:", cl.window_table_info())
        seen_table = True
    if seen_table or not table:
        extra += 1
while pend:
    collect(); pend -= 1
submit(); submit(); collect(); collect()
t0 = time.perf_counter()
pend, out = 0, []
for _ in range(steps):
    submit(); pend += 1
    if pend >= 2:
        out.append(collect()); pend -= 1
while pend:
    out.append(collect()); pend -= 1
dt = (time.perf_counter() - t0) / steps * 1e3
assert all(r == r0 for r, _ in out)
print(f"{curve} 2^{logn} table={table}: {dt:.3f} ms per MSM; phases of the last task: {out[-1][1]}")
