"""What one rank of an N-rank job does per MSM, measured on one GPU: the layout blz_msm_shard_layout_ex picks (or RANGES=R: the
candidate with R scalar ranges; R=1 is the plain element split), rank RANK's task in a stream of tasks, two in flight.
HOST=1: the scalars come from pageable host memory with every task (the reference's HBM flow) and the layout is asked for
with BLZ_SHARD_SCALARS_FROM_HOST.
    [HOST=1] [RANGES=R] python3 tools/shard_probe.py [logn] [world] [rank] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blaze_amd  # noqa: E402
from blaze_amd import DeviceBuffer  # noqa: E402
from blaze_amd._lib import check  # noqa: E402
from blaze_amd.driver_client import DriverClient  # noqa: E402
from blaze_amd.ingo_msm import Curve, MSMClient, MSMInit, MSMInput, MSMParams, PointMemoryType  # noqa: E402
from blaze_amd.multi_gpu import SHARD_SCALARS_FROM_HOST, shard_layout_ex  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 26
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
curve = os.environ.get("CURVE", "BLS381")
L = blaze_amd.lib()
n = 1 << logn
cid = int(Curve[curve])
host = os.environ.get("HOST") == "1"
ranges = int(os.environ["RANGES"]) if os.environ.get("RANGES") else None
lay = shard_layout_ex(Curve[curve], n, rank, world, SHARD_SCALARS_FROM_HOST if host else 0, ranges)
cnt = lay["count"]
ps = int(L.blz_point_size(cid))
dp = DeviceBuffer(0, cnt * ps)
ds = DeviceBuffer(0, cnt * 32)
check(blaze_amd.aux().blz_synth_points(0, cid, dp.ptr, cnt, 1, lay["first"]))
check(blaze_amd.aux().blz_synth_scalars_at(0, cid, ds.ptr, cnt, 0xB1A2E, lay["first"]))
L.blz_arena_release(0)
cl = MSMClient(MSMInit(PointMemoryType.HBM, False, Curve[curve]), DriverClient(0))
cl.set_scalar_range(lay["bit_lo"], lay["bit_hi"])
if os.environ.get("TABLE"):
    cl.set_window_table(int(os.environ["TABLE"]))   # window table of the rank's bases and range
cl.load_data_to_hbm(dp, 0, 0)
params = MSMParams(cnt, (0, 0))
sc_in = ds.download() if host else ds


def submit():
    cl.initialize(params); cl.start_process(); cl.set_data(MSMInput(None, sc_in, params))


def collect():
    cl.wait_result()
    r = cl.result().result
    return r, cl.get_api()


if os.environ.get("TABLE"):
    cl.prepare_window_table(cnt, (0, 0))     # built beside the tasks; waited for here
submit(); tinfo = cl.window_table_info(); submit(); r0, _ = collect(); collect()
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
a = out[-1][1]
gaps = [x[1]["total_ms"] - x[1]["phase1_accumulate_ms"] - x[1]["phase2_reduce_ms"] - x[1]["phase3_final_ms"] for x in out]
print(f"main stream waited for the sort (ev0 -> accumulate start), median over the tasks: {sorted(gaps)[len(gaps) // 2]:.3f} ms")
print(f"{curve} 2^{logn} rank {rank}/{world} {'host scalars' if host else 'resident scalars'} {lay}: {dt:.3f} ms per MSM; windows {int(a['windows'])} x {int(a['window_bits'])} bits, "
      f"accumulate {a['accumulate_kernel_ms']:.2f}, sort {a['sort_ms']:.2f} (hidden {a['sort_hidden']}), reduce {a['phase2_reduce_ms']:.2f}"
      + (f"; table {tinfo['bytes'] / 2**30:.1f} GiB built in {tinfo['build_ms']:.0f} ms" if tinfo["bytes"] else ""))
