#!/usr/bin/env python3
"""Driver of tools/hybrid_probe.hip.  For f in {0.15, 0.30, 0.45}: the real XYZZ pipeline on
(1 - f) 2^26 BLS12-381 elements with the memory traffic of a batched-affine path for the other f 805 M bucket entries
running underneath k_accumulate on its own stream - the best case of a hybrid - against the plain pipeline on all 2^26."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blaze_amd
from blaze_amd import DeviceBuffer
from blaze_amd._lib import check
from blaze_amd.driver_client import DriverClient
from blaze_amd.ingo_msm import *

HP = C.CDLL(os.path.join(ROOT, "build", "libhybrid_probe.so"))
HP.hp_setup.argtypes = [C.c_int, C.c_uint64]; HP.hp_launch.argtypes = [C.c_uint64]; HP.hp_wait.restype = C.c_float
L = blaze_amd.lib()
N = 1 << 26
ENTRIES = 12 * N
assert HP.hp_setup(26, ENTRIES // 2) == 0
dp = DeviceBuffer(0, N * 96); ds = DeviceBuffer(0, N * 32)
check(blaze_amd.aux().blz_synth_points(0, 1, dp.ptr, N, 1, 0)); check(blaze_amd.aux().blz_synth_scalars(0, 1, ds.ptr, N, 7))
L.blz_arena_release(0)
cl = MSMClient(MSMInit(PointMemoryType.HBM, False, Curve.BLS381), DriverClient(0))
cl.load_data_to_hbm(dp, 0, 0)

class Sub:   # a view of the first n scalars (device pointer arithmetic without a copy)
    def __init__(self, buf, n): self.ptr, self.nbytes = buf.ptr, n * 32
DeviceBufferLike = Sub

def msm(n, affine_entries=0, delay_ms=11.5):
    """one MSM over the first n elements; optionally the affine traffic launched `delay_ms` after submission (the digit
    sort runs first on the MSM's stream: the traffic is meant to sit under k_accumulate)"""
    p = MSMParams(n, (0, 0))
    sc = ds if n == N else None
    t0 = time.perf_counter()
    cl.initialize(p); cl.start_process()
    if n == N:
        cl.set_data(MSMInput(None, ds, p))
    else:
        check(L.blz_msm_set_data_device(cl._h, None, 0, ds.ptr, n * 32, n, 1, 0, 0))
    if affine_entries:
        time.sleep(delay_ms * 1e-3)
        assert HP.hp_launch(affine_entries) == 0
    cl.wait_result(); cl.result()
    aff = HP.hp_wait() if affine_entries else 0.0
    wall = (time.perf_counter() - t0) * 1e3
    a = cl.get_api()
    return dict(wall=round(wall, 1), total=round(a["total_ms"], 1), acc=round(a["accumulate_kernel_ms"], 1), sort=round(a["sort_ms"], 1), affine=round(aff, 1))

for _ in range(2): base = msm(N)
print(f"plain XYZZ pipeline, 2^26 elements:               {base}", flush=True)
assert HP.hp_launch(ENTRIES // 4) == 0
print(f"affine-path traffic alone, 25 % of the entries:   {HP.hp_wait():.1f} ms", flush=True)
for f in (0.15, 0.30, 0.45):
    n = int(N * (1 - f)) & ~1023
    ent = int(ENTRIES * f)
    for _ in range(2): alone = msm(n)
    for _ in range(2): both = msm(n, ent)
    lower = both["sort"] + max(both["acc"], both["affine"]) + (base["total"] - base["sort"] - base["acc"])
    print(f"f = {f:.2f}: XYZZ on {n} elements alone {alone}; with the affine share's traffic underneath {both}; "
          f"hybrid lower bound {lower:.1f} ms vs plain {base['total']} ms -> gain at most {base['total'] - lower:.1f} ms", flush=True)
