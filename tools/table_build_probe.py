"""Build time of the resident-base window table (k_build_window_table) per curve: python3 tools/table_build_probe.py logn
(BLAZE_HIP_LIB selects the library: A/B runs of the build kernel on one box).  The result of a task over the table is compared
with the plain path's."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import blaze_amd
from blaze_amd.ingo_msm import PointMemoryType
from gpu_util import msm_client, run_msm, synth

lg = int(sys.argv[1])
n = 1 << lg
for curve in ("BLS381", "BLS377", "BN254"):
    blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
    dp, ds = synth(curve, n)
    plain = msm_client(curve, 1, PointMemoryType.HBM)
    plain.load_data_to_hbm(dp, 0, 0)
    want = run_msm(plain, None, ds, n, hbm=(0, 0))
    cl = msm_client(curve, 1, PointMemoryType.HBM)
    cl.set_window_table(2)
    t0 = time.perf_counter()
    ok = cl.prepare_window_table(n, (0, 0))
    wall = (time.perf_counter() - t0) * 1e3
    got = run_msm(cl, None, ds, n, hbm=(0, 0))
    info = cl.window_table_info()
    print(f"{curve} 2^{lg} bases: table ready={ok} in {wall:.1f} ms (first chunk .. last: {info['build_ms']:.1f} ms; {info['windows']} windows of "
          f"{info['window_bits']} bits, {info['bytes'] / 2**30:.2f} GiB), result equals the plain path's: {got == want}", flush=True)
    cl.close(); plain.close(); dp.free(); ds.free()
blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
