"""Helper process of test_arena_across_processes: plays the role of the process that loaded the bases on the
card before the test ran (tests/integration_msm_hbm.rs:51-56 keeps its load_data_to_hbm commented out).
usage: arena_holder.py <points file> <arena addr> <registry path>; prints READY, exits when stdin closes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from blaze_amd.driver_client import DriverClient  # noqa: E402
from blaze_amd.ingo_msm import Curve, MSMClient, MSMInit, PointMemoryType  # noqa: E402

pts = open(sys.argv[1], "rb").read()
addr = int(sys.argv[2])
dc = DriverClient(0)
cl = MSMClient(MSMInit(PointMemoryType.HBM, False, Curve.BLS381), dc)
cl.load_data_to_hbm(pts, addr, 0)
if os.environ.get("BLAZE_TEST_ARENA_POLICY", "0") == "1":
    # the holder runs on the arena diet: two tasks leave it without the raw bytes, and the export has to bring them back
    import blaze_amd
    from blaze_amd.ingo_msm import MSMInput, MSMParams

    blaze_amd._lib.check(blaze_amd.lib().blz_arena_set_policy(0, 1))
    n = len(pts) // 96
    for _ in range(2):
        p = MSMParams(n, (addr, 0))
        cl.initialize(p); cl.start_process(); cl.set_data(MSMInput(None, bytes(32 * n), p)); cl.wait_result(); cl.result()
    assert cl.memory_info()["arena_raw"] == 0
dc.arena_export(sys.argv[3])
print("READY", flush=True)
sys.stdin.read()
cl.close()
