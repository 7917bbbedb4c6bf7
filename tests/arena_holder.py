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
dc.arena_export(sys.argv[3])
print("READY", flush=True)
sys.stdin.read()
cl.close()
