"""Time of the checked-table plan's device check (k_check_precompute) per curve: python3 tools/check_probe.py logn [repeats]
(BLAZE_HIP_LIB selects the library: A/B runs of the kernel's formulas on one box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import blaze_amd
from blaze_amd.ingo_msm import PointMemoryType
from gpu_util import msm_client, synth

lg = int(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 1 << lg
for curve in ("BN254", "BLS381", "BLS377"):
    ms = []
    for r in range(reps):
        blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
        dp, ds = synth(curve, n, pf=8)
        cl = msm_client(curve, 8, PointMemoryType.HBM)
        cl.set_precompute_plan(True)
        cl.load_data_to_hbm(dp, 0, 0)
        ok = cl.prepare_precompute_plan(n, (0, 0))
        ms.append(cl.precompute_plan_info()["check_ms"])
        assert ok
        cl.close(); dp.free(); ds.free()
    print(f"{curve} 2^{lg} elements x 8 bases: check {min(ms):.1f} ms (of {[round(m, 1) for m in ms]})", flush=True)
blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
