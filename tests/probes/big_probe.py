#!/usr/bin/env python3
"""Dev tool: pf = 1 MSMs beyond the reference's largest shape (2^27, 2^28 points: the window planner's u32 entry indexing stops
below 2^29), checked through linearity (P_i = (i + 1) G).   python3 tests/probes/big_probe.py CURVE logn|n [...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import blaze_amd  # noqa: E402
import oracle  # noqa: E402
from blaze_amd import DriverClientError  # noqa: E402
from blaze_amd.ingo_msm import MSMInput, MSMParams, PointMemoryType  # noqa: E402
from gpu_util import msm_client, synth  # noqa: E402

curve, _, opt = sys.argv[1].partition(":")          # CURVE, CURVE:8 (precompute handle, exact path), CURVE:8:plan (checked-table plan)
pf = 8 if opt.startswith("8") else 1
use_plan = opt.endswith("plan")
import math  # noqa: E402

for a in sys.argv[2:]:
    n = int(a) if a.isdigit() and int(a) > 64 else int(round(2 ** float(a)))   # (a count, or log2 of one)
    lg = round(math.log2(n), 3)
    blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
    try:
        dp, ds = synth(curve, n, pf=pf)
        cl = msm_client(curve, pf, PointMemoryType.HBM)
        if use_plan:
            cl.set_precompute_plan(True)
        cl.load_data_to_hbm(dp, 0, 0)
        dp.free()
        prm = MSMParams(n, (0, 0))
        outs, t0, marks = [], time.perf_counter(), []
        cl.initialize(prm); cl.start_process(); cl.set_data(MSMInput(None, ds, prm))
        marks.append(("set_data", round((time.perf_counter() - t0) * 1e3)))
        for i in range(3):
            if i < 2:
                cl.initialize(prm); cl.start_process(); cl.set_data(MSMInput(None, ds, prm))
                marks.append(("set_data", round((time.perf_counter() - t0) * 1e3)))
            cl.wait_result(); outs.append(cl.result().result)
            marks.append(("result", round((time.perf_counter() - t0) * 1e3)))
        dt = (time.perf_counter() - t0) / 3 * 1e3
        print("   host timeline (ms since the first call):", marks, flush=True)
        api = cl.get_api()
    except DriverClientError as e:
        print(f"{curve} n = {n} (2^{lg}): {e.variant}: {str(e)[:200]}", flush=True)
        continue
    k = oracle.index_weighted_sum(curve, ds.download(), n, 0, threads=16)
    exp = oracle.result_from_affine(curve, oracle.generator_mul(curve, k))
    tag = " plan" if use_plan else ""
    print(f"{curve} pf={pf}{tag} n = {n} (2^{lg}): {dt:.1f} ms per MSM (three, two in flight), windows {api['windows']:.0f} x {api['window_bits']:.0f} bits, accumulate "
          f"{api['accumulate_kernel_ms']:.1f} ms, results right: {[o == exp for o in outs]}, device memory {api['device_memory']['total'] / 2**30:.1f} GiB; phases "
          f"{ {k: round(v, 1) for k, v in api.items() if k.endswith('_ms')} }", flush=True)
    cl.close(); ds.free()
blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
