#!/usr/bin/env python3
"""Dev tool: a dense walk over task SIZES - 2^k - 1, 2^k, 2^k + 1, 3 2^(k-1) for k = 1 .. max_log - on the three curves, pf = 1 (device
buffers and arena bases in turn, two tasks in flight) and pf = 8 (exact path and checked-table plan), every result checked through
linearity (P_i = (i + 1) G).  The sizes where the planner changes structure (tiny sort / big sort, row-law levels, hot top windows,
window widths) all lie on the way.   python3 tests/probes/msm_sizes_probe.py [max_log] [max_log_pf8]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np  # noqa: E402

import blaze_amd  # noqa: E402
import oracle  # noqa: E402
from blaze_amd.ingo_msm import MSMInput, MSMParams, PointMemoryType  # noqa: E402
from gpu_util import msm_client, synth  # noqa: E402

max_log = int(sys.argv[1]) if len(sys.argv) > 1 else 22
max_log8 = int(sys.argv[2]) if len(sys.argv) > 2 else max_log - 3
L = blaze_amd.lib()


class View(blaze_amd.DeviceBuffer):
    def __init__(self, base, nbytes):
        self.device_id, self.nbytes, self.ptr = base.device_id, nbytes, base.ptr

    def free(self):
        pass


def sizes(mx):
    out = set()
    for k in range(1, mx + 1):
        for n in ((1 << k) - 1, 1 << k, (1 << k) + 1, 3 << (k - 1)):
            if 1 <= n <= (1 << mx):
                out.add(n)
    return sorted(out)


bad, t0 = 0, time.time()
for curve in ("BLS381", "BLS377", "BN254"):
    ps = 64 if curve == "BN254" else 96
    for pf, mx in ((1, max_log), (8, max_log8)):
        blaze_amd._lib.check(L.blz_arena_release(0))
        nmax = 1 << mx
        dp, ds = synth(curve, nmax, pf=pf, seed=3)
        sc = np.frombuffer(ds.download(), dtype=np.uint8).reshape(nmax, 32)
        hbm = msm_client(curve, pf, PointMemoryType.HBM)
        hbm.load_data_to_hbm(dp, 0, 0)
        if pf == 8:
            plan = msm_client(curve, 8, PointMemoryType.HBM)
            plan.set_precompute_plan(True)
            clients = [("exact", hbm, True), ("plan", plan, True)]
        else:
            clients = [("device buffers", msm_client(curve, 1), False), ("arena", hbm, True)]
        pending = []

        def collect():
            global bad
            what, cl, exp = pending.pop(0)
            cl.wait_result()
            if cl.result().result != exp:
                bad += 1
                print("MISMATCH", what, flush=True)

        for idx, n in enumerate(sizes(mx)):
            k = oracle.index_weighted_sum(curve, sc[:n].tobytes(), n, 0, threads=8)
            exp = oracle.result_from_affine(curve, oracle.generator_mul(curve, k))
            for name, cl, from_arena in (clients if pf == 8 else [clients[idx & 1]]):
                while len([p for p in pending if p[1] is cl]) >= 2:
                    collect()
                prm = MSMParams(n, (0, 0) if from_arena else None)
                cl.initialize(prm); cl.start_process()
                cl.set_data(MSMInput(None if from_arena else View(dp, n * pf * ps), View(ds, n * 32), prm))
                pending.append((f"{curve} pf={pf} n={n} {name}", cl, exp))
        while pending:
            collect()
        print(f"{curve} pf={pf}: {len(sizes(mx))} sizes up to 2^{mx}, mismatches so far {bad} ({time.time() - t0:.0f} s)", flush=True)
        for _, cl, _f in clients:
            cl.close()
        dp.free(); ds.free()
blaze_amd._lib.check(L.blz_arena_release(0))
print("mismatches:", bad)
sys.exit(1 if bad else 0)
