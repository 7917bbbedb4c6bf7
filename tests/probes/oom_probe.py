#!/usr/bin/env python3
"""Dev tool / test body: the device nearly full (all but 600 MiB taken by a hog) - a DMA-mode task, an arena load and an NTT client
must each fail with an error of the reference's enum that names the allocation (never a crash, never a stale error of an earlier
failure), and every client must work again once the memory is back.
    python3 tests/probes/oom_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import blaze_amd  # noqa: E402
import oracle  # noqa: E402
from blaze_amd import DeviceBuffer, DriverClientError  # noqa: E402
from blaze_amd.driver_client import DriverClient  # noqa: E402
from blaze_amd.ingo_msm import PointMemoryType  # noqa: E402
from blaze_amd.ingo_ntt import NTT, NTTClient  # noqa: E402
from gpu_util import msm_client, run_msm, synth  # noqa: E402

import torch  # noqa: E402  (device memory figures only)

curve, n = "BLS381", 1 << 22
dp, ds = synth(curve, n)
k = oracle.index_weighted_sum(curve, ds.download(), n, 0, threads=8)
exp = oracle.result_from_affine(curve, oracle.generator_mul(curve, k))
free, _total = torch.cuda.mem_get_info(0)
hogs, left = [], free - (600 << 20)      # a 2^22 task needs ~3 GiB of workspace and copies
while left > 0:
    sz = min(left, 32 << 30)
    hogs.append(DeviceBuffer(0, sz))
    left -= sz
print(f"free: {free / 2**30:.1f} GiB, with the hog {torch.cuda.mem_get_info(0)[0] / 2**20:.0f} MiB")
problems = []


def must_fail(what, f, names="hipMalloc"):
    try:
        f()
        problems.append(f"{what}: succeeded with the device full")
    except DriverClientError as e:
        print(f"{what}: {e.variant}: {str(e)[:140]}")
        if names not in str(e) or "out of memory" not in str(e) or "hipGetLastError" in str(e):
            problems.append(f"{what}: the error does not name the call that failed: {e}")


cl = msm_client(curve, 1)
for attempt in (1, 2):   # the second attempt must report ITS allocation, not find the first one's error lying around
    must_fail(f"DMA-mode task, attempt {attempt}", lambda: run_msm(cl, dp, ds, n))
    cl.reset()
h = msm_client(curve, 1, PointMemoryType.HBM)
must_fail("load_data_to_hbm", lambda: h.load_data_to_hbm(dp, 0, 0))
must_fail("NTT client of 2^24", lambda: NTTClient(NTT.Ntt, DriverClient(0), 24), names="failed")
for b in hogs:
    b.free()
ok = [run_msm(cl, dp, ds, n) == exp]
h.load_data_to_hbm(dp, 0, 0)
ok.append(run_msm(h, None, ds, n, hbm=(0, 0)) == exp)
nt = NTTClient(NTT.Ntt, DriverClient(0), 20)
nt.initialize()
ok.append(nt.info()["device_bytes"] > 0)
nt.close()
print("after the hog is gone: DMA-mode task, HBM task, NTT client:", ok)
if not all(ok):
    problems.append(f"a client did not recover: {ok}")
blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
for p in problems:
    print("PROBLEM:", p)
sys.exit(1 if problems else 0)
