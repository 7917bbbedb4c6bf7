#!/usr/bin/env python3
"""Dev tool: the reference's largest DMA-mode shape (tests/integration_msm.rs:386-467: 2^26 elements x PRECOMPUTE_FACTOR 8, a 48 GiB
host vector handed to ONE set_data) as a STREAMED task - queued once, fed from host slices - with the per-slice host timeline, for
slices of 3 GiB / 768 MiB / 96 MiB of points and pageable or page-locked host memory; dur_set_data / dur_wait_result as
tests/integration_msm.rs:338-355 times them.  One JSON line per variant; the result of every variant is checked (linearity over
B_ij = 2^(32 j) (i + 1) G).      python3 tests/probes/stream_probe.py [CURVE] [logn] [pf]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np  # noqa: E402

import blaze_amd  # noqa: E402
import oracle  # noqa: E402
from blaze_amd._lib import HostBuffer  # noqa: E402
from blaze_amd.ingo_msm import MSMInput, MSMParams  # noqa: E402
from gpu_util import msm_client, synth  # noqa: E402

curve = sys.argv[1] if len(sys.argv) > 1 else "BLS381"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 26
pf = int(sys.argv[3]) if len(sys.argv) > 3 else 8
n = 1 << logn
ps = 64 if curve == "BN254" else 96
blaze_amd._lib.check(blaze_amd.lib().blz_arena_release(0))
dp, ds = synth(curve, n, pf=pf, seed=0x2626)
sc = np.frombuffer(ds.download(), dtype=np.uint8)
k = oracle.index_weighted_sum(curve, sc, n, 0, threads=16)
exp = oracle.result_from_affine(curve, oracle.generator_mul(curve, k))
cl = msm_client(curve, pf)
for slice_log, pinned in ((22, False), (22, True), (20, True), (17, True), (22, False)):
    step = 1 << min(slice_log, logn)
    nb = step * pf * ps
    if pinned:
        hb = HostBuffer(0, nb)
        host = hb.array()
    else:
        hb = None
        host = np.empty(nb, dtype=np.uint8)
    cl.initialize(MSMParams(n, None)); cl.start_process()
    per, t_set = [], 0.0
    for a in range(0, n, step):
        blaze_amd._lib.check(blaze_amd.lib().blz_memcpy_d2h(0, host.ctypes.data, dp.ptr + a * pf * ps, nb))     # (the next slice "arrives" in host memory)
        t0 = time.perf_counter()
        cl.set_data(MSMInput(host, sc[a * 32: (a + step) * 32], MSMParams(step, None)))
        dt = time.perf_counter() - t0
        per.append(round(dt * 1e3, 2))
        t_set += dt
    t0 = time.perf_counter()
    cl.wait_result()
    t_wait = time.perf_counter() - t0
    ok = cl.result().result == exp
    api = cl.get_api()
    gib = n * (pf * ps + 32) / 2**30
    rec = {"curve": curve, "log_n": logn, "pf": pf, "slices": n // step, "slice_points_bytes": nb, "host_memory": "page-locked (blz_host_malloc)" if pinned else "pageable (numpy)",
           "dur_set_data_ms": round(t_set * 1e3, 1), "dur_wait_result_ms": round(t_wait * 1e3, 1), "host_gib": round(gib, 2),
           "link_gb_per_s_over_set_data": round(n * (pf * ps + 32) / t_set / 1e9, 1), "first_slices_ms": per[:3], "median_slice_ms": float(np.median(per)),
           "last_slice_ms": per[-1], "windows": api["windows"], "window_bits": api["window_bits"], "accumulate_kernel_ms_sum": round(api["accumulate_kernel_ms"], 1),
           "result_ok": bool(ok)}
    print(json.dumps(rec), flush=True)
    del host
    if hb is not None:
        hb.free()
cl.close(); dp.free(); ds.free()
