#!/usr/bin/env python3
"""Dev tool: long-lived clients fed a random mix of task kinds - plain, resident-base window table (arena bases, sub-ranges,
rewrites that drop the table), scalar ranges, host buffers (DMA mode and the HBM flow with host scalars: tasks enqueued piece by
piece, BLAZE_MSM_PIECES drawn per task), precompute clients over a resident x8 table on the checked-table plan and on the exact
path (sub-ranges on the element grid, rewrites that re-arm the check, host scalars) - at random sizes, two in flight, every result
checked through linearity (P_i = (i + 1) G).  STRESS_DIET=1: the arena drops raw bytes (blz_arena_set_policy).
    python3 tests/probes/stress_modes.py [iterations] [seed]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np  # noqa: E402

import blaze_amd  # noqa: E402
import oracle  # noqa: E402
from blaze_amd._lib import check  # noqa: E402
from blaze_amd.ingo_msm import Curve, MSMInput, MSMParams, PointMemoryType  # noqa: E402
from gpu_util import msm_client, synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
L = blaze_amd.lib()
NMAX = 1 << int(os.environ.get("STRESS_LOG_NMAX", "21"))   # (2^24: the piecewise paths at their default piece counts)
curves = ("BLS381", "BLS377", "BN254")
host_sc, host_pts, dev = {}, {}, {}
L.blz_arena_release(0)
for c in curves:
    dp, ds = synth(c, NMAX, seed=13)
    dev[c] = (dp, ds)
    host_sc[c] = np.frombuffer(ds.download(), dtype=np.uint8).reshape(NMAX, 32).copy()
    host_pts[c] = bytes(dp.download())


class View(blaze_amd.DeviceBuffer):
    def __init__(self, base, off, nbytes):
        self.device_id, self.nbytes, self.ptr = base.device_id, nbytes, base.ptr + off

    def free(self):
        pass


def expected(c, first, n, lo, hi):
    sc = host_sc[c][first:first + n].copy()
    if (lo, hi) != (0, 256):
        sc[:, : lo // 8] = 0
        sc[:, hi // 8:] = 0
    k = oracle.index_weighted_sum(c, sc.tobytes(), n, first, threads=8)
    return oracle.result_from_affine(c, oracle.generator_mul(c, k))


if os.environ.get("STRESS_DIET") == "1":
    check(L.blz_arena_set_policy(0, 1))
# precompute tables: N8 elements x 8 bases per curve, resident at their own arena address; one client on the plan, one exact
N8 = NMAX // 8
ARENA8 = {c: (i + 9) << 32 for i, c in enumerate(curves)}
pc_plan = {c: msm_client(c, 8, PointMemoryType.HBM) for c in curves}
pc_exact = {c: msm_client(c, 8, PointMemoryType.HBM) for c in curves}
dev8 = {}
for c in curves:
    dp8, _ = synth(c, N8, pf=8, seed=13)
    dev8[c] = dp8
    pc_plan[c].set_precompute_plan(True)
    pc_plan[c].load_data_to_hbm(dp8, ARENA8[c], 0)
plain = {c: msm_client(c, 1) for c in curves}
table = {c: msm_client(c, 1, PointMemoryType.HBM) for c in curves}
hbm = {c: msm_client(c, 1, PointMemoryType.HBM) for c in curves}
ARENA = {c: (i + 1) << 32 for i, c in enumerate(curves)}   # one flat arena per device: every curve's bases at their own address
for c in curves:
    table[c].set_window_table(2)
    table[c].load_data_to_hbm(dev[c][0], ARENA[c], 0)
ps = {c: (64 if c == "BN254" else 96) for c in curves}
pending = {}   # client id -> list of expected results, oldest first
bad = 0
t0 = time.time()


def collect(cl, key):
    global bad
    cl.wait_result()
    got = cl.result().result
    exp, what = pending[key].pop(0)
    if got != exp:
        bad += 1
        print("MISMATCH", what, flush=True)


for it in range(iters):
    c = rng.choice(curves)
    kind = rng.choice(("plain", "table", "table", "range", "range", "rewrite", "table+range", "dma_host", "dma_host", "hbm_host", "hbm_host",
                       "dma_host+range", "pieces", "pc_plan", "pc_plan", "pc_plan_host", "pc_exact", "pc_rewrite"))
    pc = rng.choice(("", "", "1", "2", "3", "8", "16"))
    if pc:
        os.environ["BLAZE_MSM_PIECES"] = pc
    else:
        os.environ.pop("BLAZE_MSM_PIECES", None)
    n = rng.choice([1, 63, 4096, 100001, 1 << 18, (1 << 19) + 5, 1 << 20, NMAX // 2 + 12, NMAX])
    first = 0 if n == NMAX else rng.randrange(0, NMAX - n) & ~3
    dp, ds = dev[c]
    if kind == "rewrite":
        # rewrite a span of the arena with the bytes it already holds: the table goes, the results must not change
        cl = table[c]
        key = ("t", c)
        while pending.get(key):
            collect(cl, key)
        m = rng.choice([1, 1000, 1 << 16])
        at = rng.randrange(0, NMAX - m)
        cl.load_data_to_hbm(View(dp, at * ps[c], m * ps[c]), ARENA[c], at * ps[c])
        continue
    if kind == "pc_rewrite":
        # rewrite a span of a precompute table with the bytes it already holds: the check's answer goes, the results must not change
        for cl, key in ((pc_plan[c], ("pp", c)), (pc_exact[c], ("pe", c))):
            while pending.get(key):
                collect(cl, key)
        m = rng.choice([1, 9, 1 << 13])
        at = rng.randrange(0, N8 * 8 - m)
        pc_plan[c].load_data_to_hbm(View(dev8[c], at * ps[c], m * ps[c]), ARENA8[c], at * ps[c])
        continue
    lo, hi = 0, 256
    if kind in ("pc_plan", "pc_plan_host", "pc_exact"):
        n = min(n, N8)
        first = 0 if n == N8 else rng.randrange(0, N8 - n)
        cl, key = (pc_exact[c], ("pe", c)) if kind == "pc_exact" else (pc_plan[c], ("pp", c))
        params = MSMParams(n, (ARENA8[c], first * 8 * ps[c]))
        sc_in = host_sc[c][first:first + n].tobytes() if kind == "pc_plan_host" else View(ds, first * 32, n * 32)
        inp = MSMInput(None, sc_in, params)
    elif kind == "hbm_host":
        # the reference's HBM flow: bases in the arena, the scalars a host buffer (an idle handle enqueues it piece by piece)
        cl, key = hbm[c], ("h", c)
        params = MSMParams(n, (ARENA[c], first * ps[c]))
        inp = MSMInput(None, host_sc[c][first:first + n].tobytes(), params)
    elif kind in ("dma_host", "dma_host+range"):
        cl, key = plain[c], ("p", c)
        if kind == "dma_host+range":
            a, b = sorted(rng.sample(range(0, 9), 2))
            lo, hi = 32 * a, 32 * b
        cl.set_scalar_range(lo, hi)
        params = MSMParams(n, None)
        inp = MSMInput(host_pts[c][first * ps[c]:(first + n) * ps[c]], host_sc[c][first:first + n].tobytes(), params)
    elif kind in ("table", "table+range"):
        cl, key = table[c], ("t", c)
        if kind == "table+range":
            a, b = sorted(rng.sample(range(0, 9), 2))
            lo, hi = 32 * a, 32 * b
        cl.set_scalar_range(lo, hi)
        params = MSMParams(n, (ARENA[c], first * ps[c]))
        inp = MSMInput(None, View(ds, first * 32, n * 32), params)
    else:
        cl, key = plain[c], ("p", c)
        if kind == "range":
            a, b = sorted(rng.sample(range(0, 9), 2))
            lo, hi = 32 * a, 32 * b
        cl.set_scalar_range(lo, hi)
        params = MSMParams(n, None)
        inp = MSMInput(View(dp, first * ps[c], n * ps[c]), View(ds, first * 32, n * 32), params)
    pending.setdefault(key, [])
    if len(pending[key]) >= 2:
        collect(cl, key)
    cl.initialize(params); cl.start_process(); cl.set_data(inp)
    pending[key].append((expected(c, first, n, lo, hi), f"it={it} {kind} {c} n={n} first={first} bits=[{lo},{hi})"))
    if it % 20 == 19:
        print(f"iter {it + 1}: {time.time() - t0:.1f} s, mismatches so far {bad}", flush=True)
for (k, c), lst in pending.items():
    cl = table[c] if k == "t" else hbm[c] if k == "h" else pc_plan[c] if k == "pp" else pc_exact[c] if k == "pe" else plain[c]
    while lst:
        collect(cl, (k, c))
used = {c: pc_plan[c].precompute_plan_info() for c in curves}
print("precompute plan, last task per curve:", used)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
