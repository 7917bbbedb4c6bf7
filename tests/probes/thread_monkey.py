#!/usr/bin/env python3
"""Check campaign: several HOST THREADS at once on one device, each with clients of its own (DMA mode, arena bases, window table,
precompute plan) over its own address range of the one shared arena - random tasks, rewrites of the bases, read-backs, table and plan
preparations, policy flips (the arena's diet), client close / reopen - every result checked through linearity.  What is shared is the
arena's book-keeping, the conversion ordering across handles, the window planner's memo and the device itself.
    python3 tests/probes/thread_monkey.py [seconds] [threads] [seed]"""
import os
import random
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np  # noqa: E402

import blaze_amd  # noqa: E402
import oracle  # noqa: E402
from blaze_amd._lib import DriverClientError, check  # noqa: E402
from blaze_amd.ingo_msm import MSMInput, MSMParams, PointMemoryType  # noqa: E402
from gpu_util import msm_client, synth  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
L = blaze_amd.lib()
check(L.blz_arena_release(0))
NMAX = 1 << 18
curves = ("BLS381", "BLS377", "BN254")
ps = {c: (64 if c == "BN254" else 96) for c in curves}
host, host8, known = {}, {}, {}
for c in curves:
    dp, ds = synth(c, NMAX, seed=5)
    host[c] = (bytes(dp.download()), np.frombuffer(ds.download(), dtype=np.uint8).reshape(NMAX, 32).copy())
    dp.free(); ds.free()
    dp8, ds8 = synth(c, NMAX // 8, pf=8, seed=5)
    host8[c] = bytes(dp8.download())
    dp8.free(); ds8.free()
    for n in (1000, 1 << 14, NMAX // 8, NMAX):
        k = oracle.index_weighted_sum(c, host[c][1][:n].tobytes(), n, 0, threads=8)
        known[(c, n)] = oracle.result_from_affine(c, oracle.generator_mul(c, k))
problems, counts = [], [0] * nthreads
stop = time.time() + seconds


def worker(t):
    rng = random.Random(seed * 100 + t)
    c = curves[t % 3]
    base, base8 = (t + 1) << 36, ((t + 1) << 36) + (1 << 35)
    try:
        while time.time() < stop:
            kind = rng.choice(("dma", "hbm", "table", "pc"))
            pf = 8 if kind == "pc" else 1
            cl = msm_client(c, pf, PointMemoryType.DMA if kind == "dma" else PointMemoryType.HBM)
            if kind == "table":
                cl.set_window_table(2)
            if kind == "pc":
                cl.set_precompute_plan(rng.random() < 0.7)
                cl.load_data_to_hbm(host8[c], base8, 0)
            elif kind != "dma":
                cl.load_data_to_hbm(host[c][0], base, 0)
            for _ in range(rng.randrange(2, 9)):
                n = NMAX // 8 if kind == "pc" else rng.choice((1000, 1 << 14, NMAX))
                op = rng.random()
                if op < 0.15 and kind != "dma":      # rewrite a span of the own bases with the bytes it holds
                    src, b0 = (host8[c], base8) if kind == "pc" else (host[c][0], base)
                    at = rng.randrange(0, len(src) // ps[c] - 64) * ps[c]
                    cl.load_data_to_hbm(src[at: at + rng.choice((1, 64)) * ps[c]], b0, at)
                elif op < 0.25 and kind != "dma":
                    src, b0 = (host8[c], base8) if kind == "pc" else (host[c][0], base)
                    at = rng.randrange(0, len(src) - 4096)
                    if bytes(cl.get_data_from_hbm(4096, b0, at)) != src[at: at + 4096]:
                        problems.append(f"thread {t}: read-back differs ({kind} {c} at {at})")
                elif op < 0.32 and kind == "table":
                    cl.prepare_window_table(NMAX, (base, 0), rng.choice((0, 0, -1)))
                elif op < 0.32 and kind == "pc":
                    cl.prepare_precompute_plan(n, (base8, 0))
                elif op < 0.36 and t == 0:
                    check(L.blz_arena_set_policy(0, rng.randrange(2)))
                else:
                    hbm = None if kind == "dma" else ((base8, 0) if kind == "pc" else (base, 0))
                    prm = MSMParams(n, hbm)
                    pts = host[c][0][: n * ps[c]] if kind == "dma" else None
                    two = rng.random() < 0.5
                    sliced = rng.random() < 0.35       # the task fed by several set_data calls (streamed tasks), any client kind
                    for _k in range(2 if two else 1):
                        cl.initialize(prm); cl.start_process()
                        if not sliced:
                            cl.set_data(MSMInput(pts, host[c][1][:n].tobytes(), prm))
                            continue
                        at = 0
                        while at < n:
                            m = min(n - at, rng.choice((1, 2048, n // 3 + 1, n)))
                            sp = None if pts is None else pts[at * ps[c]: (at + m) * ps[c]]
                            cl.set_data(MSMInput(sp, host[c][1][at: at + m].tobytes(), MSMParams(m, hbm)))
                            at += m
                    for _k in range(2 if two else 1):
                        cl.wait_result()
                        if cl.result().result != known[(c, n)]:
                            problems.append(f"thread {t}: MISMATCH {kind} {c} n={n}")
                        counts[t] += 1
            cl.close()
    except DriverClientError as e:
        problems.append(f"thread {t}: unexpected {e.variant}: {str(e)[:200]}")
    except Exception as e:   # noqa: BLE001
        problems.append(f"thread {t}: {type(e).__name__}: {str(e)[:200]}")


ths = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
for th in ths:
    th.start()
for th in ths:
    th.join()
check(L.blz_arena_set_policy(0, 0))
check(L.blz_arena_release(0))
print(f"{nthreads} threads, {seconds:.0f} s: tasks per thread {counts}")
for p in problems[:20]:
    print("PROBLEM:", p)
print("problems:", len(problems))
sys.exit(1 if problems else 0)
