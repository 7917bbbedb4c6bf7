#!/usr/bin/env python3
"""Dev tool: random CALL SEQUENCES against the C ABI - valid and invalid ones mixed (wrong call order, wrong lengths, unloaded arena
ranges, queue overflow, option flips in mid-stream) - on long-lived clients of every kind.  Every call either succeeds or fails with
one of the reference's error variants; nothing may crash, hang or leave a client in a state reset() does not clear: after every
burst each client is reset and must return the right bytes for a known task (P_i = (i + 1) G: linearity).
    python3 tests/probes/api_monkey.py [bursts] [seed]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np  # noqa: E402

import blaze_amd  # noqa: E402
import oracle  # noqa: E402
from blaze_amd._lib import DriverClientError, check  # noqa: E402
from blaze_amd.ingo_msm import MSMInput, MSMParams, PointMemoryType  # noqa: E402
from gpu_util import msm_client, synth  # noqa: E402

bursts = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
NMAX = 1 << 14
curves = ("BLS381", "BLS377", "BN254")
L = blaze_amd.lib()
check(L.blz_arena_release(0))
ps = {c: (64 if c == "BN254" else 96) for c in curves}
host, dev, dev8 = {}, {}, {}
ARENA = {c: (i + 1) << 32 for i, c in enumerate(curves)}
ARENA8 = {c: (i + 9) << 32 for i, c in enumerate(curves)}
for c in curves:
    dp, ds = synth(c, NMAX, seed=5)
    dev[c] = (dp, ds)
    host[c] = (bytes(dp.download()), np.frombuffer(ds.download(), dtype=np.uint8).reshape(NMAX, 32).copy())
    dp8, _ = synth(c, NMAX // 8, pf=8, seed=5)
    dev8[c] = dp8
clients = {}
for c in curves:
    clients[("dma", c)] = msm_client(c, 1)
    clients[("hbm", c)] = msm_client(c, 1, PointMemoryType.HBM)
    clients[("pc", c)] = msm_client(c, 8, PointMemoryType.HBM)
    clients[("hbm", c)].load_data_to_hbm(dev[c][0], ARENA[c], 0)
    clients[("pc", c)].load_data_to_hbm(dev8[c], ARENA8[c], 0)


def expected(c, n):
    k = oracle.index_weighted_sum(c, host[c][1][:n].tobytes(), n, 0, threads=4)
    return oracle.result_from_affine(c, oracle.generator_mul(c, k))


counts = {"ok": 0}
known = {}


def attempt(f):
    try:
        f()
        counts["ok"] += 1
    except DriverClientError as e:
        counts[e.variant] = counts.get(e.variant, 0) + 1


bad = 0
for b in range(bursts):
    for _ in range(60):
        kind, c = rng.choice(list(clients))
        cl = clients[(kind, c)]
        n = rng.choice([0, 1, 2, 255, 256, 1000, 4096, NMAX // 8, NMAX])
        pf = 8 if kind == "pc" else 1
        addr = (ARENA8 if kind == "pc" else ARENA)[c]
        off = rng.choice([0, 0, 0, ps[c] * 8 * rng.randrange(0, 64), 1 << 40, 7])
        hbm = None if kind == "dma" and rng.random() < 0.8 else (addr, off)
        op = rng.choice(("initialize", "start", "set_data", "set_data", "set_data_dev", "wait", "result", "load", "get", "range", "table", "plan",
                         "prepare_table", "prepare_plan", "info", "reset", "stream_task"))
        m = n if rng.random() < 0.8 else rng.choice([0, 1, 77, NMAX])   # (a set_data that does not match the initialize before it)
        pts_len = m * pf * ps[c]
        if op == "initialize":
            attempt(lambda: cl.initialize(MSMParams(n, hbm)))
        elif op == "start":
            attempt(cl.start_process)
        elif op == "set_data":
            pts = None if (hbm and rng.random() < 0.8) else host[c][0][: min(pts_len, NMAX * ps[c])]
            sc = host[c][1][: min(m, NMAX)].tobytes()
            if rng.random() < 0.1:
                sc = sc[:-5]
            attempt(lambda: cl.set_data(MSMInput(pts, sc, MSMParams(m, hbm))))
        elif op == "set_data_dev":
            attempt(lambda: cl.set_data(MSMInput(None if hbm else dev[c][0], dev[c][1], MSMParams(NMAX, hbm))))
        elif op == "wait":
            attempt(cl.wait_result)
        elif op == "result":
            attempt(cl.result)
        elif op == "load":
            ln = rng.choice([0, 1, ps[c], 1000 * ps[c], NMAX * ps[c]])
            attempt(lambda: cl.load_data_to_hbm(host[c][0][:ln], addr, off if off < (1 << 30) else 0))
        elif op == "get":
            attempt(lambda: cl.get_data_from_hbm(rng.choice([0, 1, 96, 5000]), addr, off))
        elif op == "range":
            a, z = sorted(rng.sample(range(0, 9), 2))
            attempt(lambda: cl.set_scalar_range(32 * a + rng.choice([0, 0, 0, 3]), 32 * z))
        elif op == "table":
            attempt(lambda: cl.set_window_table(rng.choice([0, 1, 2, 2, 5])))
        elif op == "plan":
            attempt(lambda: cl.set_precompute_plan(rng.choice([0, 1, 1])))
        elif op == "prepare_table":
            attempt(lambda: cl.prepare_window_table(n, (addr, off), rng.choice([0, 50])))
        elif op == "prepare_plan":
            attempt(lambda: cl.prepare_precompute_plan(n, (addr, off)))
        elif op == "info":
            attempt(lambda: (cl.get_api(), cl.memory_info(), cl.window_table_info(), cl.precompute_plan_info(), cl.is_msm_engine_ready()))
        elif op == "stream_task":
            # a whole task fed by several set_data calls (blaze_hip.h "STREAMED TASKS") in the middle of whatever state the client is
            # in - behind a reset, so that its result can be checked; DMA-mode clients only (the bursts rewrite the arenas' bases)
            if kind != "dma":
                continue
            cl.reset()
            cl.set_scalar_range(0, 0)
            ns = rng.choice([1, 2, 255, 1000, 4096, NMAX])
            cl.initialize(MSMParams(ns, None)); cl.start_process()
            at = 0
            while at < ns:
                mm = min(ns - at, rng.choice([0, 1, 7, 100, 2048, 4097, ns]))
                cl.set_data(MSMInput(host[c][0][at * ps[c]: (at + mm) * ps[c]], host[c][1][at: at + mm].tobytes(), MSMParams(mm, None)))
                at += mm
                if at < ns and rng.random() < 0.1:     # calls a half-fed task refuses, or ignores
                    for f in (cl.start_process, cl.wait_result, lambda: cl.set_data(MSMInput(None, b"", MSMParams(0, None))),
                              lambda: cl.set_data(MSMInput(host[c][0][: ps[c] * (ns - at + 1)], host[c][1][: ns - at + 1].tobytes(), MSMParams(ns - at + 1, None)))):
                        attempt(f)
                    assert cl.stream_progress() == (at, ns), (cl.stream_progress(), at, ns)
            cl.wait_result()
            got = cl.result().result
            counts["streamed"] = counts.get("streamed", 0) + 1
            if (c, ns) not in known:
                known[(c, ns)] = expected(c, ns)
            if got != known[(c, ns)]:
                bad += 1
                print("MISMATCH streamed task", b, c, ns, flush=True)
        else:
            attempt(cl.reset)
    # every client back to a known state, then a known task
    for (kind, c), cl in clients.items():
        cl.reset()
        cl.set_scalar_range(0, 0)
        n = NMAX // 8 if kind == "pc" else rng.choice([257, 4096, NMAX])
        if kind == "hbm":
            cl.load_data_to_hbm(dev[c][0], ARENA[c], 0)       # (the burst may have rewritten parts of it)
            inp, prm = MSMInput(None, host[c][1][:n].tobytes(), MSMParams(n, (ARENA[c], 0))), MSMParams(n, (ARENA[c], 0))
        elif kind == "pc":
            cl.load_data_to_hbm(dev8[c], ARENA8[c], 0)
            inp, prm = MSMInput(None, host[c][1][:n].tobytes(), MSMParams(n, (ARENA8[c], 0))), MSMParams(n, (ARENA8[c], 0))
        else:
            inp, prm = MSMInput(host[c][0][: n * ps[c]], host[c][1][:n].tobytes(), MSMParams(n, None)), MSMParams(n, None)
        cl.initialize(prm); cl.start_process(); cl.set_data(inp); cl.wait_result()
        got = cl.result().result
        if (c, n) not in known:
            known[(c, n)] = expected(c, n)
        if got != known[(c, n)]:
            bad += 1
            print("MISMATCH after burst", b, kind, c, n, flush=True)
    if b % 10 == 9:
        print(f"burst {b + 1}: calls by outcome {counts}, mismatches {bad}", flush=True)
print("calls by outcome:", counts)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
