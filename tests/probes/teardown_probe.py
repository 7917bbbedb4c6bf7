#!/usr/bin/env python3
"""Dev tool / test body: clients torn down with work in flight - close() with two tasks queued (device inputs, host inputs, arena
bases, window table, host scalars), blz_arena_release and a rewrite of the bases under tasks in flight, NTT close() / reset() under a
transform.  Nothing may crash or hang, tasks that were accepted return the right bytes.   python3 tests/probes/teardown_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import blaze_amd, oracle
from blaze_amd import DeviceBuffer, DriverClientError
from blaze_amd.driver_client import DriverClient
from blaze_amd.ingo_msm import MSMInput, MSMParams, PointMemoryType
from blaze_amd.ingo_ntt import NTT, NTTClient, NTTInput
from gpu_util import msm_client, run_msm, synth
L = blaze_amd.lib()
curve, n = "BLS381", 1 << 22
dp, ds = synth(curve, n)
k = oracle.index_weighted_sum(curve, ds.download(), n, 0, threads=8)
exp = oracle.result_from_affine(curve, oracle.generator_mul(curve, k))
hs = bytes(ds.download()); hp = bytes(dp.download())
def submit(cl, prm, pts, sc):
    cl.initialize(prm); cl.start_process(); cl.set_data(MSMInput(pts, sc, prm))
# 1. close with two tasks in flight (device inputs / host inputs / arena)
for kind in ("dev", "host", "hbm", "hbm_table", "hbm_host"):
    blaze_amd._lib.check(L.blz_arena_release(0))
    hbm = kind.startswith("hbm")
    cl = msm_client(curve, 1, PointMemoryType.HBM if hbm else PointMemoryType.DMA)
    if kind == "hbm_table": cl.set_window_table(2)
    prm = MSMParams(n, (0, 0) if hbm else None)
    if hbm: cl.load_data_to_hbm(dp, 0, 0)
    pts = None if hbm else (dp if kind == "dev" else hp)
    sc = hs if kind in ("host", "hbm_host") else ds
    submit(cl, prm, pts, sc); submit(cl, prm, pts, sc)
    cl.close()
    print(kind, "closed with two tasks in flight", flush=True)
# 2. arena release under tasks in flight, then collect them
blaze_amd._lib.check(L.blz_arena_release(0))
cl = msm_client(curve, 1, PointMemoryType.HBM)
cl.load_data_to_hbm(dp, 0, 0)
prm = MSMParams(n, (0, 0))
submit(cl, prm, None, ds); submit(cl, prm, None, ds)
blaze_amd._lib.check(L.blz_arena_release(0))
cl.wait_result(); r1 = cl.result().result; cl.wait_result(); r2 = cl.result().result
print("arena released under two tasks: results", r1 == exp, r2 == exp, flush=True)
try:
    submit(cl, prm, None, ds); print("PROBLEM: task over a released arena accepted")
except DriverClientError as e:
    print("task over the released arena:", e.variant)
cl.reset(); cl.load_data_to_hbm(dp, 0, 0)
print("after a new load:", run_msm(cl, None, ds, n, hbm=(0, 0)) == exp, flush=True)
# 3. rewrite the bases under tasks in flight (same bytes): results stay right
submit(cl, prm, None, ds); submit(cl, prm, None, ds)
cl2 = msm_client(curve, 1, PointMemoryType.HBM)
cl2.load_data_to_hbm(dp, 0, 0)
cl.wait_result(); r1 = cl.result().result; cl.wait_result(); r2 = cl.result().result
print("bases rewritten (same bytes) by another client under two tasks:", r1 == exp, r2 == exp, flush=True)
cl.close(); cl2.close()
# 4. NTT: close with a transform in flight; reset with one in flight
lg = 24
nt = NTTClient(NTT.Ntt, DriverClient(0), lg); nt.initialize()
v = os.urandom(32 << lg)
v = bytes(b & 0x3f if (i & 31) == 31 else b for i, b in enumerate(v[:32 << 10])) * (1 << (lg - 10))
nt.set_data(NTTInput(0, v)); nt.start_process(0); nt.close()
print("ntt closed with a transform in flight", flush=True)
nt = NTTClient(NTT.Ntt, DriverClient(0), lg); nt.initialize()
nt.set_data(NTTInput(0, v)); nt.start_process(0); nt.reset()
nt.set_data(NTTInput(1, v)); nt.start_process(1); nt.wait_result(); a = nt.result(1)
nt.set_data(NTTInput(0, v)); nt.start_process(0); nt.wait_result(); b = nt.result(0)
print("ntt reset with a transform in flight, then two transforms agree:", a == b, flush=True)
nt.close()
blaze_amd._lib.check(L.blz_arena_release(0))
print("done")
