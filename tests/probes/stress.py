#!/usr/bin/env python3
"""Dev tool: many MSMs of varying size / curve on long-lived clients (two in flight), results checked
through linearity; prints device memory in use at intervals to spot leaks."""
import sys, os, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import ctypes as C
import oracle
import blaze_amd
from blaze_amd.ingo_msm import *
from gpu_util import msm_client, synth

class View(blaze_amd.DeviceBuffer):
    """non-owning prefix of a DeviceBuffer"""
    def __init__(self, base, nbytes):
        self.device_id, self.nbytes, self.ptr = base.device_id, nbytes, base.ptr
    def free(self):
        pass

hip = C.CDLL("libamdhip64.so")
def mem_used():
    free, tot = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(free), C.byref(tot))
    return (tot.value - free.value) / 2**30

rng = random.Random(5)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
clients = {c: msm_client(c, 1) for c in ("BLS381", "BLS377", "BN254")}
bufs = {}
for c in clients:
    bufs[c] = synth(c, 1 << 22, seed=11)
pending = []
t0 = time.time()
bad = 0
for it in range(iters):
    c = rng.choice(list(clients))
    n = rng.choice([1, 77, 4096, 100000, 1 << 18, (1 << 20) + 3, 1 << 22])
    cl = clients[c]
    dp, ds = bufs[c]
    ps = 64 if c == "BN254" else 96
    p = MSMParams(n, None)
    # sub-views of the big buffers: the first n elements
    pv, sv = View(dp, n * ps), View(ds, n * 32)
    if cl.is_msm_engine_ready() == 0:
        cl.wait_result(); cl.result()
    cl.initialize(p); cl.start_process(); cl.set_data(MSMInput(pv, sv, p))
    if it % 5 == 4:
        got = None
        while True:   # drain: every task in flight on this client, oldest first
            try:
                cl.wait_result()
                got = cl.result().result
            except blaze_amd.DriverClientError:
                break
        # the last popped result belongs to the task just submitted: check it
        sc = ds.download(n * 32)
        k = oracle.index_weighted_sum(c, sc, n, 0)
        exp = oracle.result_from_affine(c, oracle.generator_mul(c, k))
        if got != exp:
            bad += 1
            print("MISMATCH", it, c, n)
    if it % 10 == 9:
        print(f"iter {it+1}: {time.time()-t0:.1f}s, device memory in use {mem_used():.2f} GiB", flush=True)
print("mismatches:", bad)
