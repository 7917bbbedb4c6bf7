#!/usr/bin/env python3
"""Dev tool: repeated NTTs of random sizes / fields / directions on long-lived clients against the oracle, plus
repeated 2^27 transforms of the same input (every repetition must give the same bytes: a race in the tile exchange
would show up as a run-to-run difference)."""
import hashlib, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
import blaze_amd
from blaze_amd import DeviceBuffer
from blaze_amd._lib import check
from blaze_amd.driver_client import DriverClient
from blaze_amd.ingo_ntt import NTT, NTTClient, NTTInput, NttInit
from oracle import pyref

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
reps27 = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = random.Random(31)
bad = 0
clients = {}
t0 = time.time()
for it in range(iters):
    field = rng.choice(["BLS381", "BLS377", "BN254"])
    logn = rng.choice([1, 4, 9, 10, 14, 17, 18, 19, 21, 22])
    inv = rng.random() < 0.3
    key = (field, logn, inv)
    if key not in clients:
        clients[key] = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, inverse=inv, field=field)
    cl = clients[key]
    r = pyref.CURVES[field]["r"]
    n = 1 << logn
    g = np.random.default_rng(rng.randrange(1 << 30))
    x = g.integers(0, 256, size=32 * n, dtype=np.uint8)
    x[31::32] &= 0x0F
    buf = rng.randrange(2)
    cl.set_data(NTTInput(buf, x)); cl.initialize(NttInit()); cl.start_process(buf); cl.wait_result()
    got = bytes(cl.result(buf))
    exp_ok = got == bytes(oracle.ntt(field, x.tobytes(), logn, inverse=inv, threads=16))
    bad += not exp_ok
    print(it, field, logn, "inv" if inv else "fwd", "OK" if exp_ok else "MISMATCH", flush=True)
for cl in clients.values():
    cl.close()
# 2^27: same input, many runs, one digest
n = 1 << 27
d = DeviceBuffer(0, 32 * n)
check(blaze_amd.aux().blz_synth_field_elements(0, d.ptr, n, 77))
nc = NTTClient(NTT.Ntt, DriverClient(0), log_size=27)
digests = set()
for i in range(reps27):
    nc.set_data(NTTInput(0, d)); nc.initialize(NttInit()); nc.start_process(0); nc.wait_result()
    digests.add(hashlib.blake2b(nc.result(0), digest_size=16).hexdigest())
print("2^27 digests over", reps27, "runs:", digests)
bad += len(digests) != 1
print("mismatches", bad, "in", round(time.time() - t0, 1), "s")
sys.exit(1 if bad else 0)
