#!/usr/bin/env python3
"""Dev tool: every transform size 2^1 .. 2^26 (forward and inverse, BLS12-381 Fr; the other two fields at the sizes where the
pass geometry changes) against the threaded CPU oracle, every output compared.   python3 tests/probes/ntt_sizes_probe.py [max_log]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

import oracle  # noqa: E402
from blaze_amd.driver_client import DriverClient  # noqa: E402
from blaze_amd.ingo_ntt import NTT, NTTClient, NTTInput  # noqa: E402

max_log = int(sys.argv[1]) if len(sys.argv) > 1 else 26
dc = DriverClient(0)
bad = 0
for lg in range(1, max_log + 1):
    x = np.random.default_rng(lg).integers(0, 256, size=32 << lg, dtype=np.uint8)
    x[31::32] &= 0x0F
    xb = x.tobytes()
    for field, inv in [("BLS381", False), ("BLS381", True)] + ([("BLS377", False), ("BN254", True)] if lg in (8, 9, 10, 18, 19, 23, 24, 25, 26) else []):
        t0 = time.perf_counter()
        cl = NTTClient(NTT.Ntt, dc, lg, inv, field)
        cl.initialize()
        cl.set_data(NTTInput(lg & 1, xb)); cl.start_process(lg & 1); cl.wait_result()
        got = bytes(cl.result(lg & 1))
        info = cl.info()
        cl.close()
        want = bytes(oracle.ntt(field, xb, lg, inv, threads=16))
        ok = got == want
        bad += not ok
        print(f"2^{lg} {field} {'inverse' if inv else 'forward'}: {'ok' if ok else 'MISMATCH'} ({time.perf_counter() - t0:.1f} s; pass-2 table {info['pass2_factor_table']}, "
              f"pass-1 table {info['pass1_boundary_table']})", flush=True)
    # ... and one caller-chosen convention per size (blz_ntt_new_ex3): a random odd power of the default root, random direction
    # and orders, in a field of the size's turn - the address arithmetic of the folded bit reversal meets every pass geometry
    import random
    from oracle import pyref
    rng = random.Random(1000 + lg)
    field = ("BLS381", "BLS377", "BN254")[lg % 3]
    r = pyref.CURVES[field]["r"]
    root = pow(oracle.omega(field, lg), 2 * rng.randrange(1 << 20) + 1, r)
    inv, brin, brout = rng.random() < 0.5, rng.random() < 0.6, rng.random() < 0.6
    flags = (NTTClient.INVERSE if inv else 0) | (NTTClient.BITREV_INPUT if brin else 0) | (NTTClient.BITREV_OUTPUT if brout else 0)
    t0 = time.perf_counter()
    cl = NTTClient(NTT.Ntt, dc, lg, field=field, flags=flags, root=root)
    cl.initialize()
    cl.set_data(NTTInput(0, xb)); cl.start_process(0); cl.wait_result()
    got = bytes(cl.result(0))
    cl.close()
    ok = got == bytes(oracle.ntt(field, xb, lg, inv, threads=16, root=root, bitrev_in=brin, bitrev_out=brout))
    bad += not ok
    print(f"2^{lg} {field} caller's root, inverse={inv} bitrev_in={brin} bitrev_out={brout}: {'ok' if ok else 'MISMATCH'} ({time.perf_counter() - t0:.1f} s)", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
