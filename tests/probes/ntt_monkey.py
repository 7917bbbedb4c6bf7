#!/usr/bin/env python3
"""Dev tool: random call sequences against the NTT side of the C ABI - the reference's double-buffer cycle cut up at random: wrong
buffer numbers, wrong lengths, results read from the buffer under transform, start_process twice, exchange on either buffer - on
clients of several sizes, fields and directions.  Every call succeeds or fails with one of the reference's error variants; after every
burst each client is reset and must transform a known vector into the oracle's bytes.
    python3 tests/probes/ntt_monkey.py [bursts] [seed]"""
import ctypes as C
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import blaze_amd  # noqa: E402
import oracle  # noqa: E402
from blaze_amd import DeviceBuffer  # noqa: E402
from blaze_amd._lib import DriverClientError, buf_ptr, lib  # noqa: E402
from blaze_amd.driver_client import DriverClient  # noqa: E402
from blaze_amd.ingo_ntt import NTT, NTTClient, NTTInput  # noqa: E402

bursts = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
dc = DriverClient(0)
shapes = [(4, "BLS381", False, 0), (10, "BLS381", False, 0), (13, "BLS381", True, 0), (12, "BLS377", False, 0), (11, "BN254", True, 0),
          (19, "BLS381", False, 1), (20, "BLS381", False, 0)]
clients, vec, want, dbuf = [], [], [], []
for lg, field, inv, flags in shapes:
    cl = NTTClient(NTT.Ntt, dc, lg, inv, field, flags)
    cl.initialize()
    clients.append(cl)
    r = random.Random(lg * 7 + len(field))
    v = b"".join(r.getrandbits(250).to_bytes(32, "little") for _ in range(1 << lg))
    vec.append(v)
    want.append(bytes(oracle.ntt(field, v, lg, inv, threads=8)))
    d = DeviceBuffer(0, len(v))
    d.upload(v)
    dbuf.append(d)
counts = {"ok": 0}


def attempt(f):
    try:
        f()
        counts["ok"] += 1
    except DriverClientError as e:
        counts[e.variant] = counts.get(e.variant, 0) + 1


bad = 0
for b in range(bursts):
    for _ in range(80):
        i = rng.randrange(len(clients))
        cl, v = clients[i], vec[i]
        buf = rng.choice([0, 0, 1, 1, 2, -1, 7])
        op = rng.choice(("init", "start", "start", "set", "set", "set_dev", "wait", "wait", "result", "result_raw", "exchange", "exchange_raw",
                         "result_dev", "info", "reset"))
        if op == "init":
            attempt(cl.initialize)
        elif op == "start":
            attempt(lambda: cl.start_process(buf))
        elif op == "set":
            data = v if rng.random() < 0.8 else v[: rng.choice([0, 32, len(v) - 32])]
            attempt(lambda: cl.set_data(NTTInput(buf, data)))
        elif op == "set_dev":
            attempt(lambda: cl.set_data(NTTInput(buf, dbuf[i] if rng.random() < 0.8 else dbuf[(i + 1) % len(dbuf)])))
        elif op == "wait":
            attempt(cl.wait_result)
        elif op == "result":
            attempt(lambda: cl.result(buf))
        elif op == "result_raw":   # a caller's buffer that is too small
            out = bytearray(rng.choice([0, 32, len(v) - 1]))
            p, n, _k = buf_ptr(out)
            attempt(lambda: blaze_amd._lib.check(lib().blz_ntt_result(cl._h, buf, p, n)))
        elif op == "exchange":
            out = bytearray(len(v))
            attempt(lambda: cl.exchange(buf, v, out))
        elif op == "exchange_raw":
            out = bytearray(len(v))
            po, no, _k2 = buf_ptr(out)
            pi, ni, _k1 = buf_ptr(v)
            attempt(lambda: blaze_amd._lib.check(lib().blz_ntt_exchange(cl._h, buf, pi, rng.choice([0, 32, ni]), po, rng.choice([0, no - 32, no]))))
        elif op == "result_dev":
            attempt(lambda: cl.result_device(buf, dbuf[i] if rng.random() < 0.5 else dbuf[0]))
            dbuf[i].upload(v)
        elif op == "info":
            attempt(cl.info)
        else:
            attempt(cl.reset)
    for i, cl in enumerate(clients):
        cl.reset()
        side = b & 1
        cl.set_data(NTTInput(side, vec[i]))
        cl.start_process(side)
        cl.wait_result()
        if bytes(cl.result(side)) != want[i]:
            bad += 1
            print("MISMATCH after burst", b, shapes[i], flush=True)
    if b % 10 == 9:
        print(f"burst {b + 1}: calls by outcome {counts}, mismatches {bad}", flush=True)
print("calls by outcome:", counts)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
