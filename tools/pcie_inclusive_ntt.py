#!/usr/bin/env python3
"""Config 5 of BASELINE.json with host buffers: the 2^27 NTT through set_data / result with 4 GiB host vectors, timed
like benches/ntt_bench.rs (one transform: set_data, initialize, start_process, wait_result, result) and as the
reference's double-buffered loop (tests/integration_ntt.rs:102-136: start the kernel on one buffer, read the other
buffer's result, write the next input into it, wait).  The PCIe-inclusive rate beside the HBM-resident latency
bench.py reports."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blaze_amd.driver_client import DriverClient
from blaze_amd.ingo_ntt import NTT, NTTClient, NTTInput, NttInit

if os.environ.get("IMPORT_TORCH") == "1":   # (does the host loop care whether torch owns the process's HIP context first?  bench.py imports it)
    import torch
    torch.cuda.synchronize(0)
    if os.environ.get("TORCH_BIG") == "1":
        _t = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); _h = _t.cpu(); del _t, _h
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 27
n = 1 << logn
x = np.random.default_rng(1).integers(0, 256, size=32 * n, dtype=np.uint8)
x[31::32] &= 0x3F
nc = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn)
if os.environ.get("ONLY_PINNED") == "1":   # (experiment runs: just the exchange loop over page-locked buffers)
    import blaze_amd
    hx, hy = blaze_amd.HostBuffer(0, 32 * n), blaze_amd.HostBuffer(0, 32 * n)
    px, py = hx.array(), hy.array()
    px[:] = x
    py[:] = 1
    nc.initialize(NttInit())
    ts = []
    for i in range(8):
        t = time.perf_counter()
        nc.start_process(1 - i % 2)
        nc.exchange(i % 2, px, py)
        nc.wait_result()
        ts.append(round((time.perf_counter() - t) * 1e3, 1))
    print(json.dumps({"exchange_cycles_ms_pinned": ts}))
    sys.exit(0)
y = np.zeros(32 * n, dtype=np.uint8)   # the host's output vector, kept between transforms (pages touched once)
y[:] = 1
t = time.perf_counter(); fresh = nc.result(0); fresh_ms = (time.perf_counter() - t) * 1e3   # a fresh 4 GiB allocation per call
del fresh
rows = []
for rep in range(4):
    t0 = time.perf_counter(); nc.set_data(NTTInput(0, x))
    t1 = time.perf_counter(); nc.initialize(NttInit()); nc.start_process(0); nc.wait_result()
    t2 = time.perf_counter(); nc.result_into(0, y)
    t3 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0, nc.last_kernel_ms()))
best = min(rows[1:], key=lambda r: r[3])
# the reference's double-buffered loop
cycles = 6
nc.initialize(NttInit())
per = []
for i in range(cycles + 2):
    t = time.perf_counter()
    bh, bk = i % 2, 1 - i % 2
    nc.start_process(bk)
    nc.result_into(bh, y)
    nc.set_data(NTTInput(bh, x))
    nc.wait_result()
    per.append(time.perf_counter() - t)
steady = sorted(per[2:])[len(per[2:]) // 2]


# the same loop with result + set_data fused into blz_ntt_exchange (full duplex): pageable buffers, then page-locked ones
def exchange_loop(xin, yout):
    nc.initialize(NttInit())
    ts = []
    for i in range(cycles + 2):
        t = time.perf_counter()
        bh, bk = i % 2, 1 - i % 2
        nc.start_process(bk)
        nc.exchange(bh, xin, yout)
        nc.wait_result()
        ts.append(time.perf_counter() - t)
    return ts


per_x = exchange_loop(x, y)
steady_x = sorted(per_x[2:])[len(per_x[2:]) // 2]
import blaze_amd
hx, hy = blaze_amd.HostBuffer(0, 32 * n), blaze_amd.HostBuffer(0, 32 * n)
px, py = hx.array(), hy.array()
px[:] = x
py[:] = 1
per_p = exchange_loop(px, py)
steady_p = sorted(per_p[2:])[len(per_p[2:]) // 2]
# (and the two plain calls with page-locked buffers, for reference)
t = time.perf_counter(); nc.set_data(NTTInput(0, px)); sd_p = time.perf_counter() - t
t = time.perf_counter(); nc.result_into(0, py); rs_p = time.perf_counter() - t
out = {"config": f"2^{logn} NTT BLS12-381 Fr, host buffers (numpy, pageable)", "host_bytes_each_way": 32 * n,
       "set_data_ms": round(best[0] * 1e3, 1), "h2d_GBps": round(32 * n / best[0] / 1e9, 1),
       "start_to_wait_ms": round(best[1] * 1e3, 2), "kernel_ms": round(best[4], 2),
       "result_ms": round(best[2] * 1e3, 1), "d2h_GBps": round(32 * n / best[2] / 1e9, 1),
       "result_into_fresh_allocation_ms": round(fresh_ms, 1),
       "one_transform_full_ms": round(best[3] * 1e3, 1),
       "double_buffered_loop_ms_per_transform": round(steady * 1e3, 1),
       "double_buffered_cycles_ms": [round(p * 1e3, 1) for p in per],
       "exchange_loop_ms_per_transform_pageable": round(steady_x * 1e3, 1), "exchange_cycles_ms_pageable": [round(p * 1e3, 1) for p in per_x],
       "exchange_loop_ms_per_transform_pinned": round(steady_p * 1e3, 1), "exchange_cycles_ms_pinned": [round(p * 1e3, 1) for p in per_p],
       "set_data_ms_pinned": round(sd_p * 1e3, 1), "result_ms_pinned": round(rs_p * 1e3, 1)}
print(json.dumps(out))
