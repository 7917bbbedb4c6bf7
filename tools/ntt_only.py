#!/usr/bin/env python3
"""Dev tool: run a few 2^logn NTTs (for profiling)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blaze_amd
from blaze_amd import DeviceBuffer
from blaze_amd._lib import check
from blaze_amd.driver_client import DriverClient
from blaze_amd.ingo_ntt import NTT, NTTClient, NTTInput, NttInit
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 27
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 1 << logn
d = DeviceBuffer(0, 32 * n)
check(blaze_amd.aux().blz_synth_field_elements(0, d.ptr, n, 5))
nc = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn)
nc.set_data(NTTInput(0, d))
for i in range(reps):
    nc.initialize(NttInit()); nc.start_process(0); nc.wait_result()
    print("kernel ms", nc.last_kernel_ms())
