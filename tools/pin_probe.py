#!/usr/bin/env python3
"""Dev tool: what pinning the host buffers would buy in DMA mode: hipHostRegister cost and host -> device rates
for pageable, registered and hipHostMalloc'd memory (4 GiB)."""
import ctypes as C, time, sys, os
hip = C.CDLL("libamdhip64.so")
n = 4 << 30
def chk(rc, what):
    if rc != 0: raise SystemExit(f"{what} failed: {rc}")
d = C.c_void_p(); chk(hip.hipMalloc(C.byref(d), C.c_size_t(n)), "hipMalloc")
buf = bytearray(n)
p = C.addressof((C.c_char * n).from_buffer(buf))
C.memset(p, 1, n)
def h2d(ptr, label):
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); chk(hip.hipMemcpy(d, C.c_void_p(ptr), C.c_size_t(n), 1), "hipMemcpy"); best = min(best, time.perf_counter() - t)
    print(f"{label}: {n / best / 1e9:.1f} GB/s ({best * 1e3:.1f} ms for 4 GiB)", flush=True)
h2d(p, "pageable")
t = time.perf_counter(); chk(hip.hipHostRegister(C.c_void_p(p), C.c_size_t(n), 0), "hipHostRegister"); tr = time.perf_counter() - t
print(f"hipHostRegister(4 GiB): {tr * 1e3:.1f} ms", flush=True)
h2d(p, "registered")
t = time.perf_counter(); chk(hip.hipHostUnregister(C.c_void_p(p)), "hipHostUnregister"); print(f"hipHostUnregister: {(time.perf_counter() - t) * 1e3:.1f} ms")
hp = C.c_void_p(); chk(hip.hipHostMalloc(C.byref(hp), C.c_size_t(n), 0), "hipHostMalloc")
C.memset(hp, 2, n)
h2d(hp.value, "hipHostMalloc")
