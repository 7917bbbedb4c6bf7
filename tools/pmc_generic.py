#!/usr/bin/env python3
"""Average value per (kernel, counter) from a rocprofv3 --pmc rocpd database.
    python tools/pmc_generic.py run_results.db [kernel-substring]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = {}
for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
    if flt in k:
        a = agg.setdefault((k, c), [0, 0.0])
        a[0] += 1
        a[1] += v
last = None
for (k, c), (n, s) in sorted(agg.items()):
    if k != last:
        print(k[:100])
        last = k
    print(f"    {c:<28} launches {n:>4}  avg {s / n:>18.1f}")
