#!/usr/bin/env python3
"""Dev tool: per-kernel averages of every counter in rocprofv3 --pmc result databases.
    python tools/pmc_dump.py [substring] a_results.db b_results.db ..."""
import sqlite3
import sys


def main(argv):
    pat = argv[0] if argv and not argv[0].endswith(".db") else ""
    dbs = [a for a in argv if a.endswith(".db")]
    rows = {}
    for path in dbs:
        cur = sqlite3.connect(path).cursor()
        for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
            if pat in k:
                rows.setdefault((k, c), []).append(v)
    for (k, c), v in sorted(rows.items()):
        print(f"{k[:70]:<70} {c:<28} n={len(v):<4} avg={sum(v) / len(v):.4g}")


if __name__ == "__main__":
    main(sys.argv[1:])
