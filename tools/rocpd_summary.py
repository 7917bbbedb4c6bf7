#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (`rocprofv3 --kernel-trace --stats`) as a per-kernel
table: calls, total / average / min / max duration, share of GPU kernel time, VGPRs.
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [d[1] for d in cur.execute("pragma table_info('kernels')")]
    rows = cur.execute("select * from kernels").fetchall()
    ix = {c: i for i, c in enumerate(cols)}
    name_col = "name" if "name" in ix else "kernel_name"
    agg = {}
    for r in rows:
        nm = r[ix[name_col]]
        dur = (r[ix["end"]] - r[ix["start"]]) if "end" in ix else r[ix["duration"]]
        a = agg.setdefault(nm, [0, 0, 1 << 62, 0, r])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values()) or 1
    print(f"# source: {path}")
    print(f"# total kernel time {tot/1e6:.3f} ms over {sum(a[0] for a in agg.values())} dispatches")
    print(f"{'kernel':<70} {'calls':>6} {'total_ms':>11} {'avg_ms':>10} {'min_ms':>10} {'max_ms':>10} {'pct':>6}")
    for nm, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = nm if len(nm) <= 70 else nm[:67] + "..."
        print(f"{short:<70} {a[0]:>6} {a[1]/1e6:>11.3f} {a[1]/a[0]/1e6:>10.4f} {a[2]/1e6:>10.4f} {a[3]/1e6:>10.4f} {100*a[1]/tot:>6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
