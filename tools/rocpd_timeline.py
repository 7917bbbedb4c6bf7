#!/usr/bin/env python3
"""Dev tool: kernel timeline (start, duration, queue/stream) of a rocprofv3 rocpd database, between the starts of the
last-but-one and the last dispatch of a kernel whose name contains PATTERN (default k_accumulate).
    python tools/rocpd_timeline.py x_results.db [pattern]"""
import sqlite3
import sys


def main(path, pat="k_accumulate"):
    cur = sqlite3.connect(path).cursor()
    cols = [d[1] for d in cur.execute("pragma table_info('kernels')")]
    ix = {c: i for i, c in enumerate(cols)}
    name_col = "name" if "name" in ix else "kernel_name"
    rows = sorted(cur.execute("select * from kernels").fetchall(), key=lambda r: r[ix["start"]])
    marks = [r[ix["start"]] for r in rows if pat in r[ix[name_col]]]
    if len(marks) < 3:
        print("not enough dispatches of", pat)
        return
    t0, t1 = marks[-3], marks[-2]
    qcol = next((c for c in ("queue_id", "stream_id", "queue") if c in ix), None)
    print(f"# window {(t1 - t0) / 1e6:.3f} ms; columns: start_ms  dur_ms  end_ms  {qcol}  kernel")
    for r in rows:
        if r[ix["end"]] < t0 or r[ix["start"]] > t1:
            continue
        nm = r[ix[name_col]]
        nm = nm.replace("void blz::", "").replace("blz::", "")[:60]
        q = r[ix[qcol]] if qcol else "-"
        print(f"{(r[ix['start']] - t0) / 1e6:9.3f} {(r[ix['end']] - r[ix['start']]) / 1e6:9.3f} {(r[ix['end']] - t0) / 1e6:9.3f}  {q}  {nm}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
