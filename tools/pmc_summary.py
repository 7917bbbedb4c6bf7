#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the
microarch guide prescribes).  FETCH_SIZE is doubled (gfx950 counts 128-B requests as 64 B on wide
streams: calibrated on k_points_to_mont, a known 6 GiB read); both are in KiB.
    python tools/pmc_summary.py fetch.db write.db"""
import sqlite3
import sys


def load(path, name):
    cur = sqlite3.connect(path).cursor()
    out = {}
    for k, v in cur.execute("select kernel_name, value from counters_collection where counter_name=?", (name,)):
        out.setdefault(k, []).append(v)
    return out


def main(fdb, wdb):
    f, w = load(fdb, "FETCH_SIZE"), load(wdb, "WRITE_SIZE")
    print(f"{'kernel':<60} {'launches':>8} {'read_GB/launch':>15} {'write_GB/launch':>16}")
    for k in sorted(set(f) | set(w), key=lambda k: -(sum(f.get(k, [0])) + sum(w.get(k, [0])))):
        fr = 2 * 1024 * sum(f.get(k, [0])) / max(1, len(f.get(k, [0]))) / 1e9
        wr = 1024 * sum(w.get(k, [0])) / max(1, len(w.get(k, [0]))) / 1e9
        print(f"{k[:60]:<60} {len(f.get(k, w.get(k))):>8} {fr:>15.3f} {wr:>16.3f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
