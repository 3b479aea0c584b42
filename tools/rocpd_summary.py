#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) kernel trace as a per-kernel stats table
(name, calls, total/avg/min/max ns, % of GPU time) -- the same content as
rocprofv3's *_kernel_stats.csv.  Usage: rocpd_summary.py results.db [out.csv]"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    return name if len(name) < 140 else name[:137] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        d = e - s
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1
    out = sorted(agg.items(), key=lambda kv: -kv[1][1])
    w = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for name, (c, t, mn, mx) in out:
        w.writerow([short(name), c, t, round(t / c, 1), mn, mx, round(100.0 * t / tot, 2)])


if __name__ == "__main__":
    main()
