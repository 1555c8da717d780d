#!/usr/bin/env python3
"""Per-kernel launch durations of a rocprofv3 kernel trace (rocpd SQLite), clustered: a bench run mixes workloads (headline,
hard, spread ...) whose launches of one kernel differ by orders of magnitude, and the trace's per-kernel average blends them.
Launches are sorted by duration and cut where the next one is more than `ratio` times longer.

usage: tools/kernel_calls.py kt_results.db [name-regex=k_scan_mfma|k_mfma_verify] [ratio=1.6]
"""
import re
import sqlite3
import sys


def main(path, rx="k_scan_mfma|k_mfma_verify", ratio=1.6):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    ni, si, ei = cols.index("name"), cols.index("start"), cols.index("end")
    by = {}
    for r in cur.execute("select * from kernels"):
        name = str(r[ni])
        if re.search(rx, name):
            by.setdefault(name[:70], []).append((r[ei] - r[si]) / 1e3)
    for name, d in sorted(by.items()):
        d.sort()
        print(f"{name}: {len(d)} launches")
        lo = 0
        for i in range(1, len(d) + 1):
            if i == len(d) or d[i] > ratio * max(d[lo], 1.0) and d[i] > ratio * d[i - 1]:
                c = d[lo:i]
                print(f"    {len(c):5d} launches  avg {sum(c) / len(c):10.1f} us  (min {c[0]:.1f}, max {c[-1]:.1f})")
                lo = i


if __name__ == "__main__":
    a = sys.argv
    main(a[1], a[2] if len(a) > 2 else "k_scan_mfma|k_mfma_verify", float(a[3]) if len(a) > 3 else 1.6)
