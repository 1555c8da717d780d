#!/usr/bin/env python3
"""Timeline of a few steady-state search steps out of a rocprofv3 kernel trace (rocpd SQLite): every kernel between the
`first`-th and the (`first` + `steps`)-th launch of the step's anchor kernel, with the idle gap in front of each and a
per-step summary (busy, idle, launches under 15 us).

usage: tools/timeline_steps.py kt_results.db [anchor=k_scan_hist] [first=10] [steps=3]
"""
import sqlite3
import sys


def main(path, anchor="k_scan_hist", first=10, steps=3):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    ni, si, ei = cols.index("name"), cols.index("start"), cols.index("end")
    rows = [(r[si], r[ei], str(r[ni])) for r in cur.execute("select * from kernels order by start")]
    idx = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(idx) < first + steps + 1:
        print(f"only {len(idx)} launches of {anchor}")
        return
    # a step starts at the first kernel after the previous step's last one: cut at the coarse stage's first kernel
    lo, hi = idx[first], idx[first + steps]
    while lo > 0 and "k_split_bf16" not in rows[lo][2] and lo > idx[first - 1]:
        lo -= 1
    while hi > 0 and "k_split_bf16" not in rows[hi][2] and hi > idx[first + steps - 1]:
        hi -= 1
    t0 = rows[lo][0]
    prev_end = t0
    busy = idle = small = n_small = 0.0
    print(f"# {steps} steps, launches {first}..{first + steps - 1} of {anchor}; times in us")
    for s, e, name in rows[lo:hi]:
        gap = (s - prev_end) / 1e3
        dur = (e - s) / 1e3
        print(f"{(s - t0) / 1e3:10.1f}  gap {gap:6.1f}  dur {dur:8.1f}  {name[:80]}")
        busy += dur
        idle += max(0.0, gap)
        if dur < 15.0:
            small += dur
            n_small += 1
        prev_end = max(prev_end, e)
    span = (prev_end - t0) / 1e3
    print(f"# per step: span {span / steps:.1f} us, kernels busy {busy / steps:.1f}, idle between kernels {idle / steps:.1f}, "
          f"{n_small / steps:.1f} launches under 15 us = {small / steps:.1f} us")


if __name__ == "__main__":
    a = sys.argv
    main(a[1], a[2] if len(a) > 2 else "k_scan_hist", int(a[3]) if len(a) > 3 else 10, int(a[4]) if len(a) > 4 else 3)
