#!/usr/bin/env python3
"""Dump the per-kernel statistics of a rocprofv3 run (rocpd SQLite output) as text.

usage: tools/rocprof_summary.py gpurun_out/prof_xxx/yyy_results.db > profiles/rNN_kernel_stats.txt
"""
import sqlite3
import sys


def main(path, top=25):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print("# rocpd top_kernels view reports microseconds; sorted by total duration")
    print(f"{'calls':>7} {'total_ms':>12} {'avg_us':>12} {'pct':>7}  kernel")
    for name, calls, total, avg, pct in rows[:top]:
        print(f"{calls:>7} {total / 1e3:>12.3f} {avg:>12.3f} {pct:>7.2f}  {name[:110]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
