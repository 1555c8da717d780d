#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes for kernels matching a pattern.
usage: tools/pmc_summary.py 'gpurun_out/pmc_r01c_*' k_scan > profiles/r01c_pmc_k_scan.txt"""
import collections
import csv
import glob
import re
import sys


def main(pattern, kpat):
    print(f"# rocprofv3 --pmc passes (one directory per pass), kernels matching '{kpat}'")
    print("# per counter: mean value per dispatch, grouped by (kernel, grid size); FETCH_SIZE/WRITE_SIZE in KiB")
    for d in sorted(glob.glob(pattern)):
        import os
        fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        f = fs[0] if fs else d + "/pmc_counter_collection.csv"
        try:
            rows = list(csv.DictReader(open(f)))
        except FileNotFoundError:
            continue
        agg = collections.defaultdict(list)
        for r in rows:
            if re.search(kpat, r["Kernel_Name"]) and "rocprim" not in r["Kernel_Name"]:
                kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[(r["Counter_Name"], kn, r["Grid_Size"])].append(float(r["Counter_Value"]))
        print(f"## {d}")
        for (cn, kn, gs), v in sorted(agg.items()):
            # a bench run mixes workloads: the launches of the busiest one (values within a factor two of the largest) are reported
            # next to the plain mean, which the near-empty launches of the others dilute
            top = [x for x in v if x >= 0.5 * max(v)] if max(v) > 0 else v
            print(f"{cn:24s} {kn:42s} grid={gs:>10s} n={len(v):3d} mean={sum(v) / len(v):.6g}  top n={len(top):3d} mean={sum(top) / len(top):.6g}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "k_scan")
