#!/usr/bin/env python3
"""tests/golden/*.npz -> one directory of CSV files per fixture, for CrossCheck.java.

Doubles travel as 16-digit hex bit patterns (exact); the quantizer files are ALSO written in the reference's own text
formats (`*_plain.csv`: one centroid per line, comma separated -- AbstractFeatureAggregator.readQuantizer, IVFPQ.java:275-288)
with repr() precision, which round-trips binary64 exactly through Double.parseDouble.

  python tools/java_crosscheck/export_fixtures.py /tmp/mmidx_fixtures
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def hexrows(a):
    a = np.ascontiguousarray(a, np.float64)
    return "\n".join(",".join(f"{int(x):016x}" for x in row.view(np.uint64)) for row in a.reshape(a.shape[0], -1)) + "\n"


def plain(a):
    # two columns at least: readQuantizer skips lines without a comma (AbstractFeatureAggregator.java:241-250)
    return "\n".join(",".join(repr(float(x)) for x in row) for row in a) + "\n"


def main(out):
    g = os.path.join(ROOT, "tests", "golden")
    for name in ("ivfpq_small", "ivfpq_perm", "ivfpq_ties", "pq_small"):
        z = np.load(os.path.join(g, name + ".npz"))
        d = os.path.join(out, name)
        os.makedirs(d, exist_ok=True)
        D, m, ks, k = int(z["D"]), int(z["m"]), int(z["ks"]), int(z["k"])
        C = int(z["C"]) if "C" in z else 0
        w = int(z["w"]) if "w" in z else 0
        tr = int(z["transform"]) if "transform" in z else 0
        base, queries = z["base"], z["queries"]
        open(os.path.join(d, "meta.csv"), "w").write(f"{D},{C},{m},{ks},{w},{k},{tr},{len(base)},{len(queries)}\n")
        open(os.path.join(d, "base.csv"), "w").write(hexrows(base))
        open(os.path.join(d, "queries.csv"), "w").write(hexrows(queries))
        open(os.path.join(d, "pq_plain.csv"), "w").write(plain(z["pq"].reshape(m * ks, D // m)))
        if C:
            open(os.path.join(d, "coarse_plain.csv"), "w").write(plain(z["coarse"]))
        # expected answers in the same form CrossCheck writes: id:distance-bits per result
        ids, ds, cnt = z["ids"], z["dists"], z["counts"]
        with open(os.path.join(d, "expected.answers.csv"), "w") as f:
            for q in range(len(cnt)):
                f.write(",".join(f"{int(ids[q, i])}:{int(np.float64(ds[q, i]).view(np.uint64)):x}" for i in range(int(cnt[q]))) + "\n")
    print("fixtures exported to", out)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/mmidx_fixtures")
