#!/usr/bin/env python3
"""Compares CrossCheck.java's dumps with the fixtures' expected answers (and the permutation known answers).

  python tools/java_crosscheck/compare.py /tmp/mmidx_fixtures /tmp/mmidx_java_out

Exit status 0 = the reference's classes return exactly the ids and distance bits the oracle pinned: "parity unpinned" can be
struck from oracle/mmidx_oracle.h and DESIGN.md.  ivfpq_ties is the FLAGGED fixture: a mismatch there (only) means assumption
A1 (LingPipe BoundedPriorityQueue tie order) is wrong, and the report shows the first differing query."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(fx, out):
    bad = 0
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "kat_hand.json")))["kat3_jdk"]
    perm = {int(l.split(",")[0]): [int(x) for x in l.strip().split(",")[1:]] for l in open(os.path.join(out, "permutation.csv"))}
    checks = [(perm[3] == kat["perm_seed1_dim3"], "RandomPermutation(1, 3)"), (perm[8] == kat["perm_seed1_dim8"], "RandomPermutation(1, 8)"),
              (perm[128][:16] == kat["perm_seed1_dim128_first16"], "RandomPermutation(1, 128) prefix"),
              (sum(i * p for i, p in enumerate(perm[128])) == kat["perm_seed1_dim128_sum_i_times_p"], "RandomPermutation(1, 128) checksum")]
    for ok, what in checks:
        print(("ok   " if ok else "FAIL ") + what)
        bad += not ok
    for name in sorted(os.listdir(fx)):
        exp = [l.strip() for l in open(os.path.join(fx, name, "expected.answers.csv"))]
        got = [l.strip() for l in open(os.path.join(out, name + ".answers.csv"))]
        diff = [q for q in range(len(exp)) if q >= len(got) or exp[q] != got[q]]
        flagged = " (FLAGGED tie fixture: assumption A1)" if name == "ivfpq_ties" else ""
        if diff:
            bad += 1
            q = diff[0]
            print(f"FAIL {name}{flagged}: {len(diff)} of {len(exp)} queries differ; first: query {q}\n  expected {exp[q]}\n  java     {got[q] if q < len(got) else '<missing>'}")
        else:
            print(f"ok   {name}{flagged}: {len(exp)} queries, ids and distance bits identical")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
