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


def rotation_case(fx, out, name):
    """A RandomRotation fixture carries no expected answers: the matrix is EJML's.  CrossCheck dumped it; the oracle runs with it.
    Ids must agree exactly, distances to 1e-12 relative (assumption A2: EJML's summation order in CommonOps.mult)."""
    import numpy as np

    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    from oracle import oracle as o

    def hexm(path):
        return np.array([[int(x, 16) for x in l.strip().split(",")] for l in open(path) if l.strip()], dtype=np.uint64).view(np.float64)

    def plainm(path):
        return np.array([[float(x) for x in l.strip().split(",")] for l in open(path) if "," in l])

    d = os.path.join(fx, name)
    D, C, m, ks, w, k, tr, n, nq = [int(x) for x in open(os.path.join(d, "meta.csv")).read().strip().split(",")]
    rot = hexm(os.path.join(out, name + ".rotation.csv"))
    ref = o.OracleIndex(o.KIND_IVFPQ, D, m, ks, C, transform=1, rot=rot)
    ref.set_coarse(plainm(os.path.join(d, "coarse_plain.csv")))
    ref.set_pq(plainm(os.path.join(d, "pq_plain.csv")).reshape(m, ks, D // m))
    ref.set_w(w)
    ref.add_vectors(hexm(os.path.join(d, "base.csv")))
    ids, ds, cnt = ref.search_batch(hexm(os.path.join(d, "queries.csv")), k)
    got = [l.strip() for l in open(os.path.join(out, name + ".answers.csv"))]
    worst, wrong = 0.0, 0
    for q in range(len(cnt)):
        items = [x.split(":") for x in got[q].split(",")] if q < len(got) and got[q] else []
        gi = [int(a) for a, _ in items]
        gd = np.array([int(b, 16) for _, b in items], dtype=np.uint64).view(np.float64)
        if gi != ids[q, :cnt[q]].tolist():
            wrong += 1
            continue
        if len(gd):
            worst = max(worst, float(np.max(np.abs(gd - ds[q, :cnt[q]]) / np.maximum(np.abs(ds[q, :cnt[q]]), 1e-300))))
    ortho = float(np.max(np.abs(rot @ rot.T - np.eye(D))))
    if wrong or worst > 1e-12:
        print(f"FAIL {name}: {wrong} of {len(cnt)} queries with other ids; largest relative distance difference {worst:.3g} (matrix: max |R R^T - I| = {ortho:.3g})")
        return 1
    print(f"ok   {name}: {len(cnt)} queries, ids identical, distances within {worst:.3g} relative (A2; max |R R^T - I| = {ortho:.3g})")
    return 0


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
        if not os.path.exists(os.path.join(fx, name, "expected.answers.csv")):
            bad += rotation_case(fx, out, name)
            continue
        exp = [l.strip() for l in open(os.path.join(fx, name, "expected.answers.csv"))]
        got = [l.strip() for l in open(os.path.join(out, name + ".answers.csv"))]
        diff = [q for q in range(len(exp)) if q >= len(got) or exp[q] != got[q]]
        flagged = " (FLAGGED tie fixture: assumption A1)" if name.endswith("_ties") else ""
        if diff:
            bad += 1
            q = diff[0]
            print(f"FAIL {name}{flagged}: {len(diff)} of {len(exp)} queries differ; first: query {q}\n  expected {exp[q]}\n  java     {got[q] if q < len(got) else '<missing>'}")
        else:
            print(f"ok   {name}{flagged}: {len(exp)} queries, ids and distance bits identical")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
