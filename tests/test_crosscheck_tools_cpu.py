"""The Java cross-check harness (tools/java_crosscheck/) end to end WITHOUT a JDK: the fixtures are exported, the part CrossCheck.java
plays -- answers per fixture in `id:distance-bits` form, the rotation matrix of the RandomRotation case -- is produced by the oracle
with a stand-in orthogonal matrix, and compare.py must accept it (and must reject a corrupted answer).  This keeps the one command
that would pin the oracle to the reference's classes working until a box with a JDK and the four jars exists."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", "java_crosscheck", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_export_and_compare_round_trip(tmp_path, oracle, capsys):
    exp, cmp_ = _load("export_fixtures"), _load("compare")
    fx, out = str(tmp_path / "fx"), str(tmp_path / "out")
    os.makedirs(out)
    exp.main(fx)
    names = sorted(os.listdir(fx))
    assert {"ivfpq_small", "ivfpq_perm", "ivfpq_ties", "pq_small", "ivfpq_k3m_overlap", "ivfpq_k3m_ties", "ivfpq_d1024_m64", "ivfpq_rotation"} <= set(names)
    # KAT-3 as CrossCheck prints it
    with open(os.path.join(out, "permutation.csv"), "w") as f:
        for dim in (3, 8, 128):
            f.write(",".join([str(dim)] + [str(int(x)) for x in oracle.random_permutation(1, dim)]) + "\n")
    for name in names:
        d = os.path.join(fx, name)
        if os.path.exists(os.path.join(d, "expected.answers.csv")):
            open(os.path.join(out, name + ".answers.csv"), "w").write(open(os.path.join(d, "expected.answers.csv")).read())
            continue
        # the rotation case: a stand-in for EJML's matrix, answers from the oracle run with it
        D, C, m, ks, w, k, tr, n, nq = [int(x) for x in open(os.path.join(d, "meta.csv")).read().strip().split(",")]
        rot = np.linalg.qr(np.random.default_rng(9).standard_normal((D, D)))[0]
        open(os.path.join(out, name + ".rotation.csv"), "w").write(exp.hexrows(rot))

        def hexm(path):
            return np.array([[int(x, 16) for x in l.strip().split(",")] for l in open(path) if l.strip()], dtype=np.uint64).view(np.float64)

        ref = oracle.OracleIndex(oracle.KIND_IVFPQ, D, m, ks, C, transform=1, rot=rot)
        ref.set_coarse(np.array([[float(x) for x in l.split(",")] for l in open(os.path.join(d, "coarse_plain.csv"))]))
        ref.set_pq(np.array([[float(x) for x in l.split(",")] for l in open(os.path.join(d, "pq_plain.csv"))]).reshape(m, ks, D // m))
        ref.set_w(w)
        ref.add_vectors(hexm(os.path.join(d, "base.csv")))
        ids, ds, cnt = ref.search_batch(hexm(os.path.join(d, "queries.csv")), k)
        with open(os.path.join(out, name + ".answers.csv"), "w") as f:
            for q in range(len(cnt)):
                f.write(",".join(f"{int(ids[q, i])}:{int(np.float64(ds[q, i]).view(np.uint64)):x}" for i in range(int(cnt[q]))) + "\n")
    assert cmp_.main(fx, out) == 0
    # a single flipped distance bit in one fixture must be reported
    p = os.path.join(out, "ivfpq_k3m_overlap.answers.csv")
    lines = open(p).read().split("\n")
    first = lines[0].split(",")
    i0, b0 = first[0].split(":")
    first[0] = f"{i0}:{int(b0, 16) ^ 1:x}"
    lines[0] = ",".join(first)
    open(p, "w").write("\n".join(lines))
    assert cmp_.main(fx, out) == 1
    assert "FAIL ivfpq_k3m_overlap" in capsys.readouterr().out
