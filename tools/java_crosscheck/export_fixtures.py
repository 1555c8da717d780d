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
    extra_cases(out)
    print("fixtures exported to", out)


def write_case(out, name, D, C, m, ks, w, k, tr, coarse, pq, base, queries, expected):
    d = os.path.join(out, name)
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "meta.csv"), "w").write(f"{D},{C},{m},{ks},{w},{k},{tr},{len(base)},{len(queries)}\n")
    open(os.path.join(d, "base.csv"), "w").write(hexrows(base))
    open(os.path.join(d, "queries.csv"), "w").write(hexrows(queries))
    open(os.path.join(d, "pq_plain.csv"), "w").write(plain(pq.reshape(m * ks, D // m)))
    open(os.path.join(d, "coarse_plain.csv"), "w").write(plain(coarse))
    if expected is not None:
        ids, ds, cnt = expected
        with open(os.path.join(d, "expected.answers.csv"), "w") as f:
            for q in range(len(cnt)):
                f.write(",".join(f"{int(ids[q, i])}:{int(np.float64(ds[q, i]).view(np.uint64)):x}" for i in range(int(cnt[q]))) + "\n")


def extra_cases(out):
    """Cases generated HERE (fixed seeds, expected answers from the oracle at export time) instead of committed: the shapes the
    kernels of rounds 4 and 5 serve, too large for tests/golden/.
      ivfpq_k3m_overlap    D = 128, 16 x 256, overlapping cells (sigma 1.0), w = C: every probe is scanned -- K3m's pass B
      ivfpq_k3m_ties       the same with every vector three times (FLAGGED: assumption A1 on the matrix-core path's shape)
      ivfpq_d1024_m64      YFCC100MExample.java:85-90's shape: D = 1024, 64 x 256, RandomPermutation -- K3mk / k_scan_hist<64, ..>
      ivfpq_rotation       TransformationType.RandomRotation: the matrix is EJML's (RandomRotation.java:30-35), so CrossCheck dumps it
                           (rotate(e_i) = row i) and compare.py runs the oracle WITH that matrix: ids exact, distances to 1e-12 (A2)"""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import synth
    from oracle import oracle as o

    def problem(seed, D, C, m, n, sigma, dup=1):
        rng = np.random.default_rng(seed)
        mu = rng.standard_normal((C, D))
        base = mu[rng.integers(0, C, n // dup)] + sigma * rng.standard_normal((n // dup, D))
        base = np.concatenate([base] * dup)[rng.permutation((n // dup) * dup)]
        ds = D // m
        res = mu[rng.integers(0, C, 2000)] + sigma * rng.standard_normal((2000, D))
        cl = ((res[:, None, :] - mu[None, :, :]) ** 2).sum(-1).argmin(1)
        r = mu[cl] - res
        return rng, mu, base, r

    for name, seed, D, C, m, ks, w, k, n, sigma, tr, dup in (("ivfpq_k3m_overlap", 501, 128, 8, 16, 256, 8, 100, 3000, 1.0, 0, 1),
                                                             ("ivfpq_k3m_ties", 502, 128, 6, 16, 256, 6, 20, 2400, 1.0, 0, 3),
                                                             ("ivfpq_d1024_m64", 503, 1024, 6, 64, 256, 6, 30, 1200, 0.6, 2, 1),
                                                             ("ivfpq_rotation", 504, 32, 8, 4, 64, 4, 10, 1500, 0.5, 1, 1)):
        rng, mu, base, r = problem(seed, D, C, m, n, sigma, dup)
        ds = D // m
        perm = o.random_permutation(1, D) if tr == 2 else None
        rt = r[:, perm] if perm is not None else r  # (the codebooks are learned on TRANSFORMED residuals, ProductQuantizationLearning.java:176-178; a rotation case uses plain ones)
        pq = np.stack([synth.kmeans(rt[:, s * ds:(s + 1) * ds], ks, iters=3, seed=s) for s in range(m)])
        queries = np.concatenate([0.5 * (base[:8] + base[50:58]), base[100:108] + 0.01 * rng.standard_normal((8, D))])
        expected = None
        if tr != 1:
            ref = o.OracleIndex(o.KIND_IVFPQ, D, m, ks, C, transform=tr, perm=perm)
            ref.set_coarse(mu)
            ref.set_pq(pq)
            ref.set_w(w)
            ref.add_vectors(base)
            expected = ref.search_batch(queries, k)
        write_case(out, name, D, C, m, ks, w, k, tr, mu, pq, base, queries, expected)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/mmidx_fixtures")
