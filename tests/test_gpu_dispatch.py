"""Shape -> kernel dispatch table of the search path, asserted through mmidx_get_dispatch.

Every row builds a small index of the given shape, runs one batch search (parity with the oracle on a few queries: the kernel that
ran must also be right) and compares the kernel families the library reports for the coarse stage, pass A, the pair pre-filter and
pass B with the documented ones.  The gates live in csrc/mmidx_api.hip (run_coarse, search_batch_device, launch_scan_grouped,
launch_mfma_common, launch_mfma_kc, passa_q_applies, passa_mfma_applies); a change there has to change this table and DESIGN.md with it."""
import numpy as np
import pytest

import synth
from test_gpu_parity import assert_same, mi, oracle_ivfpq  # noqa: F401

pytestmark = pytest.mark.gpu

# kind, D, m, ks, C, w, n, nq, k, transform, options -> coarse, pass_a, pre, pass_b
ROWS = [
    # IVFPQ, the headline's shape at small scale: short lists (K3h takes lists of >= 4096 codes on average)
    ("ivfpq", 128, 16, 256, 300, 8, 60000, 50, 10, 0, {}, ("K1e'+K1f(front_sel)", "K3", "-", "K3m")),
    # ... long lists, from 1.25 queries per non-empty list: K3q (round 6: four queries of a list per block, decided on integers)
    ("ivfpq", 128, 16, 256, 8, 4, 40000, 30, 100, 0, {}, ("K1a+K1b(exact)", "K3q", "-", "K3m")),
    # ... fewer queries than that: K3h, a block per query
    ("ivfpq", 128, 16, 256, 8, 4, 40000, 6, 100, 0, {}, ("K1a+K1b(exact)", "K3h", "-", "K3m")),
    # ... K3q switched off: K3h; and from 8 queries per list K3ma
    ("ivfpq", 128, 16, 256, 8, 4, 40000, 30, 100, 0, {"passa_q": 0}, ("K1a+K1b(exact)", "K3h", "-", "K3m")),
    ("ivfpq", 128, 16, 256, 8, 4, 40000, 80, 100, 0, {"passa_q": 0}, ("K1a+K1b(exact)", "K3ma", "-", "K3m")),
    ("ivfpq", 128, 16, 256, 8, 4, 40000, 80, 100, 0, {}, ("K1a+K1b(exact)", "K3q", "-", "K3m")),
    # ... k + 1 > 152: neither K3q (192 candidates per query at most) nor K3ma (k + 1 <= 128) applies
    ("ivfpq", 128, 16, 256, 8, 4, 40000, 80, 200, 0, {}, ("K1a+K1b(exact)", "K3h", "-", "K3m")),
    # RandomRotation at D = 128: K3q rotates its residuals itself, K3m serves pass B (k_pair_rotate)
    ("ivfpq", 128, 16, 256, 8, 4, 40000, 30, 20, 1, {}, ("K1a+K1b(exact)", "K3q", "-", "K3m")),
    # the 1024-d 64 x 256 shape (YFCC100MExample.java:85-90), RandomPermutation: K3h<64>, K3s in front of K3mk
    ("ivfpq", 1024, 64, 256, 6, 6, 30000, 20, 30, 2, {}, ("K1a+K1b(exact)", "K3h", "K3s", "K3mk")),
    # 12-dimensional sub-quantizers: outside K3m (dsub in {4, 8, 16}); K3g's generic instance takes them
    ("ivfpq", 96, 8, 256, 6, 6, 12000, 20, 10, 0, {}, ("K1a+K1b(exact)", "K3", "-", "K3g")),
    # short codes (ks > 256): the exact scan everywhere
    ("ivfpq", 32, 4, 512, 6, 6, 12000, 20, 10, 0, {}, ("K1a+K1b(exact)", "K3", "-", "K3(exact scan: no K3f instance for this shape)")),
    # m = 128 (Example.java:74 names pq_1024_128x8): the lookup table lives in global scratch for pass A, K3mk behind it (K3s sits out: its fp32 tables exist for m in {8, 16, 32, 64})
    ("ivfpq", 1024, 128, 256, 6, 6, 6000, 12, 10, 0, {}, ("K1a+K1b(exact)", "K3(table in two halves)", "-", "K3mk")),
    # K3m switched off: K3g
    ("ivfpq", 128, 16, 256, 8, 4, 40000, 30, 100, 0, {"no_mfma": 1}, ("K1a+K1b(exact)", "K3q", "-", "K3g")),
    # flat PQ (cfg2's shape at small scale): chunk 0 through K3h, the others through K3m
    ("pq", 128, 8, 256, 0, 0, 80000, 40, 100, 0, {}, ("-", "K3h", "-", "K3m")),
    # w = 1: a single pass
    ("ivfpq", 64, 8, 256, 6, 1, 12000, 20, 10, 0, {}, ("K1a+K1b(exact)", "K3(single pass)", "-", "-")),
]


@pytest.mark.parametrize("row", ROWS, ids=[f"{r[0]}-D{r[1]}-m{r[2]}-ks{r[3]}-C{r[4]}-w{r[5]}-nq{r[7]}-k{r[8]}-tr{r[9]}" + ("-" + "+".join(r[10]) if r[10] else "") for r in ROWS])
def test_dispatch_table(mi, oracle, row):
    kind, D, m, ks, C, w, n, nq, k, tr, opts, want = row
    rng = np.random.default_rng(D + m + nq)
    ds = D // m
    rot = np.linalg.qr(rng.standard_normal((D, D)))[0] if tr == 1 else None
    perm = oracle.random_permutation(1, D) if tr == 2 else None
    if kind == "ivfpq":
        mu = 0.5 * rng.standard_normal((C, D))
        base = mu[rng.integers(0, C, n)] + rng.standard_normal((n, D))
        pq = np.stack([synth.kmeans((mu[rng.integers(0, C, 2 * ks + 500)] - base[:2 * ks + 500])[:, s * ds:(s + 1) * ds], ks, iters=1, seed=s) for s in range(m)])
        ix = mi.IVFPQ(D, n, False, "", m, ks, tr, C, 512, rot=rot)
        ix.loadCoarseQuantizer(mu)
        ix.setW(w)
        ref = oracle_ivfpq(oracle, {"coarse": mu, "pq": pq}, D, m, ks, C, w, tr=tr, perm=perm, rot=rot)
    else:
        base = rng.standard_normal((n, D))
        pq = np.stack([synth.kmeans(base[:2 * ks + 500, s * ds:(s + 1) * ds], ks, iters=1, seed=s) for s in range(m)])
        ix = mi.PQ(D, n, False, "", m, ks, tr, 512, rot=rot)
        ref = oracle.OracleIndex(oracle.KIND_PQ, D, m, ks, transform=tr, perm=perm, rot=rot)
    ix.loadProductQuantizer(pq)
    ref.set_pq(pq)
    for a, b in opts.items():
        ix.set_option(a, b)
    ix.indexVectors([str(i) for i in range(n)], base)
    Q = np.concatenate([0.5 * (base[:nq // 2] + base[100:100 + nq // 2]), base[200:200 + nq - nq // 2] + 0.01 * rng.standard_normal((nq - nq // 2, D))])
    got = ix.search_batch(k, Q)
    d = ix.get_dispatch()
    ix.close()
    assert (d["coarse"], d["pass_a"], d["pre"], d["pass_b"]) == want, d
    npar = min(nq, 6)
    ref.add_vectors(base)
    assert_same(tuple(a[:npar] for a in got), ref.search_batch(Q[:npar], k))
