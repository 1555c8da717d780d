"""GPU parity of K3ma -- pass A of the IVFADC search on the matrix cores (csrc/mmidx_scan_mfma_a.h; option "passa_mfma").

Reference loop: the probe-0 iteration of computeKnnIVFADC, IVFPQ.java:414-447.  The instance is forced (the default picks it
from 8 queries per list of a long-list index) on small indexes whose shapes walk its code paths: one to four row tiles per
list, several pieces per list, lists shorter than k + 1 (the redo path), ties, permutation / rotation, every (D, dsub) pair it
is instantiated for.  Ids and distance bits are the oracle's; `passa_mfma_launches` proves the instance ran.
"""
import numpy as np
import pytest

import synth
from test_gpu_parity import assert_same, mi, oracle_ivfpq  # noqa: F401  (mi: the module fixture)

pytestmark = pytest.mark.gpu


def _problem(rng, D, m, C, n, dup=1, spread=1.0):
    mu = 0.5 * rng.standard_normal((C, D))
    base = mu[rng.integers(0, C, n // dup)] + spread * rng.standard_normal((n // dup, D))
    base = np.concatenate([base] * dup)[rng.permutation((n // dup) * dup)]
    ds = D // m
    pq = np.stack([synth.kmeans((mu[rng.integers(0, C, 3000)] - base[:3000])[:, s * ds:(s + 1) * ds], 256, iters=2, seed=s) for s in range(m)])
    return mu, base, pq


@pytest.mark.parametrize("D,m,C,n,w,k,tr,dup,nq", [
    (128, 16, 6, 30000, 3, 100, 0, 1, 90),     # <NJ 4, dsub 8>, the headline's shape: ~15 rows per list (one row tile)
    (128, 16, 4, 24000, 4, 100, 0, 1, 400),    # ~100 rows per list: full groups of 64 and a remainder group, four row tiles
    (128, 16, 5, 24000, 5, 50, 2, 3, 200),     # RandomPermutation; every vector three times: ties at the k-th distance
    (128, 8, 6, 24000, 2, 30, 0, 1, 150),      # <4, 16>
    (128, 32, 5, 20000, 5, 100, 0, 1, 120),    # <4, 4>: 32-byte codes
    (64, 8, 4, 20000, 4, 127, 2, 1, 100),      # <2, 8>, k + 1 = 128: the largest K1 the instance takes
    (64, 4, 4, 16000, 3, 10, 0, 2, 130),       # <2, 16>
    (64, 16, 4, 16000, 4, 30, 2, 2, 70),       # <2, 4>
    (32, 4, 7, 12000, 7, 1, 0, 1, 300),        # <1, 8>, k = 1
    (32, 2, 3, 9000, 3, 20, 0, 1, 64),         # <1, 16>
    (32, 8, 3, 9000, 3, 10, 0, 1, 40),         # <1, 4>
    (128, 16, 6, 24000, 6, 100, 1, 1, 80),     # RandomRotation (orthogonal): the rows come from k_pair_rotate
])
def test_passa_mfma_forced(mi, oracle, D, m, C, n, w, k, tr, dup, nq):
    """Every query's nearest list through sweep 1 (best four per slot) -> k_a1_select -> sweep 2 (bitmap) -> k_a1_verify; pass B
    behind it as usual.  Queries: self-perturbed vectors, midpoints between vectors, independent Gaussians, the centroids
    themselves.  With pieces of 1024 / 256 codes (`mfma_sub`: several items per list, k_a1_select merges their values), with K3m
    off for pass B (K3ma does not depend on it ... but shares its tables), and with K3ma off: the oracle's answers every time."""
    ks = 256
    rng = np.random.default_rng(7 * D + m + k)
    mu, base, pq = _problem(rng, D, m, C, n, dup)
    n = len(base)
    rot = np.linalg.qr(rng.standard_normal((D, D)))[0] if tr == 1 else None
    ix = mi.IVFPQ(D, n, False, "", m, ks, tr, C, 512, rot=rot)
    ix.loadCoarseQuantizer(mu)
    ix.loadProductQuantizer(pq)
    ix.setW(w)
    ref = oracle_ivfpq(oracle, {"coarse": mu, "pq": pq}, D, m, ks, C, w, tr=tr, perm=oracle.random_permutation(1, D) if tr == 2 else None, rot=rot)
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    nself = nq - 24 - 8 - min(C, 2)
    Q = np.concatenate([0.5 * (base[:24] + base[100:124]), rng.standard_normal((8, D)), base[200:200 + nself] + 0.01 * rng.standard_normal((nself, D)), mu[:2]])
    want = ref.search_batch(Q, k)
    for force, sub in ((1, 0), (1, 1024), (1, 256), (0, 0)):
        ix.set_option("passa_mfma", force)
        ix.set_option("mfma_sub", sub)
        ix.set_profiling(True)
        got = ix.search_batch(k, Q)
        st = ix.get_stats()
        assert_same(got, want)
        assert st["passa_mfma_launches"] == (1 if force else 0)
        if force:
            assert st["verified_codes"] >= len(Q) * min(k + 1, 64)  # (its survivors were verified exactly)
    ix.set_option("mfma_sub", 0)
    ix.set_option("passa_mfma", 1)
    one = ix.search_batch(k, Q[:1])  # a one-query call: one group of one pair
    assert_same(one, tuple(a[:1] for a in want))
    ix.close()


def test_passa_mfma_short_lists_and_redo(mi, oracle):
    """Lists shorter than k + 1 have no K1-th value: their queries keep T = +inf, sweep 2 marks them and K3f scans their pair
    exactly (k_mfma_redo); an empty nearest list stays empty-handed as in the exact kernels.  Forty cells of very uneven sizes
    (some empty, some with 20 vectors, some with thousands), k = 50."""
    D, m, ks, C, w, k = 64, 8, 256, 40, 6, 50
    rng = np.random.default_rng(5)
    mu = 3.0 * rng.standard_normal((C, D))
    sizes = np.array([0, 0, 3, 20, 49, 50, 51, 52, 100, 400] * 4)
    sizes[-8:] = 3000
    lab = np.repeat(np.arange(C), sizes)
    base = mu[lab] + 0.4 * rng.standard_normal((len(lab), D))
    perm = rng.permutation(len(lab))
    base, lab = base[perm], lab[perm]
    n = len(base)
    ds = D // m
    pq = np.stack([synth.kmeans((mu[lab[:3000]] - base[:3000])[:, s * ds:(s + 1) * ds], ks, iters=2, seed=s) for s in range(m)])
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(mu)
    ix.loadProductQuantizer(pq)
    ix.setW(w)
    ref = oracle_ivfpq(oracle, {"coarse": mu, "pq": pq}, D, m, ks, C, w)
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    # queries next to every centroid (empty and short lists included) and self-queries
    Q = np.concatenate([mu + 0.05 * rng.standard_normal((C, D)), mu + 0.05 * rng.standard_normal((C, D)), base[:150] + 0.01 * rng.standard_normal((150, D))])
    want = ref.search_batch(Q, k)
    ix.set_option("passa_mfma", 1)
    ix.set_profiling(True)
    got = ix.search_batch(k, Q)
    st = ix.get_stats()
    assert_same(got, want)
    assert st["passa_mfma_launches"] == 1
    assert st["mfma_redo_queries"] > 0  # (the short lists' queries went through the redo path)
    ix.close()


@pytest.mark.parametrize("scale", [1e-25, 1e9, 1e140])
def test_passa_mfma_magnitudes(mi, oracle, scale):
    """The fp16 scaling of K3m applies to pass A as well: data far from 1 (and, at 1e140, squares beyond fp32: the rows are marked
    and redone exactly) -- never a dropped neighbour."""
    D, m, C, n, w, k, ks = 64, 8, 6, 12000, 3, 20, 256
    rng = np.random.default_rng(11)
    mu, base, pq = _problem(rng, D, m, C, n)
    mu, base, pq = mu * scale, base * scale, pq * scale
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(mu)
    ix.loadProductQuantizer(pq)
    ix.setW(w)
    ref = oracle_ivfpq(oracle, {"coarse": mu, "pq": pq}, D, m, ks, C, w)
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    Q = np.concatenate([0.5 * (base[:24] + base[100:124]), base[:40] + 0.01 * scale * rng.standard_normal((40, D)), 1e6 * base[50:51]])
    ix.set_option("passa_mfma", 1)
    assert_same(ix.search_batch(k, Q), ref.search_batch(Q, k))
    ix.close()


def test_passa_mfma_default_gate(mi, oracle):
    """With K3q switched off (round 6: it serves every batch from 1.25 queries per list by default), K3ma's own gate (-1) takes pass A
    from 8 queries per list of a long-list index and leaves smaller batches to K3h: 4 cells of ~5000 vectors, 40 queries (10 per
    list: K3ma) and 20 queries (K3h) -- same answers, and the statistics say which ran.  With K3q on, neither runs."""
    D, m, C, n, w, k, ks = 128, 16, 4, 20000, 2, 100, 256
    rng = np.random.default_rng(3)
    mu, base, pq = _problem(rng, D, m, C, n)
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(mu)
    ix.loadProductQuantizer(pq)
    ix.setW(w)
    ref = oracle_ivfpq(oracle, {"coarse": mu, "pq": pq}, D, m, ks, C, w)
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    Q = base[:40] + 0.01 * rng.standard_normal((40, D))
    for passa_q, nq, ran, name in ((0, 40, 1, "K3ma"), (0, 20, 0, "K3h"), (-1, 40, 0, "K3q"), (-1, 20, 0, "K3q")):
        ix.set_option("passa_q", passa_q)
        ix.set_profiling(True)
        got = ix.search_batch(k, Q[:nq])
        st = ix.get_stats()
        assert_same(got, ref.search_batch(Q[:nq], k))
        assert st["passa_mfma_launches"] == ran
        assert ix.get_dispatch()["pass_a"] == name
    ix.close()
