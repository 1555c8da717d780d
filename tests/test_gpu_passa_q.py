"""GPU parity of K3q -- pass A of the IVFADC search decided on packed integer table sums, four queries of a nearest list per block
(csrc/mmidx_scan_q.h; option "passa_q").

Reference loop: the probe-0 iteration of computeKnnIVFADC, IVFPQ.java:414-447 (residual :417 -> :642-648, lookup table :427 ->
:525-538, scan :429-446).  The instance is forced (the default picks it from 1.25 queries per non-empty list of a long-list index) on
small indexes whose shapes walk its code paths: the three sub-quantizer widths it is instantiated for, groups of one to four pairs
and several groups per list, RandomPermutation / RandomRotation, lists shorter than k + 1 (no evidence: everything is a candidate),
empty lists, duplicated vectors (more candidates than the block takes: the query goes to the exact kernel chunk by chunk), data far
from 1 (the fp32 table's guard), k from 1 to 151.  Ids and distance bits are the oracle's; mmidx_get_dispatch proves the instance ran.
tests/test_gpu_parity.py holds the two constructions aimed at single mechanisms (a lane with more candidates than slots; the fp32
table's error bound)."""
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


def _build(mi, oracle, mu, base, pq, D, m, C, w, tr=0, rot=None):
    n = len(base)
    ix = mi.IVFPQ(D, n, False, "", m, 256, tr, C, 512, rot=rot)
    ix.loadCoarseQuantizer(mu)
    ix.loadProductQuantizer(pq)
    ix.setW(w)
    ref = oracle_ivfpq(oracle, {"coarse": mu, "pq": pq}, D, m, 256, C, w, tr=tr, perm=oracle.random_permutation(1, D) if tr == 2 else None, rot=rot)
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    ix.set_option("passa_q", 1)
    return ix, ref


@pytest.mark.parametrize("D,C,n,w,k,tr,dup,nq", [
    (128, 6, 30000, 3, 100, 0, 1, 90),     # <16, 8>, the headline's shape: ~15 queries per list, four groups per list
    (128, 4, 24000, 4, 100, 0, 1, 401),    # ~100 queries per list, a remainder group of one
    (128, 5, 24000, 5, 50, 2, 1, 200),     # RandomPermutation
    (128, 6, 24000, 6, 100, 1, 1, 80),     # RandomRotation (orthogonal): the block rotates its residuals itself
    (64, 4, 16000, 4, 30, 2, 1, 70),       # <16, 4>
    (64, 7, 21000, 7, 1, 0, 1, 300),       # k = 1
    (256, 3, 9000, 3, 20, 0, 1, 64),       # <16, 16>: three blocks per CU
    (256, 4, 12000, 2, 151, 1, 1, 40),     # k + 1 = 152: the largest K1 the instance takes; rotation at D = 256
    (128, 5, 24000, 5, 50, 0, 3, 120),     # every vector three times: ties at the k-th distance, inside the candidate range
])
def test_passa_q_forced(mi, oracle, D, C, n, w, k, tr, dup, nq):
    """Queries: midpoints between vectors, independent Gaussians, self-perturbed vectors, the centroids themselves; then a one-query
    call (a group of one pair: the two-query body) and the same batch with K3q off."""
    m = 16
    rng = np.random.default_rng(3 * D + k + nq)
    mu, base, pq = _problem(rng, D, m, C, n, dup)
    rot = np.linalg.qr(rng.standard_normal((D, D)))[0] if tr == 1 else None
    ix, ref = _build(mi, oracle, mu, base, pq, D, m, C, w, tr, rot)
    nself = nq - 24 - 8 - min(C, 2)
    Q = np.concatenate([0.5 * (base[:24] + base[100:124]), rng.standard_normal((8, D)), base[200:200 + nself] + 0.01 * rng.standard_normal((nself, D)), mu[:2]])
    want = ref.search_batch(Q, k)
    got = ix.search_batch(k, Q)
    assert ix.get_dispatch()["pass_a"] == "K3q"
    assert_same(got, want)
    for sl in (slice(0, 1), slice(5, 7), slice(30, 33)):  # groups of one, two and three pairs at most
        assert_same(ix.search_batch(k, Q[sl]), tuple(a[sl] for a in want))
    ix.set_option("passa_q", 0)
    assert_same(ix.search_batch(k, Q), want)
    assert ix.get_dispatch()["pass_a"] != "K3q"
    ix.close()


def test_passa_q_short_and_empty_lists(mi, oracle):
    """Lists shorter than k + 1 have no K1-th value: every code of such a list is a candidate and the query keeps T = +inf for pass B;
    an empty nearest list leaves the query to pass B alone.  Forty cells of very uneven sizes, k = 50."""
    D, m, C, w, k = 128, 16, 40, 6, 50
    rng = np.random.default_rng(5)
    mu = 3.0 * rng.standard_normal((C, D))
    sizes = np.array([0, 0, 3, 20, 49, 50, 51, 52, 100, 400] * 4)
    sizes[-8:] = 3000
    lab = np.repeat(np.arange(C), sizes)
    base = mu[lab] + 0.4 * rng.standard_normal((len(lab), D))
    perm = rng.permutation(len(lab))
    base, lab = base[perm], lab[perm]
    ds = D // m
    pq = np.stack([synth.kmeans((mu[lab[:3000]] - base[:3000])[:, s * ds:(s + 1) * ds], 256, iters=2, seed=s) for s in range(m)])
    ix, ref = _build(mi, oracle, mu, base, pq, D, m, C, w)
    Q = np.concatenate([mu + 0.05 * rng.standard_normal((C, D)), mu + 0.05 * rng.standard_normal((C, D)), base[:150] + 0.01 * rng.standard_normal((150, D))])
    got = ix.search_batch(k, Q)
    assert ix.get_dispatch()["pass_a"] == "K3q"
    assert_same(got, ref.search_batch(Q, k))
    ix.close()


def test_passa_q_more_candidates_than_the_block_takes(mi, oracle):
    """Every vector forty times: the K1-th smallest integer sum is shared by far more than 192 codes, the block cannot take the
    candidates and hands the query to the exact kernel -- every chunk of its list (a list of 9000 codes is several chunks)."""
    D, m, C, w, k = 128, 16, 3, 3, 100
    rng = np.random.default_rng(17)
    mu, base, pq = _problem(rng, D, m, C, 27000, dup=40)
    ix, ref = _build(mi, oracle, mu, base, pq, D, m, C, w)
    Q = np.concatenate([base[:20] + 1e-3 * rng.standard_normal((20, D)), rng.standard_normal((6, D))])
    got = ix.search_batch(k, Q)
    assert ix.get_dispatch()["pass_a"] == "K3q"
    assert_same(got, ref.search_batch(Q, k))
    ix.close()


@pytest.mark.parametrize("scale", [1e-25, 1e-9, 1e9, 1e140])
def test_passa_q_magnitudes(mi, oracle, scale):
    """The integer table is scaled per query by its own mean distance, so data far from 1 is served as long as fp32 can hold the
    table's inputs (1e-9, 1e9); beyond that (1e-25: squares under fp32's normal range; 1e140: beyond its largest number) the block's
    guard hands the query to the exact kernel.  Never a dropped neighbour."""
    D, m, C, n, w, k = 128, 16, 6, 18000, 3, 20
    rng = np.random.default_rng(11)
    mu, base, pq = _problem(rng, D, m, C, n)
    mu, base, pq = mu * scale, base * scale, pq * scale
    ix, ref = _build(mi, oracle, mu, base, pq, D, m, C, w)
    Q = np.concatenate([0.5 * (base[:24] + base[100:124]), base[:40] + 0.01 * scale * rng.standard_normal((40, D)), 1e6 * base[50:51]])
    got = ix.search_batch(k, Q)
    assert ix.get_dispatch()["pass_a"] == "K3q"
    assert_same(got, ref.search_batch(Q, k))
    ix.close()
