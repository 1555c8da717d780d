"""GPU parity of k_scan_split -- the exact scan for a lookup table of twice the LDS (m = 128 byte codes: 128 x 256 x 8 B = 256 KiB),
taken in two sweeps with half the table in LDS each (csrc/mmidx_kernels.h; option "no_split_table" = 1 is the A/B switch back to the
table-in-global kernels).

Reference loop: IVFPQ.java:429-446 / PQ.java:281-300 with the table of IVFPQ.java:525-538 -- the sum over the 128 sub-quantizers in
ascending order; the split continues the first half's partial sum, so ids AND distance bits are the oracle's.
"""
import numpy as np
import pytest

import synth
from test_gpu_parity import assert_same, mi, oracle_ivfpq  # noqa: F401  (mi: the module fixture)

pytestmark = pytest.mark.gpu


def _codebook(rng, resid, m, ds, ks=256):
    return np.stack([synth.kmeans(resid[:, s * ds:(s + 1) * ds], ks, iters=1, seed=s) for s in range(m)])


@pytest.mark.parametrize("D,C,n,w,k,tr,dup", [
    (1024, 6, 6000, 4, 10, 0, 1),    # dsub 8 (Example.java's pq_1024_128x8): lists of ~1000 codes = two segments, pruning on the way
    (1024, 3, 9000, 3, 100, 2, 3),   # RandomPermutation; every vector three times: ties at the k-th distance (the replay uses the table in global)
    (512, 5, 5000, 5, 1, 0, 1),      # dsub 4, k = 1
    (256, 4, 4000, 2, 30, 1, 1),     # dsub 2 (the generic table build), RandomRotation: pass B through the exact kernels as well
    (2048, 2, 1500, 2, 20, 0, 1),    # dsub 16
])
def test_split_table_ivfpq(mi, oracle, D, C, n, w, k, tr, dup):
    m, ks = 128, 256
    rng = np.random.default_rng(D + k)
    mu = 0.5 * rng.standard_normal((C, D))
    base = mu[rng.integers(0, C, n // dup)] + 0.3 * rng.standard_normal((n // dup, D))
    base = np.concatenate([base] * dup)[rng.permutation((n // dup) * dup)]
    n = len(base)
    pq = _codebook(rng, mu[rng.integers(0, C, 1500)] - base[:1500], m, D // m)
    rot = np.linalg.qr(rng.standard_normal((D, D)))[0] if tr == 1 else None
    ix = mi.IVFPQ(D, n, False, "", m, ks, tr, C, 512, rot=rot)
    ix.loadCoarseQuantizer(mu)
    ix.loadProductQuantizer(pq)
    ix.setW(w)
    ref = oracle_ivfpq(oracle, {"coarse": mu, "pq": pq}, D, m, ks, C, w, tr=tr, perm=oracle.random_permutation(1, D) if tr == 2 else None, rot=rot)
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    Q = np.concatenate([base[:20] + 0.01 * rng.standard_normal((20, D)), 0.5 * (base[30:40] + base[50:60]), rng.standard_normal((4, D))])
    want = ref.search_batch(Q, k)
    assert_same(ix.search_batch(k, Q), want)
    assert ix.get_dispatch()["pass_a"] == "K3(table in two halves)"
    assert_same(ix.search_batch(k, Q[:1]), tuple(a[:1] for a in want))  # the reference's own call shape
    ix.set_option("no_mfma", 1)  # every probed list through the exact kernels (pass B as well)
    assert_same(ix.search_batch(k, Q), want)
    ix.set_option("no_split_table", 1)
    assert_same(ix.search_batch(k, Q), want)
    assert ix.get_dispatch()["pass_a"] == "K3(table in global scratch)"
    ix.close()


def test_split_table_flat_pq(mi, oracle):
    """Flat PQ (PQ.java:281-300): one list in chunks; a chunk's partial sums fit the block's table slot up to 32768 codes (batches
    under 512 queries), beyond that the table-in-global kernels keep the shape"""
    D, m, ks, n, k = 512, 128, 256, 40000, 10
    rng = np.random.default_rng(8)
    base = rng.standard_normal((n, D))
    pq = _codebook(rng, base[:1500], m, D // m)
    ix = mi.PQ(D, n, False, "", m, ks, 0, 512)
    ix.loadProductQuantizer(pq)
    ref = oracle.OracleIndex(oracle.KIND_PQ, D, m, ks)
    ref.set_pq(pq)
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    Q = base[:12] + 0.01 * rng.standard_normal((12, D))
    assert_same(ix.search_batch(k, Q), ref.search_batch(Q, k))
    ix.close()
