"""Which answers depend on the one thing about the reference that cannot be read off its source tree?

The bounded queue of IVFPQ.java:409/445, PQ.java:291/318 and Linear.java:140/156 is com.aliasi.util.BoundedPriorityQueue, a jar
(LingPipe 4.0.1) that is not in the image.  The oracle restates it under assumption A1: a candidate EQUAL to the
current worst is rejected, and equal entries iterate later-inserted first.  Both choices only matter when two exact fp64
distances tie.  This file runs the oracle under A1 and under the two plausible alternatives (oracle.set_queue_rule: 1 =
accept-equal-to-worst, 2 = earlier-inserted-first among equals) and asserts:

  * ivfpq_perm, pq_small, the hand KATs and a 2048-query sample of cfg3's shape from SURVEY 8d's generator give the SAME ids,
    distance bits and counts under all three rules -- that part of the green parity suite does not lean on A1;
  * ivfpq_small turned out to carry exact ties INSIDE the top-k of 5 of its 24 queries (different vectors of a list with the same
    code): which vectors are returned, and every distance, is the same under all three rules, and the 19 tie-free queries are
    identical id for id; the ORDER of the tied ids is A1's (rule 2 reverses it).  Found by this test in round 6; the fixture is
    kept as it is and counted as tie-bearing from now on;
  * the flagged tie fixture (ivfpq_ties: every vector three times) changes in membership -- the place A1 decides WHO is returned.

It also reports what BASELINE.md section 3 asks to be logged for generated data: the duplicate-code rate of the sample index and
the number of parity queries with a tie at k / anywhere in the top-(k + 1)."""
import json
import os

import numpy as np
import pytest

from tie_census import duplicate_code_rate, tie_census

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RULES = (0, 1, 2)


@pytest.fixture
def rules(oracle):
    yield oracle
    oracle.set_queue_rule(0)  # (process-wide: never leak an alternative rule into another test)


def _ivfpq_of(o, z):
    tr = int(z["transform"])
    ref = o.OracleIndex(o.KIND_IVFPQ, int(z["D"]), int(z["m"]), int(z["ks"]), int(z["C"]), transform=tr, perm=z["perm"] if tr == 2 else None)
    ref.set_coarse(z["coarse"])
    ref.set_pq(z["pq"])
    ref.set_w(int(z["w"]))
    ref.add_vectors(z["base"])
    return ref


def _answers(o, ref, Q, k):
    out = []
    for r in RULES:
        o.set_queue_rule(r)
        assert o.get_queue_rule() == r
        out.append(ref.search_batch(Q, k))
    o.set_queue_rule(0)
    return out


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def test_untied_ivfpq_fixture_does_not_depend_on_the_queue_rule(rules):
    o, z = rules, np.load(os.path.join(GOLD, "ivfpq_perm.npz"))
    a0, a1, a2 = _answers(o, _ivfpq_of(o, z), z["queries"], int(z["k"]))
    assert np.array_equal(a0[0], z["ids"]) and np.array_equal(a0[1], z["dists"])
    assert _same(a0, a1) and _same(a0, a2)


def test_ivfpq_small_depends_on_the_rule_only_in_the_order_of_tied_ids(rules):
    o, z = rules, np.load(os.path.join(GOLD, "ivfpq_small.npz"))
    a0, a1, a2 = _answers(o, _ivfpq_of(o, z), z["queries"], int(z["k"]))
    assert np.array_equal(a0[0], z["ids"]) and np.array_equal(a0[1], z["dists"])
    assert _same(a0, a1)  # (nothing ties at the k / k + 1 boundary: accepting an equal candidate changes nothing)
    assert np.array_equal(a0[1], a2[1]) and np.array_equal(a0[2], a2[2])  # distances and counts: rule-independent
    d = a0[1]
    tied = np.any(d[:, 1:] == d[:, :-1], axis=1)
    assert int(tied.sum()) == 5  # (the queries whose top-k holds two vectors with one code)
    assert np.array_equal(a0[0][~tied], a2[0][~tied])  # tie-free queries: identical id for id
    for q in np.nonzero(tied)[0]:  # tied queries: the same vectors, the tied ones in the other order
        assert sorted(a0[0][q].tolist()) == sorted(a2[0][q].tolist())
        for v in np.unique(d[q]):
            assert sorted(a0[0][q][d[q] == v].tolist()) == sorted(a2[0][q][d[q] == v].tolist())
    assert not np.array_equal(a0[0], a2[0])


def test_untied_pq_fixture_does_not_depend_on_the_queue_rule(rules):
    o, z = rules, np.load(os.path.join(GOLD, "pq_small.npz"))
    ref = o.OracleIndex(o.KIND_PQ, int(z["D"]), int(z["m"]), int(z["ks"]))
    ref.set_pq(z["pq"])
    ref.add_vectors(z["base"])
    a0, a1, a2 = _answers(o, ref, z["queries"], int(z["k"]))
    assert np.array_equal(a0[0], z["ids"]) and np.array_equal(a0[1], z["dists"])
    assert _same(a0, a1) and _same(a0, a2)


def test_the_flagged_tie_fixture_is_where_the_rule_shows(rules):
    o, z = rules, np.load(os.path.join(GOLD, "ivfpq_ties.npz"))
    a0, a1, a2 = _answers(o, _ivfpq_of(o, z), z["queries"], int(z["k"]))
    assert np.array_equal(a0[0], z["ids"])  # (the committed answers are A1's)
    assert np.array_equal(a0[1], a1[1]) and np.array_equal(a0[1], a2[1])  # the distances never depend on the rule ...
    assert not np.array_equal(a0[0], a1[0]) or not np.array_equal(a0[0], a2[0])  # ... the ids of tied entries do


def test_hand_kats_do_not_depend_on_the_queue_rule(rules):
    o = rules
    kat = json.load(open(os.path.join(GOLD, "kat_hand.json")))
    k1, k2 = kat["kat1_pq_adc"], kat["kat2_ivfpq_residual_sign"]
    for r in RULES:
        o.set_queue_rule(r)
        ix = o.OracleIndex(o.KIND_PQ, D=k1["D"], m=k1["m"], ks=k1["ks"])
        ix.set_pq(np.array(k1["pq"], np.float64))
        for v in k1["vectors"]:
            ix.add_vector(np.array(v))
        ids, ds = ix.search(np.array(k1["query"], np.float64), k1["k"])
        assert ids.tolist() == k1["ids"] and ds.tolist() == k1["dists"]
        iv = o.OracleIndex(o.KIND_IVFPQ, D=k2["D"], m=k2["m"], ks=k2["ks"], C_=k2["C"])
        iv.set_coarse(np.array(k2["coarse"], np.float64))
        iv.set_pq(np.array(k2["pq"], np.float64))
        iv.set_w(k2["w"])
        iv.add_vector(np.array(k2["vector"], np.float64))
        ids, ds = iv.search(np.array(k2["query"], np.float64), k2["k"])
        assert ids.tolist() == k2["ids"] and ds.tolist() == k2["dists"]
    o.set_queue_rule(0)


def test_cfg3_shaped_sample_does_not_depend_on_the_queue_rule(rules):
    """SURVEY 8d's generator at cfg3's shape (m = 16 x 256, w = 8, k = 100, ~1000 codes per list), scaled to 64 cells so that the
    CPU oracle builds it in seconds; 2048 self-perturbed queries."""
    import synth

    o = rules
    D, m, ks, C, w, k, n, nq = 128, 16, 256, 64, 8, 100, 64000, 2048
    rng = np.random.default_rng(1234)
    mu = rng.standard_normal((C, D))
    base = mu[rng.integers(0, C, n)] + 0.15 * rng.standard_normal((n, D))
    ds = D // m
    samp = base[:8192]
    cell_s = np.argmin(((samp[:, None, :] - mu[None, :, :]) ** 2).sum(-1), axis=1)
    resid = mu[cell_s] - samp  # (ResidualVectorComputation.java:34: centroid - vector)
    pq = np.stack([synth.kmeans(resid[:, s * ds:(s + 1) * ds], ks, iters=4, seed=s) for s in range(m)])
    ref = o.OracleIndex(o.KIND_IVFPQ, D, m, ks, C)
    ref.set_coarse(mu)
    ref.set_pq(pq)
    ref.set_w(w)
    cells, codes = ref.encode_batch(base)
    ref.add_vectors(base)
    qrng = np.random.default_rng(4321)
    Q = base[qrng.integers(0, n, nq)] + 0.01 * qrng.standard_normal((nq, D))
    a0, a1, a2 = _answers(o, ref, Q, k)
    assert _same(a0, a1) and _same(a0, a2)
    # what BASELINE.md section 3 wants logged for generated data
    ids1, d1, _ = ref.search_batch(Q, k + 1, nthreads=4)
    census = tie_census(d1, k)
    order = np.argsort(cells, kind="stable")
    off = np.concatenate([[0], np.cumsum(np.bincount(cells, minlength=C))])
    rate, _ = duplicate_code_rate(off, codes[order])
    print(f"cfg3-shaped sample: duplicate-code rate {rate:.5f}; {census}")
    assert census["queries_with_tie_at_k"] == 0  # (a tie at the boundary is the only way the rule could have shown above)
