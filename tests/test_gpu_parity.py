"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bar: neighbour ids bit-exact, distances bit-equal (the promise to callers is 1e-5; the kernels
keep the reference's fp64 operation order, so equality is the test).  Fixtures are tie-free except
the explicitly flagged tie fixtures, which pin the bounded-queue rule (assumption A1).
"""
import importlib

import ctypes

import os

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

TOL = 1e-5  # stated tolerance (north star); asserted bit-equal below, TOL is the documented bound


@pytest.fixture(scope="module")
def mi():
    # torch (used by one test only to hold device buffers) bundles its own HIP runtime; when it is
    # going to be used in this process it has to initialise before libmmidx_hip.so does
    try:
        import torch

        torch.cuda.init()
    except Exception:
        pass
    m = importlib.import_module("multimedia-indexing_amd")
    if m.lib().mmidx_device_count() < 1:
        pytest.fail("libmmidx_hip.so found no HIP device: GPU tests must run the native path")
    return m


def oracle_ivfpq(o, p, D, m, ks, C, w, tr=0, perm=None, rot=None):
    ref = o.OracleIndex(o.KIND_IVFPQ, D, m, ks, C, transform=tr, perm=perm, rot=rot)
    ref.set_coarse(p["coarse"])
    ref.set_pq(p["pq"])
    ref.set_w(w)
    return ref


def assert_same(res, ref, bit_exact=True):
    iids, dists, counts = res
    rid, rd, rc = ref
    assert np.array_equal(counts, rc)
    assert np.array_equal(iids, rid)
    if bit_exact:
        assert np.array_equal(dists, rd)
    else:
        fin = np.isfinite(rd)
        assert np.array_equal(np.isfinite(dists), fin)
        assert np.max(np.abs(dists[fin] - rd[fin]), initial=0.0) <= TOL


@pytest.mark.parametrize("D,C,m,ks,n,w,k", [
    (32, 16, 8, 256, 5000, 4, 10),     # dsub 4
    (128, 64, 16, 256, 20000, 8, 100), # cfg3 shape, scaled down
    (64, 32, 8, 64, 3000, 32, 5),      # w == C, ks < 256
    (24, 8, 6, 256, 2000, 3, 50),      # m not a template instance (generic kernel)
])
def test_ivfpq_encode_and_search(mi, oracle, D, C, m, ks, n, w, k):
    p = synth.make_ivfpq_problem(n=n, D=D, C=C, m=m, ks=ks, nq=64, seed=D + C)
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    assert ix.getW() == int(C * 0.1)  # IVFPQ.java:188
    ix.setW(w)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    # encode parity (IVFPQ.java:309-335)
    cells, codes = ix.encode(p["base"][:1500])
    rcell, rcode = ref.encode_batch(p["base"][:1500])
    assert np.array_equal(cells, rcell)
    assert np.array_equal(codes.astype(np.int32) + 128, rcode)
    # index through indexVector (batched) and search
    assert ix.indexVectors([str(i) for i in range(n)], p["base"]) == n
    ref.add_vectors(p["base"])
    assert np.array_equal(ix.listSizes(), ref.list_sizes())
    assert_same(ix.search_batch(k, p["queries"]), ref.search_batch(p["queries"], k))
    # single-query surface: Answer with external ids
    a = ix.computeNearestNeighbors(k, p["queries"][0])
    rid, rd = ref.search(p["queries"][0], k)
    assert a.getIds() == [str(i) for i in rid] and np.array_equal(a.getDistances(), rd)
    ix.close()


@pytest.mark.parametrize("tr", [1, 2])
def test_ivfpq_transforms(mi, oracle, tr):
    D, C, m, ks, n, w, k = 32, 16, 8, 256, 4000, 5, 20
    p = synth.make_ivfpq_problem(n=n, D=D, C=C, m=m, ks=ks, nq=32, seed=77 + tr)
    rot = np.linalg.qr(np.random.default_rng(5).standard_normal((D, D)))[0] if tr == 1 else None
    ix = mi.IVFPQ(D, n, False, "", m, ks, tr, C, 512, rot=rot)
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    ix.setW(w)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w, tr=tr, rot=rot)  # perm=None -> RandomPermutation(1, D)
    ix.indexVectors([str(i) for i in range(n)], p["base"])
    ref.add_vectors(p["base"])
    assert np.array_equal(ix.listSizes(), ref.list_sizes())
    # rotation: same sequential order on both sides (EJML order itself is assumption A2)
    assert_same(ix.search_batch(k, p["queries"]), ref.search_batch(p["queries"], k))
    ix.close()


@pytest.mark.parametrize("D,m,ks,n,k", [(128, 8, 256, 40000, 100), (32, 4, 256, 70000, 10), (16, 16, 16, 1000, 3)])
def test_pq_adc(mi, oracle, D, m, ks, n, k):
    o = oracle
    p = synth.make_pq_problem(n=min(n, 20000), D=D, m=m, ks=ks, nq=16, seed=n)
    rng = np.random.default_rng(1)
    base = rng.standard_normal((n, D))
    ix = mi.PQ(D, n, False, "", m, ks, 0, 512)
    ix.loadProductQuantizer(p["pq"])
    ref = o.OracleIndex(o.KIND_PQ, D, m, ks)
    ref.set_pq(p["pq"])
    cells, codes = ix.encode(base)
    assert np.all(cells == -1)
    sample = rng.choice(n, size=800, replace=False)
    _, rcode = ref.encode_batch(base[sample])
    assert np.array_equal(codes[sample].astype(np.int32) + 128, rcode)
    # bulk-load path (loadIndexInMemory, PQ.java:436-483) with GPU-produced codes
    ix.loadIndex(codes)
    ref.add_codes(np.arange(n), None, codes.astype(np.int32) + 128)
    assert ix.size() == n
    assert_same(ix.search_batch(k, p["queries"]), ref.search_batch(p["queries"], k))
    ix.close()


def test_short_codes(mi, oracle):
    """numProductCentroids > 256 -> int16 codes (PQ.transformToShort, PQ.java:544-550)."""
    D, C, m, ks, n, w, k = 32, 8, 8, 512, 6000, 3, 10
    p = synth.make_ivfpq_problem(n=n, D=D, C=C, m=m, ks=ks, nq=16, seed=4)
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    ix.setW(w)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    cells, codes = ix.encode(p["base"][:500])
    rcell, rcode = ref.encode_batch(p["base"][:500])
    assert codes.dtype == np.int16 and np.array_equal(codes, rcode) and np.array_equal(cells, rcell)
    ix.indexVectors([str(i) for i in range(n)], p["base"])
    ref.add_vectors(p["base"])
    assert_same(ix.search_batch(k, p["queries"]), ref.search_batch(p["queries"], k))
    with pytest.raises(mi.MmidxError) as ei:
        ix.indexPQCode("x", 0, np.zeros(m, np.int8))
    assert ei.value.status == 4  # IVFPQ.java:358-361
    ix.close()


@pytest.mark.parametrize("kind", ["ivfpq", "ivfpq_ties", "pq"])
def test_lookup_table_larger_than_lds(mi, oracle, kind):
    """m x ks x 8 bytes beyond the 160 KiB LDS (here 32 x 1024 doubles = 256 KiB, short codes): the reference has no such limit
    (IVFPQ.java:525-538 allocates double[m][ks]); the table then lives in global scratch (k_scan / k_tie_resolve with GLUT).
    Same ids and distance bits as the oracle, straddling ties (every vector three times) included."""
    D, C, m, ks, n, w, k = 64, 6, 32, 1024, 4000, 4, 10
    if kind == "pq":
        p = synth.make_pq_problem(n=n, D=D, m=m, ks=ks, nq=8, seed=11)
        ix = mi.PQ(D, 3 * n, False, "", m, ks, 0, 512)
        ref = oracle.OracleIndex(oracle.KIND_PQ, D, m, ks)
    else:
        p = synth.make_ivfpq_problem(n=n, D=D, C=C, m=m, ks=ks, nq=8, seed=11)
        ix = mi.IVFPQ(D, 3 * n, False, "", m, ks, 0, C, 512)
        ix.loadCoarseQuantizer(p["coarse"])
        ix.setW(w)
        ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    ix.loadProductQuantizer(p["pq"])
    if kind == "pq":
        ref.set_pq(p["pq"])
    base = p["base"]
    if kind == "ivfpq_ties":
        base = np.concatenate([base] * 3)[np.random.default_rng(2).permutation(3 * n)]
    ix.indexVectors([str(i) for i in range(len(base))], base)
    ref.add_vectors(base)
    got, want = ix.search_batch(k, p["queries"]), ref.search_batch(p["queries"], k)
    assert_same(got, want)
    # one query per call -- the reference's own call shape -- and three: pass B's grid is the pair count rounded up to eight
    # blocks, which the scratch slots must cover (nq = 1 with w = 4 used to fail with "scratch too small")
    if kind != "pq":
        for nq_, w_ in ((1, 4), (3, 2), (1, 6), (2, 5)):
            ix.setW(w_)
            ref.set_w(w_)
            assert_same(ix.search_batch(k, p["queries"][:nq_]), ref.search_batch(p["queries"][:nq_], k))
        ix.setW(w)
        ref.set_w(w)
    else:
        assert_same(ix.search_batch(k, p["queries"][:1]), ref.search_batch(p["queries"][:1], k))
    if kind == "ivfpq_ties":  # the comparison covers the replay: some query's k-th and (k+1)-th distances tie
        w1 = ref.search_batch(p["queries"], k + 1)
        assert any(w1[2][i] > k and w1[1][i, k - 1] == w1[1][i, k] for i in range(len(p["queries"])))
    ix.close()


@pytest.mark.parametrize("k", [1500, 4095])
def test_large_k(mi, oracle, k):
    """k beyond 1023 (the reference's queue is unbounded, IVFPQ.java:409): candidate and merge buffers grow with k up to 4095;
    IVFPQ (w = C so that more than k candidates exist), flat PQ, and the sharded merge K5."""
    import torch

    sh = importlib.import_module("multimedia-indexing_amd.sharded")
    D, C, m, ks, n, w = 16, 6, 8, 256, 9000, 6
    p = synth.make_ivfpq_problem(n=n, D=D, C=C, m=m, ks=ks, nq=6, seed=70)
    rng = np.random.default_rng(3)
    base = rng.standard_normal((n, D)) * 2.0  # (spread out: distinct codes, few exact ties)
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    ix.setW(w)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    assert_same(ix.search_batch(k, p["queries"]), ref.search_batch(p["queries"], k))
    cells, codes = ix.encode(base)
    ix.close()
    pq = mi.PQ(D, n, False, "", m, ks, 0, 512)
    pq.loadProductQuantizer(p["pq"])
    rpq = oracle.OracleIndex(oracle.KIND_PQ, D, m, ks)
    rpq.set_pq(p["pq"])
    pq.indexVectors([str(i) for i in range(n)], base)
    rpq.add_vectors(base)
    assert_same(pq.search_batch(k, p["queries"]), rpq.search_batch(p["queries"], k))
    pq.close()
    with pytest.raises(mi.MmidxError):
        bad = mi.PQ(D, 10, False, "", m, ks, 0, 512)
        bad.loadProductQuantizer(p["pq"])
        bad.indexVectors(["0"], base[:1])
        bad.search_batch(4096, p["queries"])
    # two virtual shards: partial lists of k + 1 entries each, merged by K5 with the larger buffer
    shards = []
    for r in range(2):
        sx = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
        sx.loadCoarseQuantizer(p["coarse"])
        sx.loadProductQuantizer(p["pq"])
        sx.setW(w)
        own = np.nonzero(sh.owner_of_cell(cells, 2) == r)[0]
        sx.loadIndex(own.astype(np.int32), cells[own], codes[own])
        shards.append(sx)
    Q = torch.tensor(p["queries"], dtype=torch.float64, device="cuda")
    engines = [sh.HipShardEngine(sx._h, D, w, 0) for sx in shards]
    probe, _ = engines[0].coarse(Q)
    parts = [e.search_partial(k, Q, probe) for e in engines]
    iid, dd, cnt, _ = engines[0].merge(k, torch.stack([x[0] for x in parts]), torch.stack([x[1] for x in parts]), torch.stack([x[2] for x in parts]))
    torch.cuda.synchronize()
    rid, rd, rc = ref.search_batch(p["queries"], k)
    _, rd1, rc1 = ref.search_batch(p["queries"], k + 1)
    for qi in range(len(rc)):
        if rc1[qi] > k and rd1[qi, k - 1] == rd1[qi, k]:
            continue  # (a straddling tie is the tie replay's business: test_virtual_shards_straddling_ties)
        assert np.array_equal(iid.cpu().numpy()[qi], rid[qi]) and np.array_equal(dd.cpu().numpy()[qi], rd[qi])
    for sx in shards:
        sx.close()


@pytest.mark.parametrize("k", [3600, 3839])
def test_large_k_with_the_table_in_global_scratch(mi, oracle, k):
    """m = 32 x 256 byte codes and k in 3584 .. 3839: the table (64 KiB) plus the candidate buffer no longer fit the LDS, the
    generic two-codes-per-thread kernels run with the table in global scratch, and pass A must give them their full buffer
    (k + 1 + 512 entries -- ADVICE r2: the one-code-per-thread shrink of pass A left 4096 < k + 1 + 512)."""
    D, C, m, ks, n, w = 64, 4, 32, 256, 12000, 3
    p = synth.make_ivfpq_problem(n=4000, D=D, C=C, m=m, ks=ks, nq=5, seed=91, iters=2)
    base = np.random.default_rng(4).standard_normal((n, D)) * 1.5
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    ix.setW(w)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    assert_same(ix.search_batch(k, p["queries"]), ref.search_batch(p["queries"], k))
    assert_same(ix.search_batch(k, p["queries"][:1]), ref.search_batch(p["queries"][:1], k))
    ix.close()


@pytest.mark.parametrize("ks,tr", [(256, 0), (256, 2), (256, 1), (512, 0)])
def test_per_id_utilities(mi, oracle, ks, tr):
    """computeDistanceIVFADC (IVFPQ.java:464-497), getPQCodeByte / getPQCodeShort (:801-856), getInvertedListId (:865-880)
    against the oracle's restatement; the distance is also the one a search reports for that candidate."""
    D, C, m, n, w, k = 32, 8, 8, 3000, 8, 50
    p = synth.make_ivfpq_problem(n=n, D=D, C=C, m=m, ks=ks, nq=8, seed=40 + ks + tr)
    rot = np.linalg.qr(np.random.default_rng(6).standard_normal((D, D)))[0] if tr == 1 else None
    ix = mi.IVFPQ(D, n + 1, False, "", m, ks, tr, C, 512, rot=rot)
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    ix.setW(w)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w, tr=tr, rot=rot)
    ids = [f"v{i}" for i in range(n)]
    ix.indexVectors(ids, p["base"])
    ref.add_vectors(p["base"])
    rng = np.random.default_rng(1)
    for iid in rng.choice(n, 40, replace=False):
        cell, code = ref.get_record(int(iid))
        assert ix.getInvertedListId(ids[iid]) == cell
        got = ix.getPQCodeByte(ids[iid]) if ks <= 256 else ix.getPQCodeShort(ids[iid])
        assert got.dtype == (np.int8 if ks <= 256 else np.int16) and np.array_equal(got.astype(np.int32), code)
        q = p["queries"][int(iid) % 8]
        d = ix.computeDistanceIVFADC(q, ids[iid])
        rd = ref.distance(q, int(iid))
        if tr == 1:
            assert abs(d - rd) <= 1e-9 * max(1.0, abs(rd))  # (rotation: summation order of the matrix product, assumption A2)
        else:
            assert d == rd
    # the same number a search reports (w = C: every vector is a candidate)
    iids, dists, counts = ix.search_batch(k, p["queries"][:2])
    for qi in range(2):
        for j in range(0, k, 7):
            assert ix.computeDistanceIVFADC(p["queries"][qi], ids[iids[qi, j]]) == dists[qi, j]
    # error behaviour: unknown id, wrong variant (IVFPQ.java:803-809, :835-841)
    with pytest.raises(mi.MmidxError, match="Id does not exist!"):
        ix.getInvertedListId("nope")
    with pytest.raises(mi.MmidxError, match="Id does not exist!"):
        ix.computeDistanceIVFADC(p["queries"][0], "nope")
    with pytest.raises(mi.MmidxError, match="short variant"):
        (ix.getPQCodeShort if ks <= 256 else ix.getPQCodeByte)(ids[0])
    # batched C entry points + a record added through indexPQCode after the first lookups (the id map is rebuilt)
    L = mi.lib()
    if ks <= 256:
        assert ix.indexPQCode("late", 3, np.full(m, -128, np.int8))
        assert ix.getInvertedListId("late") == 3 and np.array_equal(ix.getPQCodeByte("late"), np.full(m, -128, np.int8))
    q3 = np.ascontiguousarray(p["queries"][:3])
    want = np.array([5, 17, 2999], np.int32)
    out = np.zeros(3)
    assert L.mmidx_distance(ix._h, 3, q3.ctypes.data, want.ctypes.data, out.ctypes.data) == 0
    for i in range(3):
        rd = ref.distance(q3[i], int(want[i]))
        assert out[i] == rd or (tr == 1 and abs(out[i] - rd) <= 1e-9 * max(1.0, abs(rd)))
    bad = np.array([5, n + 100], np.int32)
    assert L.mmidx_get_codes(ix._h, 2, bad.ctypes.data, None, None) == 6 and b"Id does not exist!" in L.mmidx_last_error()
    dims = [ctypes.c_int() for _ in range(5)]
    assert L.mmidx_get_dims(ix._h, *[ctypes.byref(x) for x in dims]) == 0
    assert [x.value for x in dims] == [D, m, ks, C, 1 if ks <= 256 else 2]
    ix.close()


def test_add_codes_validation_keeps_the_index_usable(mi, oracle):
    """a record with a list id outside [0, C) or a code value >= ks must be rejected at add time, and the handle
    must stay usable afterwards (the reference would throw ArrayIndexOutOfBounds at invertedLists[listId], IVFPQ.java:371)."""
    D, C, m, ks, n, w, k = 16, 4, 4, 16, 400, 4, 5
    p = synth.make_ivfpq_problem(n=n, D=D, C=C, m=m, ks=ks, nq=4, seed=9)
    ix = mi.IVFPQ(D, n + 8, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    ix.setW(w)
    ix.indexVectors([str(i) for i in range(n)], p["base"])
    before = ix.search_batch(k, p["queries"])
    with pytest.raises(mi.MmidxError):
        ix.loadIndex(np.array([n], np.int32), np.array([C + 3], np.int32), np.full((1, m), -128, np.int8))   # bad list id
    with pytest.raises(mi.MmidxError):
        ix.loadIndex(np.array([n], np.int32), np.array([0], np.int32), np.full((1, m), -128 + ks, np.int8))  # code == ks
    assert ix.size() == n
    after = ix.search_batch(k, p["queries"])
    assert all(np.array_equal(a, b) for a, b in zip(before, after))
    ix.loadIndex(np.array([n], np.int32), np.array([1], np.int32), np.full((1, m), -128, np.int8))
    assert ix.size() == n + 1 and int(ix.listSizes().sum()) == n + 1
    ix.close()


def test_index_pq_code_roundtrip_and_incremental(mi, oracle):
    """encode -> indexPQCode == indexVector (IVFPQ.java:357-386), adds interleaved with searches."""
    D, C, m, ks, n, w, k = 32, 8, 8, 256, 1200, 8, 7
    p = synth.make_ivfpq_problem(n=n, D=D, C=C, m=m, ks=ks, nq=12, seed=8)
    a = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    b = mi.IVFPQ(D, n + 10, False, "", m, ks, 0, C, 512)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    for ix in (a, b):
        ix.loadCoarseQuantizer(p["coarse"])
        ix.loadProductQuantizer(p["pq"])
        ix.setW(w)
    cells, codes = a.encode(p["base"])
    for lo, hi in ((0, 300), (300, 301), (301, 1200)):
        a.indexVectors([str(i) for i in range(lo, hi)], p["base"][lo:hi])
        for i in range(lo, hi):
            assert b.indexPQCode(str(i), int(cells[i]), codes[i])
            ref.add_vector(p["base"][i])
        r = ref.search_batch(p["queries"], k)
        assert_same(a.search_batch(k, p["queries"]), r)
        assert_same(b.search_batch(k, p["queries"]), r)
    # duplicate id / capacity -> False, nothing indexed (ASS:232-240)
    assert a.indexVector("0", p["base"][0]) is False
    assert a.indexVector("new", p["base"][0]) is False  # capacity n reached
    assert a.size() == n
    with pytest.raises(mi.MmidxError) as ei:
        b.indexVector("zz", np.zeros(D + 1))
    assert ei.value.status == 2
    a.close()
    b.close()


def test_edge_cases(mi, oracle):
    D, C, m, ks, w = 16, 8, 4, 256, 3
    p = synth.make_ivfpq_problem(n=400, D=D, C=C, m=m, ks=ks, nq=6, seed=2)
    ix = mi.IVFPQ(D, 1000, False, "", m, ks, 0, C, 512)
    # quantizers not loaded -> NOT_READY (NullPointerException in the reference)
    with pytest.raises(mi.MmidxError) as ei:
        ix.search_batch(1, p["queries"])
    assert ei.value.status == 7
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    ix.setW(w)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    # empty index: every answer is empty
    iids, dists, counts = ix.search_batch(5, p["queries"])
    assert np.all(counts == 0) and np.all(iids == -1)
    # ragged: only a few vectors, most probed lists empty, k larger than the candidates
    ix.indexVectors([str(i) for i in range(5)], p["base"][:5])
    ref.add_vectors(p["base"][:5])
    assert_same(ix.search_batch(50, p["queries"]), ref.search_batch(p["queries"], 50))
    ix.indexVectors([str(i) for i in range(5, 400)], p["base"][5:400])
    ref.add_vectors(p["base"][5:400])
    for k in (1, 399, 1023):
        assert_same(ix.search_batch(k, p["queries"]), ref.search_batch(p["queries"], k))
    # invalid k / w (LingPipe ctor rejects 0; w > C is an NPE at IVFPQ.java:597-599)
    for bad_k in (0, 4096):
        with pytest.raises(mi.MmidxError) as ei:
            ix.search_batch(bad_k, p["queries"])
        assert ei.value.status == 6
    for bad_w in (0, C + 1):
        ix.setW(bad_w)
        with pytest.raises(mi.MmidxError) as ei:
            ix.search_batch(1, p["queries"])
        assert ei.value.status == 6
    ix.setW(C)
    ref.set_w(C)
    assert_same(ix.search_batch(10, p["queries"]), ref.search_batch(p["queries"], 10))
    # zero queries
    iids, dists, counts = ix.search_batch(3, np.zeros((0, D)))
    assert iids.shape == (0, 3)
    ix.close()


def test_flagged_ties(mi, oracle):
    """FLAGGED tie fixture: duplicate codes create exact distance ties straddling k; the GPU path
    must reproduce the bounded queue's order (assumption A1, replayed by the oracle)."""
    D, C, m, ks, w = 16, 4, 4, 256, 4
    p = synth.make_ivfpq_problem(n=600, D=D, C=C, m=m, ks=ks, nq=8, seed=12)
    base = np.concatenate([p["base"][:200]] * 3 + [p["base"][200:260]])  # every code three times
    rng = np.random.default_rng(0)
    base = base[rng.permutation(base.shape[0])]
    ix = mi.IVFPQ(D, len(base), False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    ix.setW(w)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    ix.indexVectors([str(i) for i in range(len(base))], base)
    ref.add_vectors(base)
    ix.set_profiling(True)
    total_fallbacks = 0
    for k in (1, 2, 4, 5, 10, 31):
        res = ix.search_batch(k, p["queries"])
        total_fallbacks += ix.get_stats()["tie_fallbacks"]
        assert_same(res, ref.search_batch(p["queries"], k))
    assert total_fallbacks > 0, "fixture did not exercise the tie replay"
    ix.close()
    # flat PQ variant: heavy ties (tight clusters -> few distinct codes)
    base, _ = synth.mixture(3000, 16, 6, sigma=0.01, seed=3)
    pp = synth.make_pq_problem(n=2000, D=16, m=4, ks=16, nq=8, seed=3)
    pq = mi.PQ(16, 3000, False, "", 4, 16, 0, 512)
    pq.loadProductQuantizer(pp["pq"])
    rpq = oracle.OracleIndex(oracle.KIND_PQ, 16, 4, 16)
    rpq.set_pq(pp["pq"])
    pq.indexVectors([str(i) for i in range(3000)], base)
    rpq.add_vectors(base)
    q = base[:8] + 0.001
    for k in (1, 7, 100):
        assert_same(pq.search_batch(k, q), rpq.search_batch(q, k))
    pq.close()


def test_large_batch_properties(mi):
    """Size-independent properties at a larger size (no oracle): sortedness, idempotence,
    self-query hits itself, k-prefix consistency."""
    D, C, m, ks, n, w = 64, 128, 8, 256, 200000, 8
    rng = np.random.default_rng(6)
    base, mu = synth.mixture(n, D, C, seed=6)
    coarse = mu
    resid_pq = np.stack([synth.kmeans((0.15 * rng.standard_normal((4000, D)))[:, s * 8:(s + 1) * 8], ks, iters=3, seed=s)
                         for s in range(m)])
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(coarse)
    ix.loadProductQuantizer(resid_pq)
    ix.setW(w)
    ix.indexVectors(list(range(n)), base)
    sizes = ix.listSizes()
    assert sizes.sum() == n
    qi = rng.choice(n, 2048, replace=False)
    Q = base[qi] + 0.01 * rng.standard_normal((2048, D))
    i1, d1, c1 = ix.search_batch(100, Q)
    i2, d2, c2 = ix.search_batch(100, Q)
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)  # deterministic despite atomics
    assert np.all(np.diff(d1, axis=1) >= 0)
    i10, d10, _ = ix.search_batch(10, Q)
    assert np.array_equal(d10, d1[:, :10])  # k-prefix (tie-free data)
    assert np.array_equal(i10, i1[:, :10])
    recall1 = np.mean(i1[:, 0] == qi)
    assert recall1 > 0.8, recall1
    ix.close()


def test_pass_b_in_one_launch_after_an_empty_one(mi, oracle):
    """When the call before kept at most 64 (query, probe) pairs behind the coarse bound, the next call's pass B is K3f's looping kernel
    alone -- one launch instead of K3m's six over what is usually nothing (launch_scan_grouped; option "passb_small").  The guess can be
    wrong: well separated cells and self-queries (every far pair pruned), THEN queries between the cells (every far pair kept) -- that
    call is served by the one kernel, exactly; the call after it sees the real count and goes back to K3m.  The oracle's answers every
    time, with the option and without."""
    D, C, m, ks, n, w, k = 64, 40, 8, 256, 30000, 8, 20
    rng = np.random.default_rng(23)
    mu = 6.0 * rng.standard_normal((C, D))
    lab = rng.integers(0, C, n)
    base = mu[lab] + 0.3 * rng.standard_normal((n, D))
    ds = D // m
    pq = np.stack([synth.kmeans((mu[lab[:3000]] - base[:3000])[:, s * ds:(s + 1) * ds], ks, iters=2, seed=s) for s in range(m)])  # (the true residuals: a small Rmax)
    ref = oracle_ivfpq(oracle, {"coarse": mu, "pq": pq}, D, m, ks, C, w)
    ref.add_vectors(base)
    Qself = base[:200] + 0.001 * rng.standard_normal((200, D))
    Qmid = np.concatenate([0.5 * (mu[:20] + mu[20:40]), rng.standard_normal((180, D))])  # far from every cell: nothing can be pruned; a batch like Qself (the hint is trusted for a like call only)
    want_self, want_mid = ref.search_batch(Qself, k), ref.search_batch(Qmid, k)
    for small in (1, 0):
        ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
        ix.loadCoarseQuantizer(mu)
        ix.loadProductQuantizer(pq)
        ix.setW(w)
        ix.set_option("passb_small", small)
        ix.indexVectors([str(i) for i in range(n)], base)
        ix.set_profiling(True)
        assert_same(ix.search_batch(k, Qself), want_self)   # (first call: no figure yet, K3m)
        assert ix.get_stats()["passb_items_last"] <= 64
        assert_same(ix.search_batch(k, Qself), want_self)   # the call before kept nothing: one launch
        assert_same(ix.search_batch(k, Qmid), want_mid)     # ... and so does this one, with every pair kept
        st = ix.get_stats()
        assert st["passb_items_last"] > 64
        if small:
            assert st["mfma_survivors"] == 0                # (K3m has not run since the first call)
        assert_same(ix.search_batch(k, Qmid), want_mid)     # the real count is known: K3m again
        assert ix.get_stats()["mfma_survivors"] > 0
        # a call of another size behind an empty one: the figure says nothing about it -- K3m, not the looping kernel
        assert_same(ix.search_batch(k, Qself), want_self)
        assert_same(ix.search_batch(k, Qself), want_self)
        ix.get_stats()
        part = tuple(a[:50] for a in want_mid)
        assert_same(ix.search_batch(k, Qmid[:50]), part)
        assert ix.get_stats()["mfma_survivors"] > 0
        ix.close()


def _coarse_cells(mi, ix, Q):
    """computeNearestCoarseIndices through the device entry point (torch only holds the buffers)."""
    import torch

    nat = importlib.import_module("multimedia-indexing_amd._native")
    dQ = torch.tensor(np.ascontiguousarray(Q), dtype=torch.float64, device="cuda")
    w = ix.getW()
    cells = torch.empty(dQ.shape[0], w, dtype=torch.int32, device="cuda")
    nat.check(mi.lib().mmidx_coarse_device(ix._h, dQ.shape[0], dQ.data_ptr(), cells.data_ptr(), None, None))
    torch.cuda.synchronize()
    return cells.cpu().numpy()


@pytest.mark.parametrize("C", [12, 300, 2500])
def test_coarse_topw_with_duplicate_centroids(mi, oracle, C):
    """FLAGGED tie fixture for the coarse stage: duplicated centroids give exactly equal
    distances; the probe order / choice must follow the bounded queue (IVFPQ.java:576-600)."""
    D, m, ks = 8, 2, 16
    rng = np.random.default_rng(C)
    coarse = rng.standard_normal((C, D))
    dup = rng.integers(0, C // 3, size=C // 2)          # half of the rows duplicate an earlier row
    coarse[C // 2:C // 2 + len(dup)] = coarse[dup]
    pq = rng.standard_normal((m, ks, D // m))
    ix = mi.IVFPQ(D, 10, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(coarse)
    ix.loadProductQuantizer(pq)
    ref = oracle.OracleIndex(oracle.KIND_IVFPQ, D, m, ks, C)
    ref.set_coarse(coarse)
    ref.set_pq(pq)
    Q = np.concatenate([rng.standard_normal((24, D)), coarse[dup[:8]] + 1e-3])
    for w in (1, 2, 3, 5, C // 2, C - 1, C):
        ix.setW(w)
        got = _coarse_cells(mi, ix, Q)
        exp = np.stack([ref.nearest_coarse(q, w) for q in Q])
        assert np.array_equal(got, exp), w
    ix.close()


@pytest.mark.parametrize("S", [2, 3])
def test_virtual_shards_on_one_device(mi, oracle, S):
    """Multi-GPU path without a cluster: S shards (whole lists, cell mod S) on one device, the
    same partial-search and merge kernels the RCCL path uses (kernel K5)."""
    import torch

    sh = importlib.import_module("multimedia-indexing_amd.sharded")
    D, C, m, ks, n, w = 32, 24, 8, 256, 9000, 6
    p = synth.make_ivfpq_problem(n=n, D=D, C=C, m=m, ks=ks, nq=40, seed=5 + S)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    ref.add_vectors(p["base"])
    full = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    full.loadCoarseQuantizer(p["coarse"])
    full.loadProductQuantizer(p["pq"])
    full.setW(w)
    cells, codes = full.encode(p["base"])
    shards = []
    for r in range(S):
        ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
        ix.loadCoarseQuantizer(p["coarse"])
        ix.loadProductQuantizer(p["pq"])
        ix.setW(w)
        own = np.nonzero(sh.owner_of_cell(cells, S) == r)[0]
        ix.loadIndex(own.astype(np.int32), cells[own], codes[own])
        shards.append(ix)
    Q = torch.tensor(p["queries"], dtype=torch.float64, device="cuda")
    engines = [sh.HipShardEngine(ix._h, D, w, 0) for ix in shards]
    probe, pdist = engines[0].coarse(Q)
    for k in (1, 10, 100):
        # two-phase form with the threshold exchange the RCCL path does (MIN over shards)
        Ts = [e.pass_a(k, Q, probe) for e in engines]
        Tmin = torch.stack(Ts).min(0).values.contiguous()
        parts2 = [e.pass_b(k, Q, probe, pdist if k != 10 else None, Tmin) for e in engines]  # (k = 10: bound from the centroids)
        i2, d2, c2, _ = engines[0].merge(k, torch.stack([x[0] for x in parts2]), torch.stack([x[1] for x in parts2]),
                                      torch.stack([x[2] for x in parts2]))
        torch.cuda.synchronize()
        assert_same((i2.cpu().numpy(), d2.cpu().numpy(), c2.cpu().numpy()), ref.search_batch(p["queries"], k))
        parts = [e.search_partial(k, Q, probe) for e in engines]
        iid, dd, cnt, _ = engines[0].merge(k, torch.stack([x[0] for x in parts]), torch.stack([x[1] for x in parts]),
                                           torch.stack([x[2] for x in parts]))
        torch.cuda.synchronize()
        # ragged form (what the variable-size all-to-all delivers): compact every shard's lists, concatenate
        pcs = torch.stack([x[2] for x in parts])
        comp = [engines[0].compact(k, x[0], x[1], x[2], int(x[2].sum())) for x in parts]
        flat = pcs.reshape(-1).to(torch.int64)
        poff = (torch.cumsum(flat, 0) - flat).reshape(pcs.shape).contiguous()
        ri, rd_, rcnt, _ = engines[0].merge(k, torch.cat([c[0] for c in comp]), torch.cat([c[1] for c in comp]), pcs.contiguous(), poff)
        torch.cuda.synchronize()
        assert torch.equal(ri, iid) and torch.equal(rd_, dd) and torch.equal(rcnt, cnt)
        # same merge on the host mirror
        hi, hd, hc, _ = sh.merge_partials_host(k, torch.stack([x[0] for x in parts]).cpu().numpy(),
                                            torch.stack([x[1] for x in parts]).cpu().numpy(),
                                            torch.stack([x[2] for x in parts]).cpu().numpy())
        assert np.array_equal(iid.cpu().numpy(), hi) and np.array_equal(dd.cpu().numpy(), hd)
        assert_same((iid.cpu().numpy(), dd.cpu().numpy(), cnt.cpu().numpy()), ref.search_batch(p["queries"], k))
    for ix in shards + [full]:
        ix.close()


@pytest.mark.parametrize("S,shape", [(2, "bytes"), (3, "bytes"), (2, "big_table")])
def test_virtual_shards_straddling_ties(mi, oracle, S, shape):
    """FLAGGED tie fixture for the multi-GPU path: every vector indexed three times, so exact distance ties straddle k; the
    cross-shard replay (mmidx_merge_partials_device flags + the three mmidx_shard_tie_phase_device passes, reductions done
    here over S virtual shards on one device) must return the single queue's answer (IVFPQ.java:445)."""
    import torch

    sh = importlib.import_module("multimedia-indexing_amd.sharded")
    D, C, m, ks, w = 32, 12, 8, 256, 5
    if shape == "big_table":  # short codes, 32 x 1024 doubles = 256 KiB: the table in global scratch (k_shard_tie with GLUT)
        D, m, ks = 64, 32, 1024
    p = synth.make_ivfpq_problem(n=1200, D=D, C=C, m=m, ks=ks, nq=48, seed=60 + S)
    base = np.concatenate([p["base"]] * 3)
    base = base[np.random.default_rng(2).permutation(len(base))]
    n = len(base)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    ref.add_vectors(base)
    full = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    full.loadCoarseQuantizer(p["coarse"])
    full.loadProductQuantizer(p["pq"])
    full.setW(w)
    cells, codes = full.encode(base)
    shards = []
    for r in range(S):
        ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
        ix.loadCoarseQuantizer(p["coarse"])
        ix.loadProductQuantizer(p["pq"])
        ix.setW(w)
        own = np.nonzero(sh.owner_of_cell(cells, S) == r)[0]
        ix.loadIndex(own.astype(np.int32), cells[own], codes[own])
        shards.append(ix)
    Q = torch.tensor(p["queries"], dtype=torch.float64, device="cuda")
    engines = [sh.HipShardEngine(ix._h, D, w, 0) for ix in shards]
    probe, pdist = engines[0].coarse(Q)
    tied_total = 0
    for k in (1, 5, 30):
        Ts = [e.pass_a(k, Q, probe) for e in engines]
        Tmin = torch.stack(Ts).min(0).values.contiguous()
        parts = [e.pass_b(k, Q, probe, pdist, Tmin) for e in engines]
        iid, dd, cnt, flag = engines[0].merge(k, torch.stack([x[0] for x in parts]), torch.stack([x[1] for x in parts]),
                                              torch.stack([x[2] for x in parts]))
        rid, rd, rc = ref.search_batch(p["queries"], k)
        _, rd1, rc1 = ref.search_batch(p["queries"], k + 1)
        want_flag = np.array([int(rc1[q] > k and rd1[q, k - 1] == rd1[q, k]) for q in range(len(rc))])
        assert np.array_equal(flag.cpu().numpy(), want_flag)
        tied_total += int(want_flag.sum())
        fq = torch.nonzero(flag, as_tuple=False).reshape(-1).to(torch.int32)
        fq = torch.cat([fq, torch.full((3,), -1, dtype=torch.int32, device="cuda")])  # (unused slots are skipped)
        F = fq.shape[0]
        safe = fq.clamp(min=0).long()
        tau = dd[safe, k - 1].contiguous()
        cnts = [torch.zeros((F, w, 2), dtype=torch.int32, device="cuda") for _ in engines]
        for e, c in zip(engines, cnts):
            e.tie_phase(0, k, Q, probe, fq, tau, c, torch.zeros(F, dtype=torch.int32, device="cuda"), torch.zeros((F, k), dtype=torch.int32, device="cuda"))
        counts = torch.stack(cnts).sum(0).to(torch.int32).contiguous()
        pbs = [torch.zeros(F, dtype=torch.int32, device="cuda") for _ in engines]
        for e, pb in zip(engines, pbs):
            e.tie_phase(1, k, Q, probe, fq, tau, counts, pb, torch.zeros((F, k), dtype=torch.int32, device="cuda"))
        pB = torch.stack(pbs).sum(0).to(torch.int32).contiguous()
        tis = [torch.full((F, k), -1, dtype=torch.int32, device="cuda") for _ in engines]
        for e, ti in zip(engines, tis):
            e.tie_phase(2, k, Q, probe, fq, tau, counts, pB, ti)
        ties = torch.stack(tis).max(0).values
        torch.cuda.synchronize()
        out = iid.clone()
        for f in range(F - 3):
            row = int(fq[f])
            out[row] = torch.where(ties[f] >= 0, ties[f], out[row])
        assert_same((out.cpu().numpy(), dd.cpu().numpy(), cnt.cpu().numpy()), (rid, rd, rc))
    assert tied_total >= 20
    for ix in shards + [full]:
        ix.close()


@pytest.mark.parametrize("margin", [0, 1024])
def test_shard_pass_a_item_compaction(mi, oracle, margin):
    """A shard's pass A runs over the queries whose nearest list it holds (k_passa_items), in a grid sized for the expected
    share; with margin 0 and queries that crowd onto one shard most of them overflow that grid and must come back through
    the hand-back list (the K3 launch behind K3h).  Three shards, long lists (K3h forced), results against the oracle."""
    import torch

    sh = importlib.import_module("multimedia-indexing_amd.sharded")
    S, D, C, m, ks, n, w, k = 3, 32, 24, 16, 256, 30000, 5, 20
    p = synth.make_ivfpq_problem(n=6000, D=D, C=C, m=m, ks=ks, nq=8, seed=77)
    base, _ = synth.mixture(n, D, C, sigma=0.3, seed=3)
    rng = np.random.default_rng(8)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    ref.add_vectors(base)
    full = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    full.loadCoarseQuantizer(p["coarse"])
    full.loadProductQuantizer(p["pq"])
    full.setW(w)
    cells, codes = full.encode(base)
    # queries: 260 near vectors of shard 0's cells, 40 anywhere
    own0 = np.nonzero(sh.owner_of_cell(cells, S) == 0)[0]
    qsrc = np.concatenate([rng.choice(own0, 260), rng.choice(n, 40)])
    Qn = base[qsrc] + 0.01 * rng.standard_normal((len(qsrc), D))
    shards = []
    for r in range(S):
        ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
        ix.loadCoarseQuantizer(p["coarse"])
        ix.loadProductQuantizer(p["pq"])
        ix.setW(w)
        own = np.nonzero(sh.owner_of_cell(cells, S) == r)[0]
        ix.loadIndex(own.astype(np.int32), cells[own], codes[own])
        ix.set_option("passa_hist", 1)
        ix.set_option("passa_item_min", 1)
        ix.set_option("passa_item_margin", margin)
        shards.append(ix)
    # the fixture must be tie-free (the codebook was learned on other data: equal codes -> equal distances): keep the
    # queries whose k + 1 best distances are distinct
    _, ed, _ = ref.search_batch(Qn, k + 1)
    Qn = Qn[(np.diff(ed, axis=1) > 0).all(1)]
    assert len(Qn) > 200
    Q = torch.tensor(Qn, dtype=torch.float64, device="cuda")
    engines = [sh.HipShardEngine(ix._h, D, w, 0) for ix in shards]
    probe, pdist = engines[0].coarse(Q)
    Ts = [e.pass_a(k, Q, probe) for e in engines]
    Tmin = torch.stack(Ts).min(0).values.contiguous()
    parts = [e.pass_b(k, Q, probe, pdist, Tmin) for e in engines]
    iid, dd, cnt, _ = engines[0].merge(k, torch.stack([x[0] for x in parts]), torch.stack([x[1] for x in parts]), torch.stack([x[2] for x in parts]))
    torch.cuda.synchronize()
    assert_same((iid.cpu().numpy(), dd.cpu().numpy(), cnt.cpu().numpy()), ref.search_batch(Qn, k))
    # the overflow really happened with margin 0: more owned queries on shard 0 than blocks launched
    own_q = sh.owner_of_cell(probe.cpu().numpy()[:, 0], S)
    if margin == 0:
        assert (own_q == 0).sum() > 1.15 * len(Qn) / 3 + 20
    for ix in shards + [full]:
        ix.close()


def test_coarse_topw_massive_ties(mi, oracle):
    """FLAGGED tie fixture: almost all coarse centroids identical -> > 1024 exactly equal distances;
    exercises the overflow path of the fast top-w kernel and the queue's tie rule."""
    D, m, ks, C = 8, 2, 16, 1300
    rng = np.random.default_rng(3)
    coarse = np.tile(rng.standard_normal((1, D)), (C, 1))
    special = rng.choice(C, size=9, replace=False)
    coarse[special] += 0.05 * rng.standard_normal((9, D))
    pq = rng.standard_normal((m, ks, D // m))
    ix = mi.IVFPQ(D, 10, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(coarse)
    ix.loadProductQuantizer(pq)
    ref = oracle.OracleIndex(oracle.KIND_IVFPQ, D, m, ks, C)
    ref.set_coarse(coarse)
    ref.set_pq(pq)
    Q = coarse[special[:4]] + 1e-4 * rng.standard_normal((4, D))
    Q = np.concatenate([Q, rng.standard_normal((4, D))])
    for w in (1, 5, 12, 200):
        ix.setW(w)
        got = _coarse_cells(mi, ix, Q)
        exp = np.stack([ref.nearest_coarse(q, w) for q in Q])
        assert np.array_equal(got, exp), w
    ix.close()


def test_yfcc_shape_d1024_m64(mi, oracle):
    """The reference's own headline shape (YFCC100MExample.java:85-90: d = 1024, m = 64 x 256),
    scaled down in n: exercises the largest LUT (128 KiB of LDS) and the exact coarse fallbacks."""
    D, C, m, ks, n, w, k = 1024, 32, 64, 256, 2500, 4, 10
    rng = np.random.default_rng(7)
    mu = rng.standard_normal((C, D))
    base = mu[rng.integers(0, C, n)] + 0.15 * rng.standard_normal((n, D))
    pq = 0.15 * rng.standard_normal((m, ks, D // m))
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(mu)
    ix.loadProductQuantizer(pq)
    ix.setW(w)
    ref = oracle_ivfpq(oracle, {"coarse": mu, "pq": pq}, D, m, ks, C, w)
    ix.indexVectors(list(range(n)), base)
    ref.add_vectors(base)
    assert np.array_equal(ix.listSizes(), ref.list_sizes())
    Q = base[:24] + 0.01 * rng.standard_normal((24, D))
    assert_same(ix.search_batch(k, Q), ref.search_batch(Q, k))
    ix.close()


@pytest.mark.parametrize("w,k,hist", [(2, 30, -1), (8, 30, -1), (8, 100, 0)])
def test_yfcc_shape_long_lists_fast_path(mi, oracle, w, k, hist):
    """YFCC100MExample.java:85-90, :155 -- d = 1024, m = 64 x 256, RandomPermutation, k = 30 -- with lists long enough
    (5,000 codes) for the fast kernels: pass A through K3h with one 1024-thread block per CU (128 KiB of table), pass B
    through K3g<64, 4, 16>.  The oracle is loaded with the device's own codes (encode parity is asserted on a sample)."""
    D, C, m, ks, n = 1024, 8, 64, 256, 40000
    rng = np.random.default_rng(17)
    mu = 0.3 * rng.standard_normal((C, D))  # (cells overlap: radius 16 against 14 between the means -- far probes hold neighbours)
    base = mu[rng.integers(0, C, n)] + 0.5 * rng.standard_normal((n, D))
    pq = 0.5 * rng.standard_normal((m, ks, D // m))
    ix = mi.IVFPQ(D, n, False, "", m, ks, 2, C, 512)  # RandomPermutation(1, D) derived natively
    ix.loadCoarseQuantizer(mu)
    ix.loadProductQuantizer(pq)
    ix.setW(w)
    ix.set_option("passa_hist", hist)
    ref = oracle_ivfpq(oracle, {"coarse": mu, "pq": pq}, D, m, ks, C, w, tr=2)
    cells, codes = ix.encode(base[:300])
    rcell, rcode = ref.encode_batch(base[:300])
    assert np.array_equal(cells, rcell) and np.array_equal(codes.astype(np.int32) + 128, rcode)
    ix.indexVectors(list(range(n)), base)
    off, iids, cds = ix.export()
    ref.load_lists(off, iids, cds)
    # perturbed base vectors, cell means, and points half way between two base vectors (their neighbours sit in several cells:
    # far probes pass the lower-bound filter and are verified exactly)
    mid = 0.5 * (base[100:124] + base[200:224])
    Q = np.concatenate([base[:40] + 0.01 * rng.standard_normal((40, D)), mu + 0.3 * rng.standard_normal((C, D)), mid])
    ix.set_profiling(True)
    got = ix.search_batch(k, Q)
    st = ix.get_stats()
    assert_same(got, ref.search_batch(Q, k))
    if w > 1:
        assert st["passb_items_last"] > 0 and st["verified_codes"] > 0  # (pass B ran, through K3g: K3f cannot hold this table)
    ix.close()


@pytest.mark.parametrize("D,m,C,n,w,k,dup", [
    (128, 16, 6, 30000, 6, 100, 1),   # <16, 8, 8>: the headline's instance
    (128, 8, 5, 24000, 5, 50, 3),     # <8, 8, 16>, every vector three times: ties everywhere, buffers overflow and are pruned
    (64, 16, 4, 20000, 4, 300, 1),    # dsub 4, k = 300: candidate buffers of 512 entries
    (128, 32, 4, 16000, 4, 20, 2),    # <32, 4, 4>: groups of four
    (96, 8, 4, 12000, 3, 10, 1),      # dsub 12: the run-time-dsub instance has no UNION twin (the plain one must serve it)
])
def test_union_instances_forced(mi, oracle, D, m, C, n, w, k, dup):
    """The K3g instances that rank the union of the verified candidates (`no_union` = -1 forces them; normally a device-reported
    hint picks them): overlapping cells and queries between cells, so that far probes feed the queue, survivors are verified a
    lane each, candidate buffers overflow and thresholds come down from the per-query histogram.  Same ids and distance bits as
    the oracle, and as the plain instances (`no_union` = 1)."""
    ks = 256
    rng = np.random.default_rng(D + m)
    mu = 0.5 * rng.standard_normal((C, D))
    base = mu[rng.integers(0, C, n // dup)] + rng.standard_normal((n // dup, D))
    base = np.concatenate([base] * dup)[rng.permutation((n // dup) * dup)]
    n = len(base)
    p = {"coarse": mu, "pq": np.stack([synth.kmeans((mu[rng.integers(0, C, 3000)] - base[:3000])[:, s * (D // m):(s + 1) * (D // m)], ks, iters=2, seed=s)
                                       for s in range(m)])}
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    ix.setW(w)
    ix.set_option("passa_hist", 1)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    Q = np.concatenate([0.5 * (base[:24] + base[100:124]), rng.standard_normal((8, D)), base[:8] + 0.01 * rng.standard_normal((8, D))])
    want = ref.search_batch(Q, k)
    got = {}
    for mode in (-1, 1, 0):
        ix.set_option("no_union", mode)
        ix.set_profiling(True)
        got[mode] = ix.search_batch(k, Q)
        st = ix.get_stats()
        assert_same(got[mode], want)
        if mode == -1 and D % 8 == 0 and (D // m) in (4, 8, 16):
            assert st["verified_codes"] > 0  # (the forced instance really had survivors to verify)
    ix.close()


@pytest.mark.parametrize("D,m,C,n,w,k,tr,dup", [
    (128, 16, 6, 30000, 6, 100, 0, 1),    # <NJ 4, dsub 8>: the headline's shape
    (128, 16, 5, 24000, 5, 50, 2, 3),     # RandomPermutation; every vector three times: ties everywhere, pools fill up
    (128, 8, 6, 24000, 6, 30, 0, 1),      # <4, 16>: cfg2's sub-quantizers in an IVF index
    (64, 8, 4, 20000, 4, 300, 2, 1),      # <2, 8>, k = 300 (beyond K3g's candidate buffers)
    (64, 4, 4, 16000, 3, 10, 0, 2),       # <2, 16>
    (32, 4, 7, 12000, 7, 1, 0, 1),        # <1, 8>, k = 1
    (32, 2, 3, 9000, 3, 700, 0, 1),       # <1, 16>, k = 700
    (128, 32, 5, 20000, 5, 100, 0, 1),    # <4, 4>: 4-dimensional sub-quantizers (32-byte codes): two 8-byte gathers per fragment, 32 lanes per verified survivor
    (64, 16, 4, 16000, 4, 30, 2, 2),      # <2, 4>, RandomPermutation, ties
    (32, 8, 3, 9000, 3, 10, 0, 1),        # <1, 4>
    (128, 16, 6, 24000, 6, 100, 1, 1),    # RandomRotation (an orthogonal matrix: the coarse bound applies with its measured margin)
    (64, 8, 5, 16000, 5, 20, 3, 2),       # a "rotation" matrix that is NOT orthogonal (columns stretched): the coarse bound stays off, K3m serves it
    (256, 16, 5, 12000, 5, 50, 0, 1),     # K3mk (mmidx_scan_mfma_kc.h): two 128-dimension chunks, 16-dimensional sub-quantizers
    (256, 32, 4, 10000, 4, 30, 2, 2),     # K3mk, 8-dimensional sub-quantizers, RandomPermutation, ties
    (384, 48, 4, 8000, 4, 20, 0, 1),      # K3mk, three chunks, 48-byte codes (three sub-quantizers per verifying lane)
    (1024, 64, 6, 8000, 6, 30, 2, 1),     # K3mk at YFCC100MExample.java:85-90's shape: eight chunks, 64 x 16 (8 code bytes per lane load)
    (512, 32, 5, 9000, 5, 40, 0, 1),      # K3mk, 32 x 16: 8 code bytes per lane load
    (1024, 128, 4, 6000, 4, 10, 0, 1),    # K3mk, 128 x 8: four 8-byte loads per lane and tile
    (256, 16, 3, 60000, 3, 50, 2, 1),     # K3mk, lists of ~20 k codes: items of eight 1024-code passes, several items per group, the chunk rotation across passes
])
def test_mfma_pass_b(mi, oracle, D, m, C, n, w, k, tr, dup):
    """K3m (`k_scan_mfma` + `k_mfma_verify` + `k_mfma_redo`, csrc/mmidx_scan_mfma.h): pass B as a certified lower bound on the matrix
    cores -- fp16 residuals x fp16 decoded codes, accumulators started at -||x||^2 / 2, one compare per (query, code) -- with exact
    fp64 distances for the survivors only (IVFPQ.java:429-446).  Overlapping cells and queries between cells, so that far probes
    feed the queue.  Default sizing, pieces of 64 codes (`mfma_sub`), a survivor list of 16 records (`mfma_qcap`: nearly every
    query is handed back to K3f through the redo path) and K3m off (`no_mfma`): ids and distance bits are the oracle's every time."""
    ks = 256
    rng = np.random.default_rng(D + m + k)
    mu = 0.5 * rng.standard_normal((C, D))
    base = mu[rng.integers(0, C, n // dup)] + rng.standard_normal((n // dup, D))
    base = np.concatenate([base] * dup)[rng.permutation((n // dup) * dup)]
    n = len(base)
    ds = D // m
    pq = np.stack([synth.kmeans((mu[rng.integers(0, C, 3000)] - base[:3000])[:, s * ds:(s + 1) * ds], ks, iters=2, seed=s) for s in range(m)])
    rot = None
    if tr in (1, 3):  # RandomRotation.java:44-49: the matrix is an input of the library
        rot = np.linalg.qr(rng.standard_normal((D, D)))[0]
        if tr == 3:
            rot = rot * np.linspace(1.0, 1.3, D)[None, :]
            tr = 1
    ix = mi.IVFPQ(D, n, False, "", m, ks, tr, C, 512, rot=rot)
    ix.loadCoarseQuantizer(mu)
    ix.loadProductQuantizer(pq)
    ix.setW(w)
    ref = oracle_ivfpq(oracle, {"coarse": mu, "pq": pq}, D, m, ks, C, w, tr=tr, perm=oracle.random_permutation(1, D) if tr == 2 else None, rot=rot)
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    Q = np.concatenate([0.5 * (base[:24] + base[100:124]), rng.standard_normal((8, D)), base[:40] + 0.01 * rng.standard_normal((40, D)), mu[:2]])
    want = ref.search_batch(Q, k)
    for sub, qcap, off in ((0, 0, 0), (64, 0, 0), (0, 16, 0), (4096, 0, 0), (0, 0, 1)):
        ix.set_option("mfma_sub", sub)
        ix.set_option("mfma_qcap", qcap)
        ix.set_option("no_mfma", off)
        ix.set_profiling(True)
        got = ix.search_batch(k, Q)
        st = ix.get_stats()
        assert_same(got, want)
        if off:
            assert st["mfma_survivors"] == 0
        elif w > 1:
            assert st["mfma_survivors"] > 0  # (K3m ran and its bound let codes through; those still at or below the final thresholds were verified exactly)
            if qcap:
                assert st["mfma_redo_queries"] > 0  # (the short survivor list really sent queries through the redo path)
    if D > 128:  # K3mk without LDS-DMA (the form D = 384 takes anyway), with 8 and 16 code tiles per wave
        ix.set_option("mfma_sub", 0)
        ix.set_option("mfma_qcap", 0)
        ix.set_option("no_mfma", 0)
        ix.set_option("mfma_kc_v1", 1)
        for tpw in (8, 16):
            ix.set_option("mfma_kc_tpw", tpw)
            ix.set_profiling(True)
            assert_same(ix.search_batch(k, Q), want)
            assert ix.get_stats()["mfma_survivors"] > 0
        ix.set_option("mfma_kc_tpw", 8)
        ix.set_option("mfma_kc_v1", 0)
    one = ix.search_batch(k, Q[:1])  # a one-query call: groups of one pair
    assert_same(one, tuple(a[:1] for a in want))
    ix.close()


@pytest.mark.parametrize("D,m,n,k,tr,chunk", [(128, 8, 70000, 100, 0, 8192), (128, 16, 40000, 10, 2, 4096), (64, 8, 30000, 200, 0, 4096), (32, 2, 50000, 5, 0, 16384),
                                                  (128, 32, 30000, 50, 0, 4096), (256, 16, 30000, 20, 0, 4096), (512, 64, 20000, 10, 2, 4096)])
def test_mfma_flat_pq(mi, oracle, D, m, n, k, tr, chunk):
    """Flat PQ (PQ.computeKnnADC, PQ.java:290-322) through K3m: the chunks 1 .. of the single list stand in for inverted lists, the
    residual is the query itself, survivors are verified from the queries' exact tables (k_flat_lut)."""
    ks = 256
    p = synth.make_pq_problem(n=8000, D=D, m=m, ks=ks, nq=12, seed=D + m, iters=2)
    rng = np.random.default_rng(n)
    base = rng.standard_normal((n, D))
    ix = mi.PQ(D, n, False, "", m, ks, tr, 512)
    ix.loadProductQuantizer(p["pq"])
    ix.set_option("flat_chunk", chunk)
    ref = oracle.OracleIndex(oracle.KIND_PQ, D, m, ks, transform=tr, perm=oracle.random_permutation(1, D) if tr == 2 else None)
    ref.set_pq(p["pq"])
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    Q = np.concatenate([base[:20] + 0.05 * rng.standard_normal((20, D)), rng.standard_normal((80, D))])
    want = ref.search_batch(Q, k)
    for sub, qcap, off in ((0, 0, 0), (256, 0, 0), (0, 64, 0), (0, 0, 1)):
        ix.set_option("mfma_sub", sub)
        ix.set_option("mfma_qcap", qcap)
        ix.set_option("no_mfma", off)
        ix.set_profiling(True)
        got = ix.search_batch(k, Q)
        st = ix.get_stats()
        assert_same(got, want)
        if not off:
            assert st["mfma_survivors"] > 0
    ix.close()


@pytest.mark.parametrize("D,m,C,n,w,k,tr,sep", [
    (128, 16, 24, 30000, 24, 100, 0, 0.8),   # <8, 16 waves>: the headline's shape; cells far enough apart that far pairs end at Smin >= T
    (128, 8, 12, 20000, 12, 20, 2, 0.8),     # <16, 8 waves>, RandomPermutation
    (64, 16, 10, 20000, 10, 50, 0, 0.3),     # dsub 4, overlapping cells: most pairs stay alive
    (128, 32, 12, 16000, 12, 20, 0, 0.8),    # m = 32: two slice groups, the parts of Smin meet by atomicAdd
    (1024, 64, 12, 24000, 12, 30, 2, 0.25),  # YFCC100MExample.java:85-90: 64 x 16, four slice groups
    (96, 8, 6, 12000, 6, 10, 0, 0.8),        # dsub 12: K3s does not apply, the option must be harmless
])
def test_smin_prefilter_forced(mi, oracle, D, m, C, n, w, k, tr, sep):
    """K3s (`k_pair_smin` + `k_pair_recount`, option `smin_pre`): the certified lower bound of every far pair's Smin, computed in
    front of pass B's counting sort with the codebook in registers; pairs with Smin >= T leave before K3g builds a table for them.
    Forced on (1; matrix-core and packed-FMA form), off (0) and hint-driven (-1, three calls so that the device's figures of one
    call steer the next): ids and distance bits are the oracle's every time, and the forced filter never leaves more pairs than the
    plain path."""
    ks = 256
    rng = np.random.default_rng(D + m + C)
    mu = sep * rng.standard_normal((C, D))
    base = mu[rng.integers(0, C, n)] + 0.5 * rng.standard_normal((n, D))
    pq = np.stack([synth.kmeans((mu[rng.integers(0, C, 2000)] - base[:2000])[:, s * (D // m):(s + 1) * (D // m)], ks, iters=1, seed=s)
                   for s in range(m)])
    ix = mi.IVFPQ(D, n, False, "", m, ks, tr, C, 512)
    ix.loadCoarseQuantizer(mu)
    ix.loadProductQuantizer(pq)
    ix.setW(w)
    ref = oracle_ivfpq(oracle, {"coarse": mu, "pq": pq}, D, m, ks, C, w, tr=tr)
    ix.indexVectors(list(range(n)), base)
    off, iids, cds = ix.export()
    ref.load_lists(off, iids, cds)
    Q = np.concatenate([base[:48] + 0.01 * rng.standard_normal((48, D)), 0.5 * (base[100:116] + base[200:216]), mu[:4]])
    want = ref.search_batch(Q, k)
    items = {}
    for mode, valu, b16 in ((1, 0, 1), (1, 1, 1), (1, 0, 0), (0, 0, 1), (-1, 0, 1), (-1, 0, 1), (-1, 0, 1)):
        ix.set_option("smin_pre", mode)
        ix.set_option("smin_valu", valu)  # (1: the packed-FMA form of the kernel instead of the matrix-core one)
        ix.set_option("smin_bf16", b16)   # (0: without the bf16 first stage that 16-dimensional sub-quantizers get)
        ix.set_profiling(True)
        got = ix.search_batch(k, Q)
        st = ix.get_stats()
        assert_same(got, want)
        items.setdefault((mode, valu) if b16 else (mode, 2), st["passb_items_last"])
    assert items[(1, 2)] == items[(1, 0)]  # (what the bf16 stage cannot drop is decided by the fp32 stage: the same pairs are left)
    # the two forms bound the same minima (their entries differ in the last bits): a pair can change sides only at Smin ~ T
    assert abs(items[(1, 0)] - items[(1, 1)]) <= 2 + items[(0, 0)] // 200
    items = {1: items[(1, 0)], 0: items[(0, 0)]}
    assert 0 <= items[1] <= items[0]
    if sep >= 0.8 and m <= 16 and (D // m) in (4, 8, 16):
        assert items[1] < items[0]  # (the filter really removed pairs the coarse bound had left)
    ix.close()


@pytest.mark.parametrize("D,C,m,w,k,tr,n", [
    (1024, 300, 64, 8, 30, 2, 6000),     # YFCC100MExample.java:85-90: eight k chunks, RandomPermutation
    (256, 260, 32, 5, 10, 0, 5000),      # two chunks
    (384, 520, 8, 3, 20, 0, 4000),       # three chunks, dsub 48
    (1024, 1100, 64, 64, 30, 2, 5000),   # w = 64 (Example.java:96-97): the lane-per-candidate exact stage behind it
])
def test_coarse_stage_long_vectors_dma(mi, oracle, D, C, m, w, k, tr, n):
    """K1e' for vectors of several 128-dimension chunks (`k_coarse_gmin16_dma_kc`, option `coarse_dma_kc`): the centroid tiles of
    every chunk go through the two LDS-DMA half-tile buffers, the accumulators stay across the chunks.  With and without it the
    answers are the oracle's, ids and distance bits (the certified selection behind it does not depend on which kernel produced
    the group minima, as long as they are within the error bound)."""
    rng = np.random.default_rng(D + C)
    mu = rng.standard_normal((C, D))
    base = mu[rng.integers(0, C, n)] + 0.3 * rng.standard_normal((n, D))
    pq = 0.3 * rng.standard_normal((m, 256, D // m))
    ref = oracle_ivfpq(oracle, {"coarse": mu, "pq": pq}, D, m, 256, C, w, tr=tr)
    Q = np.concatenate([base[:40] + 0.01 * rng.standard_normal((40, D)), mu[:8] + 0.2 * rng.standard_normal((8, D))])
    ix = mi.IVFPQ(D, n, False, "", m, 256, tr, C, 512)
    ix.loadCoarseQuantizer(mu)
    ix.loadProductQuantizer(pq)
    ix.setW(w)
    ix.indexVectors(list(range(n)), base)
    off, iids, cds = ix.export()
    ref.load_lists(off, iids, cds)
    want = ref.search_batch(Q, k)
    for opt in (1, 0):
        ix.set_option("coarse_dma_kc", opt)
        assert_same(ix.search_batch(k, Q), want)
    ix.close()


def test_snapshot_roundtrip(mi, oracle, tmp_path):
    """saveSnapshot / loadSnapshot (flat restart path): identical answers, ids preserved."""
    D, C, m, ks, n, w, k = 32, 16, 8, 256, 3000, 4, 10
    p = synth.make_ivfpq_problem(n=n, D=D, C=C, m=m, ks=ks, nq=16, seed=41)
    a = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    b = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    for ix in (a, b):
        ix.loadCoarseQuantizer(p["coarse"])
        ix.loadProductQuantizer(p["pq"])
        ix.setW(w)
    a.indexVectors([f"im{i}" for i in range(n)], p["base"])
    f = str(tmp_path / "snap.mmidx")
    a.saveSnapshot(f)
    b.loadSnapshot(f)
    assert b.size() == n and b.getLoadCounter() == n and np.array_equal(a.listSizes(), b.listSizes())
    ra, rb = a.search_batch(k, p["queries"]), b.search_batch(k, p["queries"])
    assert_same(rb, ra)
    assert b.computeNearestNeighbors(k, p["queries"][0]).getIds() == a.computeNearestNeighbors(k, p["queries"][0]).getIds()
    assert b.indexVector("im5", p["base"][5]) is False  # duplicate id known after the reload
    a.close()
    b.close()


def test_native_snapshot_reload_1m_against_the_oracle(mi, oracle, tmp_path):
    """mmidx_save / mmidx_load (ABI 8, the fast-restart path next to loadIndexInMemory, IVFPQ.java:680-728): one million codes are
    written by one handle and read by a FRESH one; the reloaded index is compared with the ORACLE holding the same records (not
    with the first handle): list sizes, exported lists (order inside a list = arrival order), ids and distance bits of a search.
    Also: a shape mismatch, a non-empty target and a truncated file are refused."""
    D, C, m, ks, n, w, k, nq = 64, 128, 16, 256, 1_000_000, 8, 20, 64
    rng = np.random.default_rng(5)
    coarse = rng.standard_normal((C, D))
    pq = 0.3 * rng.standard_normal((m, ks, D // m))
    # records straight as codes (indexPQCode, IVFPQ.java:357-386): a million encodes on the CPU oracle would take minutes
    cells = rng.integers(0, C, n).astype(np.int32)
    codes = rng.integers(0, ks, (n, m)).astype(np.int32)
    stored = (codes - 128).astype(np.int8)
    iids = np.arange(n, dtype=np.int32)
    a = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    a.loadCoarseQuantizer(coarse)
    a.loadProductQuantizer(pq)
    a.setW(w)
    N = mi._native
    N.check(N.lib().mmidx_add_codes(a._h, n, iids.ctypes.data, cells.ctypes.data, stored.ctypes.data))
    f = str(tmp_path / "big.mmidx")
    N.check(N.lib().mmidx_save(a._h, f.encode()))
    a.close()
    assert os.path.getsize(f) == 56 + 8 * (C + 1) + 4 * n + m * n
    b = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    b.loadCoarseQuantizer(coarse)
    b.loadProductQuantizer(pq)
    b.setW(w)
    N.check(N.lib().mmidx_load(b._h, f.encode()))
    ref = oracle.OracleIndex(oracle.KIND_IVFPQ, D, m, ks, C)
    ref.set_coarse(coarse)
    ref.set_pq(pq)
    ref.set_w(w)
    order = np.argsort(cells, kind="stable")  # list-major, arrival order inside a list
    ref.load_lists(np.concatenate([[0], np.cumsum(np.bincount(cells, minlength=C))]).astype(np.int64), iids[order], stored[order])
    assert b.size() == n and np.array_equal(b.listSizes(), ref.list_sizes())
    off, eid, ecodes = b.export()
    assert np.array_equal(eid, iids[order]) and np.array_equal(ecodes, stored[order])
    Q = coarse[cells[:nq]] + 0.3 * rng.standard_normal((nq, D))
    got = b.search_batch(k, Q)
    exp = ref.search_batch(Q, k, nthreads=4)
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]) and np.array_equal(got[2], exp[2])
    # refused: a non-empty index, another shape, a damaged file
    assert N.lib().mmidx_load(b._h, f.encode()) == N.ERR_INVALID_ARG
    c = mi.IVFPQ(D, n, False, "", m, ks, 0, C // 2, 512)
    assert N.lib().mmidx_load(c._h, f.encode()) == N.ERR_INVALID_ARG
    c.close()
    g = str(tmp_path / "cut.mmidx")
    open(g, "wb").write(open(f, "rb").read()[:1 << 20])
    d = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    assert N.lib().mmidx_load(d._h, g.encode()) == N.ERR_INVALID_ARG
    d.close()
    b.close()


@pytest.mark.parametrize("D,m,ks,n,k", [(32, 8, 256, 9000, 10), (128, 16, 256, 3000, 100), (24, 6, 64, 2000, 5), (16, 4, 16, 3000, 7)])
def test_pq_sdc_by_internal_id(mi, oracle, D, m, ks, n, k):
    """computeNearestNeighbors(k, internalId) on PQ = computeKnnSDC (PQ.java:334-374): code-to-code
    distances as ONE sequential fp64 chain over all D dimensions.  The last case has few distinct
    codes -> heavy exact ties (the query's own duplicates at distance 0 included)."""
    p = synth.make_pq_problem(n=2000, D=D, m=m, ks=ks, nq=4, seed=n)
    if ks == 16:
        base, _ = synth.mixture(n, D, 6, sigma=0.01, seed=3)
    else:
        base = np.random.default_rng(5).standard_normal((n, D))
    ix = mi.PQ(D, n, False, "", m, ks, 0, 512)
    ix.loadProductQuantizer(p["pq"])
    ref = oracle.OracleIndex(oracle.KIND_PQ, D, m, ks)
    ref.set_pq(p["pq"])
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    qi = np.array([0, 1, n // 2, n - 1, 17, 17, 3], np.int32)
    iids, dists, cnt = ix.search_sdc_batch(k, qi)
    for j, iid in enumerate(qi):
        rid, rd = ref.search_sdc(int(iid), k)
        assert cnt[j] == len(rid)
        assert np.array_equal(iids[j, :cnt[j]], rid)
        assert np.array_equal(dists[j, :cnt[j]], rd)
    a_i, a_d = ix.computeNearestNeighborsInternalById(k, 17)
    assert np.array_equal(a_i, iids[4, :cnt[4]]) and np.array_equal(a_d, dists[4, :cnt[4]])
    ans = ix.computeNearestNeighbors(k, "17")  # ASS:320-333: id -> iid -> SDC
    assert len(ans.getIds()) == cnt[4] and ans.getIds()[0] in [str(i) for i in iids[4, :cnt[4]]]
    with pytest.raises(mi.MmidxError):
        ix.search_sdc_batch(k, np.array([n], np.int32))
    ix.close()
    # IVFPQ: computeKnnIVFSDC is a stub returning null in the reference (IVFPQ.java:509-511)
    pi = synth.make_ivfpq_problem(n=500, D=16, C=4, m=4, ks=16, nq=2, seed=1)
    iv = mi.IVFPQ(16, 500, False, "", 4, 16, 0, 4, 512)
    iv.loadCoarseQuantizer(pi["coarse"])
    iv.loadProductQuantizer(pi["pq"])
    iv.indexVectors([str(i) for i in range(500)], pi["base"])
    with pytest.raises(mi.MmidxError) as ei:
        iv.search_sdc_batch(3, np.array([0], np.int32))
    assert ei.value.status == 10
    iv.close()


@pytest.mark.parametrize("case", ["long_lists", "ties", "short_lists", "flat_pq", "k1023_falls_back", "two_chunks", "two_chunks_ties"])
def test_pass_a_histogram_kernel(mi, oracle, case):
    """K3h (histogram-thresholded pass A) forced on: same answers as the oracle on long lists, on
    tie-heavy lists (handed back to K3 through the fallback list), on lists shorter than a segment,
    and on the flat PQ chunk-as-probe path."""
    rng = np.random.default_rng(9)
    if case == "flat_pq":
        D, m, ks, n, k = 32, 8, 256, 150000, 20
        p = synth.make_pq_problem(n=4000, D=D, m=m, ks=ks, nq=12, seed=21)
        base = rng.standard_normal((n, D))
        ix = mi.PQ(D, n, False, "", m, ks, 0, 512)
        ix.loadProductQuantizer(p["pq"])
        ref = oracle.OracleIndex(oracle.KIND_PQ, D, m, ks)
        ref.set_pq(p["pq"])
        ix.set_option("passa_hist", 1)
        ix.indexVectors([str(i) for i in range(n)], base)
        ref.add_vectors(base)
        assert_same(ix.search_batch(k, p["queries"]), ref.search_batch(p["queries"], k))
        ix.close()
        return
    D, m, ks = 32, 8, 256
    if case == "long_lists":
        C, w, n, ks_ = 4, 3, 60000, 256
    elif case == "ties":
        C, w, n, ks_ = 4, 4, 30000, 4  # 4 centroids per sub-quantizer: few distinct codes, thousands of exact ties
    elif case == "two_chunks":  # lists longer than one 16384-code chunk: two K3h blocks per (query, list)
        C, w, n, ks_ = 3, 3, 75000, 256
    elif case == "two_chunks_ties":  # ... and handed back to K3 per (item, chunk) through the fallback list
        C, w, n, ks_ = 3, 2, 70000, 4
    elif case == "short_lists":
        C, w, n, ks_ = 64, 8, 6000, 256
    else:
        C, w, n, ks_ = 4, 2, 20000, 256
    p = synth.make_ivfpq_problem(n=min(n, 8000), D=D, C=C, m=m, ks=ks_, nq=10, seed=17)
    base, _ = synth.mixture(n, D, C, sigma=0.3, seed=23)
    ix = mi.IVFPQ(D, n, False, "", m, ks_, 0, C, 512)
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    ix.setW(w)
    ix.set_option("passa_hist", 1)
    ref = oracle_ivfpq(oracle, p, D, m, ks_, C, w)
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    q = base[rng.choice(n, 12, replace=False)] + 0.01 * rng.standard_normal((12, D))
    ks_list = (1023,) if case == "k1023_falls_back" else (1, 10, 100, 255)
    for k in ks_list:
        assert_same(ix.search_batch(k, q), ref.search_batch(q, k))
    # the 512-thread form of K3h (eight waves over one table; measured slower on cfg4, kept as a switch)
    ix.set_option("passa_wide", 1)
    for k in ks_list[-2:]:
        assert_same(ix.search_batch(k, q), ref.search_batch(q, k))
    ix.close()


@pytest.mark.parametrize("D,C,w", [(24, 1000, 7), (130, 700, 12), (128, 2049, 31), (260, 1536, 5), (64, 4100, 40), (16, 16390, 9), (100, 1000, 9), (128, 8192, 32), (128, 128, 3),
                                   (256, 1536, 6), (128, 4096, 8), (64, 900, 63)])
def test_coarse_stage_group_minima_shapes(mi, oracle, D, C, w):
    """K1e/K1f (bf16-split MFMA dot products + group minima) on shapes that exercise its padding: D not a multiple
    of 32, more than one 128-wide k chunk (D > 128), C not a multiple of 128, C / 8 groups per thread > 1; clustered
    centroids (realistic) plus exact duplicates (ties).  Selected cells AND the exact distances that travel to the
    other ranks must equal the oracle's / the exact fp64 path's."""
    import torch

    nat = importlib.import_module("multimedia-indexing_amd._native")
    rng = np.random.default_rng(D * 7 + C)
    centers = 4.0 * rng.standard_normal((C // 40 + 2, D))
    coarse = centers[rng.integers(0, len(centers), C)] + 0.7 * rng.standard_normal((C, D))
    coarse[C // 2:C // 2 + 20] = coarse[:20]  # duplicates -> equal distances
    m = 2 if D % 2 == 0 else 1
    pq = rng.standard_normal((m, 16, D // m))
    ix = mi.IVFPQ(D, 10, False, "", m, 16, 0, C, 512)
    ix.loadCoarseQuantizer(coarse)
    ix.loadProductQuantizer(pq)
    ix.setW(w)
    ref = oracle.OracleIndex(oracle.KIND_IVFPQ, D, m, 16, C)
    ref.set_coarse(coarse)
    ref.set_pq(pq)
    Q = np.concatenate([coarse[rng.integers(0, C, 40)] + 0.05 * rng.standard_normal((40, D)), coarse[:6] + 1e-9,
                        5.0 * rng.standard_normal((10, D))])
    dQ = torch.tensor(Q, dtype=torch.float64, device="cuda")
    res = {}
    # (2: the exact stage by a block per query (k_coarse_select_list) also where the wave-per-query form (k_coarse_front_sel: D = 64,
    #  128, 256, w < 64, at most 64 candidates) would have answered)
    for v1 in (0, 1, 2):
        ix.set_option("coarse_v1", v1 & 1)
        ix.set_option("coarse_wave_sel", 0 if v1 == 2 else 1)
        cells = torch.empty(len(Q), w, dtype=torch.int32, device="cuda")
        cd = torch.empty(len(Q), w, dtype=torch.float64, device="cuda")
        nat.check(mi.lib().mmidx_coarse_device(ix._h, len(Q), dQ.data_ptr(), cells.data_ptr(), cd.data_ptr(), None))
        torch.cuda.synchronize()
        res[v1] = (cells.cpu().numpy(), cd.cpu().numpy())
    exp = np.stack([ref.nearest_coarse(q, w) for q in Q])
    assert np.array_equal(res[0][0], exp) and np.array_equal(res[1][0], exp) and np.array_equal(res[2][0], exp)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][1], res[2][1])
    # exact distances: sequential fp64 sums of the selected cells
    for qi in (0, 17, len(Q) - 1):
        for r in (0, w - 1):
            acc = 0.0
            for a, b in zip(coarse[exp[qi, r]], Q[qi]):
                df = a - b
                acc += df * df
            assert res[0][1][qi, r] == acc
    ix.close()


@pytest.mark.parametrize("m,D,k", [(16, 64, 50), (32, 64, 100), (8, 128, 200)])
def test_pass_a_histogram_kernel_other_widths(mi, oracle, m, D, k):
    """K3h template instances m = 16 / 32 and a larger k, long lists (forced on)."""
    ks, C, w, n = 256, 3, 3, 36000
    p = synth.make_ivfpq_problem(n=6000, D=D, C=C, m=m, ks=ks, nq=8, seed=m + k)
    base, _ = synth.mixture(n, D, C, sigma=0.3, seed=m)
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    ix.setW(w)
    ix.set_option("passa_hist", 1)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    rng = np.random.default_rng(1)
    q = base[rng.choice(n, 10, replace=False)] + 0.01 * rng.standard_normal((10, D))
    assert_same(ix.search_batch(k, q), ref.search_batch(q, k))
    ix.close()


@pytest.mark.parametrize("offset", [0.0, 100.0, 1e5])
def test_pass_a_integer_kernel_fp32_table_guard(mi, oracle, offset):
    """K3q builds its integer table in fp32 from fp32 copies of residual and codebook; a per-query bound on what that can cost
    (2 u sqrt(4128 scale) (||r|| + max ||x||) <= 0.25 table units) decides whether the query may use it.  A codebook -- and with it
    the residuals -- far from the origin with a small spread is what the bound is for: offset 100 passes it with a visible error
    (0.07 units), offset 1e5 fails it and every query goes to the exact kernel; the answers are the oracle's either way."""
    D, C, m, ks, w, k = 128, 2, 16, 256, 2, 30
    rng = np.random.default_rng(int(offset) + 11)
    coarse = rng.standard_normal((C, D))
    pq = offset + rng.standard_normal((m, ks, D // m))
    n = 9000
    base = -offset + 1.5 * rng.standard_normal((n, D))  # residuals c - v = offset + noise: the codebook's neighbourhood
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(coarse)
    ix.loadProductQuantizer(pq)
    ix.setW(w)
    ix.set_option("passa_q", 1)
    ref = oracle.OracleIndex(oracle.KIND_IVFPQ, D, m, ks, C)
    ref.set_coarse(coarse)
    ref.set_pq(pq)
    ref.set_w(w)
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    Q = base[rng.choice(n, 24, replace=False)] + 0.05 * rng.standard_normal((24, D))
    got = ix.search_batch(k, Q)
    assert ix.get_dispatch()["pass_a"].startswith("K3q")
    assert_same(got, ref.search_batch(Q, k))
    ix.close()


@pytest.mark.parametrize("nq", [1, 2, 3, 4, 7])
def test_pass_a_integer_kernel_lane_with_more_candidates_than_slots(mi, oracle, nq):
    """K3q keeps six keys per lane and query; a lane sees the list positions p = lane (mod 256).  Here the 40 codes nearest to the
    queries are INSERTED at positions 7, 263, 519, ... of the queries' list, so lane 7 holds far more than six of every query's
    K1 = 21 best: its sixth key lies within the cut, the block builds its table again and the lane walks its codes once more
    (mmidx_scan_q.h: the rescue).  One and two queries take the two-query body, three and four the four-query body, seven two groups."""
    D, C, m, ks, w, k = 128, 2, 16, 256, 2, 20  # (w = 2: the two-pass search; one pass would be the exact kernel K3)
    n_list = 10400
    p = synth.make_ivfpq_problem(n=6000, D=D, C=C, m=m, ks=ks, nq=8, seed=606)
    rng = np.random.default_rng(607)
    c0 = p["coarse"][0]
    V = c0 + 0.25 * rng.standard_normal((n_list, D))
    q0 = c0 + 0.25 * rng.standard_normal(D)
    Q = q0 + 0.002 * rng.standard_normal((nq, D))
    # the ADC order of the list for q0, from a scratch oracle index in arbitrary order
    tmp = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    cells, _ = tmp.encode_batch(V)
    V = V[cells == 0]
    n0 = len(V)
    assert n0 > 10240
    tmp.add_vectors(V)
    near = tmp.search_batch(q0[None, :], 40)[0][0]
    rest = np.setdiff1d(np.arange(n0), near)
    order = np.empty(n0, np.int64)
    slots = 7 + 256 * np.arange(40)
    order[slots] = near
    order[np.setdiff1d(np.arange(n0), slots)] = rest
    far = p["coarse"][1] + 0.25 * rng.standard_normal((300, D))  # (the other cell's list, appended behind: ids n0 ...)
    far = far[tmp.encode_batch(far)[0] == 1]
    Vo = np.concatenate([V[order], far])
    ix = mi.IVFPQ(D, len(Vo), False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    ix.setW(w)
    ix.set_option("passa_q", 1)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    ix.indexVectors([str(i) for i in range(len(Vo))], Vo)
    ref.add_vectors(Vo)
    got = ix.search_batch(k, Q)
    assert ix.get_dispatch()["pass_a"].startswith("K3q")
    assert_same(got, ref.search_batch(Q, k))
    # (the construction holds: the queries' best 21 sit on lane 7's positions)
    assert np.all(np.isin(got[0][:, :12], slots))
    ix.close()


@pytest.mark.parametrize("name", ["ivfpq_small.npz", "ivfpq_perm.npz", "ivfpq_ties.npz", "pq_small.npz"])
def test_golden_fixtures_gpu(mi, name):
    """HIP path against the COMMITTED answers of tests/golden/ (no live oracle in the loop): encode output, neighbour
    ids and distance bits.  ivfpq_ties is the flagged tie fixture (every vector three times, ties straddle k)."""
    import os

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))
    D, m, ks, k = int(z["D"]), int(z["m"]), int(z["ks"]), int(z["k"])
    n = len(z["base"])
    if "coarse" in z.files:
        tr = {0: mi.TransformationType.None_, 2: mi.TransformationType.RandomPermutation}[int(z["transform"])]
        ix = mi.IVFPQ(D, n, False, "", m, ks, tr, int(z["C"]), 512)
        ix.loadCoarseQuantizer(z["coarse"])
        ix.loadProductQuantizer(z["pq"])
        ix.setW(int(z["w"]))
        cells, codes = ix.encode(z["base"])
        assert np.array_equal(cells, z["cells"])
    else:
        ix = mi.PQ(D, n, False, "", m, ks, 0, 512)
        ix.loadProductQuantizer(z["pq"])
        cells, codes = ix.encode(z["base"])
    assert np.array_equal(codes.astype(np.int32) + (128 if ks <= 256 else 0), z["codes"])
    ix.indexVectors([str(i) for i in range(n)], z["base"])
    iids, dists, counts = ix.search_batch(k, z["queries"])
    assert np.array_equal(counts, z["counts"]) and np.array_equal(iids, z["ids"]) and np.array_equal(dists, z["dists"])
    ix.close()


def test_pass_b_tail_kernel(mi, oracle):
    """Pass B's launch is sized from the item count of the previous call; what lies beyond it is handled by a small
    grid of looping blocks (k_scan_filt_tail).  Force almost everything through that tail: same answers."""
    D, C, m, ks, n, w = 32, 64, 8, 256, 30000, 32
    p = synth.make_ivfpq_problem(n=8000, D=D, C=C, m=m, ks=ks, nq=40, seed=77)
    base, _ = synth.mixture(n, D, C, sigma=0.5, seed=78)
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    ix.setW(w)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    rng = np.random.default_rng(2)
    q = base[rng.choice(n, 40, replace=False)] + 0.2 * rng.standard_normal((40, D))
    ix.set_option("no_bound", 1)  # keep every probe: plenty of pass-B items
    want = ref.search_batch(q, 20)
    for g in (8, 64, 0, 0):  # 0 = sized from the hint of the previous call (first large, then fitted)
        ix.set_option("passb_main_grid", g)
        assert_same(ix.search_batch(20, q), want)
    ix.close()


@pytest.mark.parametrize("n,D,k", [(10000, 128, 10), (40000, 32, 5), (300, 16, 400), (17000, 24, 100)])
def test_linear_exact_search(mi, oracle, n, D, k):
    """BASELINE config 1 (Linear, 10k x 128, k = 10) through the native path: ids and distance bits equal the oracle's
    Linear restatement (Linear.java:138-163); n <= 16384 runs the certified matrix-core filter, larger n the plain exact
    kernels; k > n returns n results; duplicated vectors exercise the queue's tie order."""
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, D))
    X[n // 2:n // 2 + 30] = X[:30]  # exact duplicates
    ix = mi.Linear(D, n + 5)
    ix.indexVectors([f"v{i}" for i in range(n)], X)
    assert ix.size() == n and np.array_equal(ix.getVector(7), X[7])
    Q = np.concatenate([X[rng.choice(n, 20, replace=False)] + 0.05 * rng.standard_normal((20, D)), X[:4], rng.standard_normal((8, D))])
    iids, dists, counts = ix.search_batch(k, Q)
    for qi, q in enumerate(Q):
        rid, rd = oracle.linear_search(X, q, k)
        assert counts[qi] == len(rid) == min(k, n)
        assert np.array_equal(iids[qi, :counts[qi]], rid), qi
        assert np.array_equal(dists[qi, :counts[qi]], rd), qi
    ans = ix.computeNearestNeighbors(3, "v11")  # by id: the stored vector is the query (Linear.java:181-186)
    assert ans.getIds()[0] in ("v11", f"v{n // 2 + 11}") and ans.getDistances()[0] == 0.0
    with pytest.raises(mi.MmidxError):
        ix.indexVector("bad", np.zeros(D + 1))
    ix.close()


def test_linear_single_query_callers_are_combined(mi, oracle):
    """Linear.computeNearestNeighbors from many threads, one query per call (the reference's usage): the callers are
    served together as one device batch and every one of them gets the answer of its own query."""
    import threading

    n, D = 3000, 32
    rng = np.random.default_rng(77)
    X = rng.standard_normal((n, D))
    ix = mi.Linear(D, n)
    ix.indexVectors([f"v{i}" for i in range(n)], X)
    Q = X[rng.choice(n, 64, replace=False)] + 0.05 * rng.standard_normal((64, D))
    want = {k: [oracle.linear_search(X, q, k) for q in Q] for k in (1, 10)}
    errors = []
    start = threading.Barrier(16)

    def worker(t):
        try:
            start.wait()
            k = (1, 10)[t % 2]
            for rep in range(30):
                q0 = (t * 5 + rep * 3) % 60
                nq = 1 if rep % 4 else 3
                iids, dists, counts = ix.search_batch(k, Q[q0:q0 + nq])
                for j in range(nq):
                    rid, rd = want[k][q0 + j]
                    if counts[j] != len(rid) or not np.array_equal(iids[j, :counts[j]], rid) or not np.array_equal(dists[j, :counts[j]], rd):
                        errors.append((t, rep, k, q0, j))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    ix.close()


def test_concurrent_reader_threads(mi, oracle):
    """computeNearestNeighbors is not synchronized in the reference (ASS:281-291): several threads may query one index.
    The native host-pointer search queues them on a per-handle lock; every thread must get its own right answer."""
    import threading

    D, C, m, ks, n, w, k = 32, 16, 8, 256, 8000, 5, 10
    p = synth.make_ivfpq_problem(n=n, D=D, C=C, m=m, ks=ks, nq=64, seed=91)
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    ix.setW(w)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    ix.indexVectors([str(i) for i in range(n)], p["base"])
    ref.add_vectors(p["base"])
    want = ref.search_batch(p["queries"], k)
    errors = []

    def worker(t):
        try:
            for rep in range(6):
                sl = slice((t * 7 + rep * 5) % 48, (t * 7 + rep * 5) % 48 + 16)
                got = ix.search_batch(k, p["queries"][sl])
                for a, b in zip(got, want):
                    if not np.array_equal(a, b[sl]):
                        errors.append((t, rep))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    ix.close()


@pytest.mark.parametrize("slots", [1, 0])
def test_large_host_requests_of_several_callers_overlap(mi, oracle, slots):
    """Requests of more than 4096 queries go around the combiner (search_host_big): up to three callers in flight, each with its own
    copy stream and buffers, the kernels one caller at a time.  Five threads with requests of different sizes, small calls mixed in:
    every caller gets exactly its own answers -- also with the slots switched off (one request at a time, rounds 1-5)."""
    import threading

    D, C, m, ks, n, w, k = 32, 16, 8, 256, 8000, 5, 10
    p = synth.make_ivfpq_problem(n=n, D=D, C=C, m=m, ks=ks, nq=96, seed=93)
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    ix.setW(w)
    ix.set_option("host_slots", slots)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    ix.indexVectors([str(i) for i in range(n)], p["base"])
    ref.add_vectors(p["base"])
    want = ref.search_batch(p["queries"], k)
    errors = []
    start = threading.Barrier(5)

    def worker(t):
        try:
            start.wait()
            for rep in range(4):
                nq = 4200 + 450 * ((t + rep) % 5)
                sel = (np.arange(nq) * (t + 3) + rep) % 96
                got = ix.search_batch(k, np.ascontiguousarray(p["queries"][sel]))
                for a, b in zip(got, want):
                    if not np.array_equal(a, b[sel]):
                        errors.append((t, rep, nq))
                got = ix.search_batch(k, p["queries"][t:t + 2])  # (a small call through the combiner in between)
                for a, b in zip(got, want):
                    if not np.array_equal(a, b[t:t + 2]):
                        errors.append((t, rep, "small"))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(5)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    ix.close()


@pytest.mark.parametrize("combine", [1, 0])
def test_single_query_callers_are_combined(mi, oracle, combine):
    """The reference's API is one query per call from many reader threads.  mmidx_search serves the callers that arrive
    while a batch is running TOGETHER (one staged copy in, one search, one copy out, results scattered); every caller
    must get exactly the answer of its own query, also when calls with different k and sizes are mixed, and with
    combining switched off."""
    import threading

    D, C, m, ks, n, w = 32, 16, 8, 256, 8000, 5
    p = synth.make_ivfpq_problem(n=n, D=D, C=C, m=m, ks=ks, nq=96, seed=92)
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    ix.setW(w)
    ix.set_option("combine", combine)
    ref = oracle_ivfpq(oracle, p, D, m, ks, C, w)
    ix.indexVectors([str(i) for i in range(n)], p["base"])
    ref.add_vectors(p["base"])
    want = {k: ref.search_batch(p["queries"], k) for k in (1, 10, 37)}
    errors = []
    start = threading.Barrier(24)

    def worker(t):
        try:
            start.wait()
            k = (1, 10, 10, 37)[t % 4]
            for rep in range(40):
                q0 = (t * 11 + rep * 7) % 90
                nq = 1 if (t + rep) % 5 else 1 + (rep % 6)
                got = ix.search_batch(k, p["queries"][q0:q0 + nq])
                for a, b in zip(got, want[k]):
                    if not np.array_equal(a, b[q0:q0 + nq]):
                        errors.append((t, rep, k, q0, nq))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(24)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    # a failing call reports its own error and does not disturb the others
    with pytest.raises(Exception):
        ix.search_batch(5000, p["queries"][:1])
    got = ix.search_batch(10, p["queries"][:3])
    for a, b in zip(got, want[10]):
        assert np.array_equal(a, b[:3])
    ix.close()


@pytest.mark.parametrize("D,C", [(32, 300), (128, 1000), (130, 260), (24, 5000)])
def test_assignment_kernel_ties_and_padding(mi, oracle, D, C):
    """computeNearestCoarseIndex (IVFPQ.java:547-564) through the bf16-split certified assignment: duplicated centroids
    (exact ties: the first index must win), vectors sitting on centroids, C / D not multiples of the tile sizes; the
    flagged vectors go through the exact kernel.  Cells must equal the oracle's for every vector."""
    rng = np.random.default_rng(C + D)
    coarse = rng.standard_normal((C, D))
    coarse[C // 2:C // 2 + 25] = coarse[10:35]          # duplicates of earlier rows
    coarse[C - 3] = coarse[C - 4] + 1e-13                # a near tie far below any fp32 / bf16 resolution
    m = 2
    pq = rng.standard_normal((m, 16, D // m))
    ix = mi.IVFPQ(D, 10, False, "", m, 16, 0, C, 512)
    ix.loadCoarseQuantizer(coarse)
    ix.loadProductQuantizer(pq)
    ref = oracle.OracleIndex(oracle.KIND_IVFPQ, D, m, 16, C)
    ref.set_coarse(coarse)
    ref.set_pq(pq)
    X = np.concatenate([coarse[10:35], coarse[C // 2:C // 2 + 25] + 1e-12, coarse[[C - 3, C - 4]], 0.5 * (coarse[C - 3] + coarse[C - 4])[None],
                        coarse[rng.integers(0, C, 3000)] + 0.3 * rng.standard_normal((3000, D)), 3.0 * rng.standard_normal((500, D))])
    cells, _ = ix.encode(X)
    rcells, _ = ref.encode_batch(X)
    assert np.array_equal(cells, rcells)
    assert np.all(cells[:25] == np.arange(10, 35))       # on a duplicated centroid: the lower index
    ix.close()


@pytest.mark.parametrize("scale", [1e25, 1e17, 3e17, 1.5e18, 6e18])
def test_coarse_stage_huge_magnitudes(mi, oracle, scale):
    """Coordinates near and beyond the fp32 range: at 1e25 the matrix-core filter overflows to inf / NaN everywhere, around
    1e18 only some of its fp32 quantities do (squared norms ~1e37..1e39) -- it must hand every such query to the exact path
    instead of certifying garbage (at 1e17 everything is still finite and the certified path runs); the encoder's
    assignment flags such vectors for the exact kernel."""
    rng = np.random.default_rng(4)
    D, C, w = 32, 1200, 6
    coarse = scale * rng.standard_normal((C, D))
    pq = rng.standard_normal((2, 16, D // 2))
    ix = mi.IVFPQ(D, 10, False, "", 2, 16, 0, C, 512)
    ix.loadCoarseQuantizer(coarse)
    ix.loadProductQuantizer(pq)
    ix.setW(w)
    ref = oracle.OracleIndex(oracle.KIND_IVFPQ, D, 2, 16, C)
    ref.set_coarse(coarse)
    ref.set_pq(pq)
    Q = coarse[rng.integers(0, C, 12)] * (1.0 + 1e-3 * rng.standard_normal((12, D)))
    got = _coarse_cells(mi, ix, Q)
    exp = np.stack([ref.nearest_coarse(q, w) for q in Q])
    assert np.array_equal(got, exp)
    cells, _ = ix.encode(Q)
    rcells, _ = ref.encode_batch(Q)
    assert np.array_equal(cells, rcells)
    ix.close()


def test_randomised_differential_fuzz():
    """tests/fuzz_parity.py: random small PQ / IVFPQ configurations (shapes, transforms, duplicates, k up to 600, forced
    kernel variants, adds interleaved with a search; round 3: m = 64, lookup tables beyond the LDS -- ks = 700 with m >= 32 --,
    k = 384 and 4095, rotation compared bit for bit) through the HIP path and the oracle; 150 cases run here."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "fuzz_parity.py"), "150", "11"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("D,m", [(64, 8), (256, 16)])
@pytest.mark.parametrize("scale,qscale", [(1e-25, 1.0), (1e-9, 1.0), (1e9, 1.0), (1e25, 1.0), (1.0, 1e6), (1e-160, 1.0), (1e140, 1.0)])
def test_mfma_pass_b_magnitudes(mi, oracle, scale, qscale, D, m):
    """K3m scales residuals and codebook by powers of two before the fp16 rounding (per item / per index): data 25 orders of
    magnitude away from 1, queries far outside the data (residuals a million times the codebook's scale: the exponent difference
    between the two scales is limited, such queries go to the redo), and magnitudes whose squares leave fp32 or fp64's normal range.
    The bound must never drop a true neighbour: ids and distance bits are the oracle's.  D = 256: K3mk, whose residual scale is one
    power of two per LAUNCH (from the largest centroid and query elements): one far query among ordinary ones coarsens every row's
    scale, the certificate's subnormal term grows with it and the survivors are verified or redone -- never dropped."""
    C, n, w, k, ks = 6, 12000 if D == 64 else 6000, 6, 20, 256
    rng = np.random.default_rng(11)
    mu = 0.5 * rng.standard_normal((C, D))
    base = mu[rng.integers(0, C, n)] + rng.standard_normal((n, D))
    ds = D // m
    pq = np.stack([synth.kmeans((mu[rng.integers(0, C, 3000)] - base[:3000])[:, s * ds:(s + 1) * ds], ks, iters=2, seed=s) for s in range(m)])
    mu, base, pq = mu * scale, base * scale, pq * scale
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
    ix.loadCoarseQuantizer(mu)
    ix.loadProductQuantizer(pq)
    ix.setW(w)
    ref = oracle_ivfpq(oracle, {"coarse": mu, "pq": pq}, D, m, ks, C, w)
    ix.indexVectors([str(i) for i in range(n)], base)
    ref.add_vectors(base)
    Q = np.concatenate([0.5 * (base[:24] + base[100:124]), base[:24] + 0.01 * scale * rng.standard_normal((24, D))]) * qscale
    assert_same(ix.search_batch(k, Q), ref.search_batch(Q, k))
    if qscale == 1.0:  # one query a million times the data's scale in the same call (K3mk: it sets the launch's residual scale)
        Q2 = np.concatenate([Q[:40], 1e6 * Q[40:41], 1e-6 * Q[41:42]])
        assert_same(ix.search_batch(k, Q2), ref.search_batch(Q2, k))
    ix.close()
