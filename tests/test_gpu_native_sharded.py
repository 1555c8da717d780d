"""GPU parity of the single-process multi-GPU index (mmidx_create_sharded, include/mmidx.h ABI 5) against the CPU oracle.

One box has one GPU, so two configurations stand in for the 8-GPU node:
  * devices = [0]            one shard, collectives on REAL RCCL (a 1-rank communicator made by ncclCommInitAll);
  * devices = [0, 0(, 0)]    virtual shards on one device, in-process collectives (RCCL refuses duplicate devices); the
                              partial lists still travel through the owners' receive buffers and K5.
Both must return the single queue's answer (IVFPQ.java:408-450), flagged tie fixtures included.
"""
import ctypes as C
import importlib
import threading

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mi():
    try:
        import torch

        torch.cuda.init()
    except Exception:
        pass
    m = importlib.import_module("multimedia-indexing_amd")
    if m.lib().mmidx_device_count() < 1:
        pytest.fail("libmmidx_hip.so found no HIP device: GPU tests must run the native path")
    return m


def make_ref(o, p, D, m, ks, C_, w, tr=0):
    ref = o.OracleIndex(o.KIND_IVFPQ, D, m, ks, C_, transform=tr)
    ref.set_coarse(p["coarse"])
    ref.set_pq(p["pq"])
    ref.set_w(w)
    return ref


def make_sharded(mi, p, D, m, ks, C_, w, n, devices, tr=0):
    ix = mi.IVFPQ(D, n, False, "", m, ks, tr, C_, 512, devices=devices)
    ix.loadCoarseQuantizer(p["coarse"])
    ix.loadProductQuantizer(p["pq"])
    ix.setW(w)
    return ix


def assert_same(res, ref):
    iids, dists, counts = res
    rid, rd, rc = ref
    assert np.array_equal(counts, rc)
    assert np.array_equal(iids, rid)
    assert np.array_equal(dists, rd)  # bit-equal (the promise to callers is 1e-5)


DEVS = [[0], [0, 0], [0, 0, 0]]


@pytest.mark.parametrize("devices", DEVS, ids=["rccl1", "virt2", "virt3"])
def test_native_sharded_index_and_search(mi, oracle, devices):
    D, C_, m, ks, n, w = 32, 24, 8, 256, 9000, 6
    p = synth.make_ivfpq_problem(n=n, D=D, C=C_, m=m, ks=ks, nq=41, seed=11 + len(devices))
    ref = make_ref(oracle, p, D, m, ks, C_, w)
    ref.add_vectors(p["base"])
    ix = make_sharded(mi, p, D, m, ks, C_, w, n, devices)
    L = mi.lib()
    ns, dev, sz, rc_ = C.c_int(), C.c_int(), C.c_int64(), C.c_int()
    assert L.mmidx_shard_count(ix._h, C.byref(ns)) == 0 and ns.value == len(devices)
    # encode parity through the sharded handle (rows split over the shards)
    cells, codes = ix.encode(p["base"][:1501])
    rcell, rcode = ref.encode_batch(p["base"][:1501])
    assert np.array_equal(cells, rcell) and np.array_equal(codes.astype(np.int32) + 128, rcode)
    assert ix.indexVectors([str(i) for i in range(n)], p["base"]) == n
    assert ix.size() == n
    assert np.array_equal(ix.listSizes(), ref.list_sizes())
    tot = 0
    for r in range(len(devices)):
        assert L.mmidx_shard_info(ix._h, r, C.byref(dev), C.byref(sz), C.byref(rc_)) == 0
        assert dev.value == 0 and rc_.value == (1 if len(devices) == 1 else 0)
        tot += sz.value
    assert tot == n
    for k in (1, 10, 100):
        assert_same(ix.search_batch(k, p["queries"]), ref.search_batch(p["queries"], k))
    # ragged sizes: fewer queries than shards, one query (the reference's own call shape)
    for nq in (1, 2, 5):
        assert_same(ix.search_batch(7, p["queries"][:nq]), ref.search_batch(p["queries"][:nq], 7))
    a = ix.computeNearestNeighbors(10, p["queries"][3])
    rid, rd = ref.search(p["queries"][3], 10)
    assert a.getIds() == [str(i) for i in rid] and np.array_equal(a.getDistances(), rd)
    # the snapshot is list-major over all shards, identical to a plain handle's
    off, iids, cds = ix.export()
    plain = mi.IVFPQ(D, n, False, "", m, ks, 0, C_, 512)
    plain.loadCoarseQuantizer(p["coarse"])
    plain.loadProductQuantizer(p["pq"])
    plain.setW(w)
    plain.indexVectors([str(i) for i in range(n)], p["base"])
    poff, piids, pcds = plain.export()
    assert np.array_equal(off, poff) and np.array_equal(iids, piids) and np.array_equal(cds, pcds)
    # per-id utilities (getInvertedListId, getPQCodeByte, computeDistanceIVFADC) find the shard that holds the id
    for id_ in ("0", "17", str(n - 1)):
        assert ix.getInvertedListId(id_) == plain.getInvertedListId(id_)
        assert np.array_equal(ix.getPQCodeByte(id_), plain.getPQCodeByte(id_))
        assert ix.computeDistanceIVFADC(p["queries"][1], id_) == plain.computeDistanceIVFADC(p["queries"][1], id_)
    with pytest.raises(mi.MmidxError):
        ix.getPQCodeByte("nope")
    plain.close()
    ix.close()


@pytest.mark.parametrize("devices,exchange", [([0], 0), ([0], 1), ([0, 0], 0), ([0, 0, 0], 0)], ids=["rccl1", "rccl1-sendrecv", "virt2", "virt3"])
def test_native_sharded_straddling_ties(mi, oracle, devices, exchange):
    """FLAGGED tie fixture: every vector indexed three times, so exact distance ties straddle k and the answer depends on the
    offer order over ALL probed lists (IVFPQ.java:445); tie_slots = 2 forces several replay rounds per batch."""
    D, C_, m, ks, w = 32, 12, 8, 256, 5
    p = synth.make_ivfpq_problem(n=1200, D=D, C=C_, m=m, ks=ks, nq=48, seed=60 + len(devices))
    base = np.concatenate([p["base"]] * 3)
    base = base[np.random.default_rng(2).permutation(len(base))]
    n = len(base)
    ref = make_ref(oracle, p, D, m, ks, C_, w)
    ref.add_vectors(base)
    ix = make_sharded(mi, p, D, m, ks, C_, w, n, devices)
    ix.set_option("shard_exchange", exchange)
    ix.indexVectors([str(i) for i in range(n)], base)
    tied = 0
    for slots in (32, 2):
        ix.set_option("tie_slots", slots)
        for k in (1, 5, 30):
            rid, rd, rc = ref.search_batch(p["queries"], k)
            _, rd1, rc1 = ref.search_batch(p["queries"], k + 1)
            tied += sum(int(rc1[q] > k and rd1[q, k - 1] == rd1[q, k]) for q in range(len(rc)))
            assert_same(ix.search_batch(k, p["queries"]), (rid, rd, rc))
    assert tied >= 40
    ix.close()


@pytest.mark.parametrize("devices", [[0], [0, 0]], ids=["rccl1", "virt2"])
def test_native_sharded_load_index_and_transform(mi, oracle, devices):
    """indexPQCode / loadIndexInMemory through the sharded handle (records routed by list id), RandomPermutation on."""
    D, C_, m, ks, n, w, k = 32, 16, 8, 256, 5000, 5, 20
    p = synth.make_ivfpq_problem(n=n, D=D, C=C_, m=m, ks=ks, nq=32, seed=79)
    ref = make_ref(oracle, p, D, m, ks, C_, w, tr=2)
    ref.add_vectors(p["base"])
    enc = make_sharded(mi, p, D, m, ks, C_, w, n, devices, tr=2)
    cells, codes = enc.encode(p["base"])
    ix = make_sharded(mi, p, D, m, ks, C_, w, n, devices, tr=2)
    ix.loadIndex(np.arange(3000, dtype=np.int32), cells[:3000], codes[:3000])
    for i in range(3000, 3010):
        assert ix.indexPQCode(str(i), int(cells[i]), codes[i])
    ix.loadIndex(np.arange(3010, n, dtype=np.int32), cells[3010:], codes[3010:])
    assert np.array_equal(ix.listSizes(), ref.list_sizes())
    assert_same(ix.search_batch(k, p["queries"]), ref.search_batch(p["queries"], k))
    # a bad record rejects the whole batch and leaves the index usable
    bad_cells = cells[:4].copy()
    bad_cells[2] = C_
    with pytest.raises(mi.MmidxError):
        ix.loadIndex(np.arange(n, n + 4, dtype=np.int32), bad_cells, codes[:4])
    assert ix.size() == n
    assert_same(ix.search_batch(k, p["queries"][:5]), ref.search_batch(p["queries"][:5], k))
    enc.close()
    ix.close()


@pytest.mark.parametrize("devices,tr", [([0], 0), ([0, 0], 2), ([0, 0, 0], 2)], ids=["rccl1", "virt2-perm", "virt3-perm"])
def test_native_sharded_smin_prefilter(mi, oracle, devices, tr):
    """K3s (the certified Smin of every far pair in front of pass B's sort) inside every shard of a sharded handle:
    the option goes to every shard, each shard keeps its own transformed copy of the centroids and its own hint words.  Forced on
    (with and without the bf16 first stage of 16-dimensional sub-quantizers), off, and hint-driven over three calls: the oracle's ids
    and distance bits every time."""
    D, C_, m, ks, n, w, k = 128, 24, 8, 256, 24000, 12, 20  # (dsub = 16: the bf16 stage runs)
    rng = np.random.default_rng(5)
    mu = 0.8 * rng.standard_normal((C_, D))
    base = mu[rng.integers(0, C_, n)] + 0.5 * rng.standard_normal((n, D))
    pq = np.stack([synth.kmeans((mu[rng.integers(0, C_, 2000)] - base[:2000])[:, s * (D // m):(s + 1) * (D // m)], ks, iters=1, seed=s) for s in range(m)])
    p = {"coarse": mu, "pq": pq}
    ref = make_ref(oracle, p, D, m, ks, C_, w, tr=tr)
    ref.add_vectors(base)
    ix = make_sharded(mi, p, D, m, ks, C_, w, n, devices, tr=tr)
    ix.indexVectors([str(i) for i in range(n)], base)
    Q = np.concatenate([base[:40] + 0.01 * rng.standard_normal((40, D)), 0.5 * (base[100:116] + base[200:216]), mu[:4]])
    want = ref.search_batch(Q, k)
    for mode, b16 in ((1, 1), (1, 0), (0, 1), (-1, 1), (-1, 1), (-1, 1)):
        ix.set_option("smin_pre", mode)
        ix.set_option("smin_bf16", b16)
        assert_same(ix.search_batch(k, Q), want)
    ix.close()


@pytest.mark.parametrize("S,m,route_host", [(3, 8, 0), (3, 8, 1), (4, 5, 0), (8, 16, 0)])
def test_native_sharded_device_side_routing(mi, oracle, S, m, route_host):
    """mmidx_add_vectors_sliced_device keeps the records on the devices (round 5): every shard picks its own (cell mod S) out of all
    slices in batch order -- count, scan, stable scatter, mmidx_add_codes_device.  Uneven and empty slices, slices that straddle the
    4096-record blocks, 5-byte codes (the byte-wise copy) and 16-byte ones (the 16-byte copy); the lists -- ids in arrival order,
    codes -- equal those of a plain handle fed the same batches, and so do the search results.  `shard_route_host` = 1: the host path."""
    import torch

    L, nat = mi.lib(), importlib.import_module("multimedia-indexing_amd._native")
    D, C_, ks, n, w, k = 80, 37, 256, 21000, 5, 10
    p = synth.make_ivfpq_problem(n=n, D=D, C=C_, m=m, ks=ks, nq=30, seed=9, iters=2)
    plain = mi.IVFPQ(D, n, False, "", m, ks, 0, C_, 512)
    plain.loadCoarseQuantizer(p["coarse"])
    plain.loadProductQuantizer(p["pq"])
    plain.setW(w)
    plain.indexVectors([str(i) for i in range(n)], p["base"])
    ix = make_sharded(mi, p, D, m, ks, C_, w, n, [0] * S)
    ix.set_option("shard_route_host", route_host)
    X = torch.tensor(p["base"], dtype=torch.float64, device="cuda")
    rng = np.random.default_rng(S)
    for i0, i1 in ((0, 9000), (9000, 9001), (9001, n)):
        cuts = np.sort(np.concatenate([[i0, i1], rng.integers(i0, i1 + 1, S - 1)])).astype(np.int64)
        if i1 - i0 > 5000:
            cuts[1] = cuts[0]  # an empty slice
        parts = [X[cuts[r]:cuts[r + 1]].contiguous() if cuts[r + 1] > cuts[r] else X[:1].contiguous() for r in range(S)]
        ns = (C.c_int64 * S)(*[int(cuts[r + 1] - cuts[r]) for r in range(S)])
        ptrs = (C.c_void_p * S)(*[t.data_ptr() for t in parts])
        torch.cuda.synchronize()
        nat.check(L.mmidx_add_vectors_sliced_device(ix._h, ns, ptrs, int(i0)))
    a, b = plain.export(), ix.export()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert_same(ix.search_batch(k, p["queries"]), plain.search_batch(k, p["queries"]))
    ix.close()
    plain.close()


@pytest.mark.parametrize("S", [1, 2])
def test_native_sharded_sliced_device_entry_points(mi, oracle, S):
    """The device-resident forms: slice r of the batch lives in the HBM of shard r's device (torch tensors as plumbing)."""
    import torch

    L, nat = mi.lib(), importlib.import_module("multimedia-indexing_amd._native")
    D, C_, m, ks, n, w, k = 32, 24, 8, 256, 8000, 6, 10
    p = synth.make_ivfpq_problem(n=n, D=D, C=C_, m=m, ks=ks, nq=40 * S, seed=5)
    ref = make_ref(oracle, p, D, m, ks, C_, w)
    ref.add_vectors(p["base"])
    ix = make_sharded(mi, p, D, m, ks, C_, w, n, [0] * S)
    # build: the batch is the concatenation of the slices, row i gets iid0 + i
    X = torch.tensor(p["base"], dtype=torch.float64, device="cuda")
    half = n // 2
    for i0, i1 in ((0, half), (half, n)):
        cuts = np.linspace(i0, i1, S + 1).astype(np.int64)
        parts = [X[cuts[r]:cuts[r + 1]].contiguous() for r in range(S)]
        ns = (C.c_int64 * S)(*[int(cuts[r + 1] - cuts[r]) for r in range(S)])
        ptrs = (C.c_void_p * S)(*[t.data_ptr() for t in parts])
        torch.cuda.synchronize()
        nat.check(L.mmidx_add_vectors_sliced_device(ix._h, ns, ptrs, int(i0)))
    assert np.array_equal(ix.listSizes(), ref.list_sizes())
    per = 40
    Q = torch.tensor(p["queries"], dtype=torch.float64, device="cuda")
    qs = [Q[r * per:(r + 1) * per].contiguous() for r in range(S)]
    oi = [torch.full((per, k), -7, dtype=torch.int32, device="cuda") for _ in range(S)]
    od = [torch.zeros((per, k), dtype=torch.float64, device="cuda") for _ in range(S)]
    oc = [torch.zeros(per, dtype=torch.int32, device="cuda") for _ in range(S)]
    torch.cuda.synchronize()
    arr = lambda ts: (C.c_void_p * S)(*[t.data_ptr() for t in ts])
    nat.check(L.mmidx_search_sliced_device(ix._h, k, per, arr(qs), arr(oi), arr(od), arr(oc)))
    got = (torch.cat(oi).cpu().numpy(), torch.cat(od).cpu().numpy(), torch.cat(oc).cpu().numpy())
    assert_same(got, ref.search_batch(p["queries"], k))
    # small rounds: several collective rounds per call
    ix.set_option("shard_max_round", 16 * S)
    nat.check(L.mmidx_search_sliced_device(ix._h, k, per, arr(qs), arr(oi), arr(od), arr(oc)))
    got = (torch.cat(oi).cpu().numpy(), torch.cat(od).cpu().numpy(), torch.cat(oc).cpu().numpy())
    assert_same(got, ref.search_batch(p["queries"], k))
    # the one-device entry points are refused on a sharded handle
    assert L.mmidx_search_device(ix._h, k, per, qs[0].data_ptr(), oi[0].data_ptr(), od[0].data_ptr(), oc[0].data_ptr(), None) == nat.ERR_UNSUPPORTED
    ix.close()


def test_native_sharded_concurrent_single_query_callers(mi, oracle):
    """ASS.computeNearestNeighbors is unsynchronised: reader threads on one sharded handle, one query per call."""
    D, C_, m, ks, n, w, k = 32, 16, 8, 256, 6000, 4, 10
    p = synth.make_ivfpq_problem(n=n, D=D, C=C_, m=m, ks=ks, nq=96, seed=8)
    ref = make_ref(oracle, p, D, m, ks, C_, w)
    ref.add_vectors(p["base"])
    ix = make_sharded(mi, p, D, m, ks, C_, w, n, [0, 0])
    ix.indexVectors([str(i) for i in range(n)], p["base"])
    rid, rd, rc = ref.search_batch(p["queries"], k)
    errs = []

    def reader(t):
        try:
            for q in range(t, 96, 12):
                i, d, c = ix.search_batch(k, p["queries"][q:q + 1])
                assert c[0] == rc[q] and np.array_equal(i[0], rid[q]) and np.array_equal(d[0], rd[q])
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=reader, args=(t,)) for t in range(12)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    ix.close()


def test_native_sharded_rejects_what_it_cannot_do(mi):
    L, nat = mi.lib(), importlib.import_module("multimedia-indexing_amd._native")
    h = C.c_void_p()
    devs = (C.c_int * 1)(0)
    assert L.mmidx_create_sharded(nat.KIND_PQ, 32, 8, 256, 0, 0, None, None, 1, devs, C.byref(h)) == nat.ERR_UNSUPPORTED
    bad = (C.c_int * 1)(99)
    assert L.mmidx_create_sharded(nat.KIND_IVFPQ, 32, 8, 256, 16, 0, None, None, 1, bad, C.byref(h)) == nat.ERR_NO_DEVICE
    assert L.mmidx_create_sharded(nat.KIND_IVFPQ, 32, 8, 256, 16, 0, None, None, 0, devs, C.byref(h)) == nat.ERR_INVALID_ARG
    assert L.mmidx_create_sharded(nat.KIND_IVFPQ, 32, 5, 256, 16, 0, None, None, 1, devs, C.byref(h)) == nat.ERR_INVALID_SUBVECTORS


def _ndev(mi):
    return int(mi.lib().mmidx_device_count())


@pytest.mark.parametrize("ndev,exchange", [(2, 0), (2, 1), (4, 0), (8, 0)], ids=["2gpu-p2p", "2gpu-sendrecv", "4gpu-p2p", "8gpu-p2p"])
def test_native_sharded_distinct_devices(mi, oracle, ndev, exchange):
    """The production form (`-Dmmidx.devices=0,...`): one shard per PHYSICAL device -- multi-rank ncclCommInitAll with a worker
    thread per device, K4's stores into peer HBM over xGMI (or ncclSend / ncclRecv), cross-device hipStreamWaitEvent, the cached
    table of the owners' buffers.  Skipped on boxes with fewer devices (the round's one-GPU boxes): the first node with several
    GPUs runs it, tie fixture included (every vector three times, tie_slots = 2: several replay rounds)."""
    if _ndev(mi) < ndev:
        pytest.skip(f"needs {ndev} HIP devices, this box has {_ndev(mi)}")
    D, C_, m, ks, w = 32, 40, 8, 256, 9
    p = synth.make_ivfpq_problem(n=3000, D=D, C=C_, m=m, ks=ks, nq=64, seed=90 + ndev)
    base = np.concatenate([p["base"]] * 3)
    base = base[np.random.default_rng(4).permutation(len(base))]
    n = len(base)
    ref = make_ref(oracle, p, D, m, ks, C_, w)
    ref.add_vectors(base)
    ix = make_sharded(mi, p, D, m, ks, C_, w, n, list(range(ndev)))
    ix.set_option("shard_exchange", exchange)
    ix.set_option("tie_slots", 2)
    assert ix.indexVectors([str(i) for i in range(n)], base) == n
    assert np.array_equal(ix.listSizes(), ref.list_sizes())
    Q = np.concatenate([p["queries"], base[:32]])
    for k in (1, 10, 100):
        assert_same(ix.search_batch(k, Q), ref.search_batch(Q, k))
    assert_same(ix.search_batch(5, Q[:1]), ref.search_batch(Q[:1], 5))
    # several rounds per call, and a second call on the warmed-up handle (cached destination table)
    ix.set_option("shard_max_round", 4 * ndev)
    assert_same(ix.search_batch(10, Q), ref.search_batch(Q, 10))
    assert_same(ix.search_batch(10, Q), ref.search_batch(Q, 10))
    ix.close()


@pytest.mark.parametrize("devices", [[0], [0, 0]], ids=["rccl1", "virt2"])
def test_native_sharded_concurrent_adds_number_their_vectors_once(mi, oracle, devices):
    """indexVector is `synchronized` in the reference (ASS:229): concurrent callers that let the library number their vectors
    (loadCounter, ASS:251) must get disjoint internal ids -- the total is read under the lock that serialises the adds."""
    D, C_, m, ks, w = 32, 16, 8, 256, 4
    p = synth.make_ivfpq_problem(n=4000, D=D, C=C_, m=m, ks=ks, nq=8, seed=5)
    ix = make_sharded(mi, p, D, m, ks, C_, w, 4000, devices)
    nat = importlib.import_module("multimedia-indexing_amd._native")
    L = mi.lib()
    parts = np.array_split(np.ascontiguousarray(p["base"]), 8)
    errs = []

    def add(X):
        X = np.ascontiguousarray(X)
        rc = L.mmidx_add_vectors(ix._h, len(X), X.ctypes.data, None, None, None)
        if rc:
            errs.append(rc)

    ts = [threading.Thread(target=add, args=(x,)) for x in parts]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs
    off, iids, cds = ix.export()
    assert len(iids) == 4000 and np.array_equal(np.sort(iids), np.arange(4000, dtype=np.int32))
    ix.close()


def test_native_sharded_search_right_after_a_large_add(mi, oracle):
    """The round size is planned on the workers AFTER their CSR rebuild: the first large search behind an add used to plan with the
    stale (empty) CSR on the caller's thread and fail with 'shard phases take at most ...' for large k * w."""
    D, C_, m, ks, w, k = 32, 8, 8, 256, 8, 1000
    p = synth.make_ivfpq_problem(n=30000, D=D, C=C_, m=m, ks=ks, nq=64, seed=21)
    ref = make_ref(oracle, p, D, m, ks, C_, w)
    ref.add_vectors(p["base"])
    ix = make_sharded(mi, p, D, m, ks, C_, w, 30000, [0, 0])
    ix.indexVectors([str(i) for i in range(30000)], p["base"])
    Q = np.concatenate([p["queries"]] * 40)[:2500]
    assert_same(ix.search_batch(k, Q), ref.search_batch(Q, k))
    ix.close()


@pytest.mark.parametrize("devices", DEVS, ids=["rccl1", "virt2", "virt3"])
def test_native_sharded_query_exchange_pipeline(mi, oracle, devices):
    """`shard_pipeline`: the rounds' query exchange on the shards' second streams and communicators -- next to the coarse stage of
    the own slice, and round i + 1's under round i's scans (two query buffers) -- against the one-stream form (0).  Calls of 1, 2,
    3 and 7 rounds (`shard_max_round`), host rows and device slices, ties included: the single queue's answer every time."""
    import torch

    D, C_, m, ks, w, k = 32, 20, 8, 256, 5, 10
    p = synth.make_ivfpq_problem(n=2500, D=D, C=C_, m=m, ks=ks, nq=32, seed=31 + len(devices))
    base = np.concatenate([p["base"]] * 2)
    n, W = len(base), len(devices)
    ref = make_ref(oracle, p, D, m, ks, C_, w)
    ref.add_vectors(base)
    ix = make_sharded(mi, p, D, m, ks, C_, w, n, devices)
    ix.indexVectors([str(i) for i in range(n)], base)
    Q = np.concatenate([p["queries"], base[:52]])  # 84 queries
    want = ref.search_batch(Q, k)
    L = mi.lib()
    for pl in (1, 0, 1):
        ix.set_option("shard_pipeline", pl)
        for max_round in (262144, 48, 30, 12):
            ix.set_option("shard_max_round", max_round)
            assert_same(ix.search_batch(k, Q), want)
            # device slices (mmidx_search_sliced_device): slice r = rows [r per, (r + 1) per) of the batch
            per = len(Q) // W
            dev = torch.device("cuda", 0)
            Qd = [torch.from_numpy(np.ascontiguousarray(Q[r * per:(r + 1) * per])).to(dev) for r in range(W)]
            oi = [torch.empty(per, k, dtype=torch.int32, device=dev) for _ in range(W)]
            od = [torch.empty(per, k, dtype=torch.float64, device=dev) for _ in range(W)]
            oc = [torch.empty(per, dtype=torch.int32, device=dev) for _ in range(W)]
            arr = lambda ts: (C.c_void_p * W)(*[t.data_ptr() for t in ts])
            torch.cuda.synchronize()
            rc = L.mmidx_search_sliced_device(ix._h, k, per, arr(Qd), arr(oi), arr(od), arr(oc))
            assert rc == 0, mi.lib().mmidx_last_error()
            torch.cuda.synchronize()
            got = (np.concatenate([t.cpu().numpy() for t in oi]), np.concatenate([t.cpu().numpy() for t in od]), np.concatenate([t.cpu().numpy() for t in oc]))
            assert_same(got, tuple(a[:per * W] for a in want))
    ix.close()


@pytest.mark.parametrize("D,m", [(64, 8), (256, 16)], ids=["k3m", "k3mk"])
@pytest.mark.parametrize("devices", DEVS, ids=["rccl1", "virt2", "virt3"])
def test_native_sharded_pass_b_on_the_matrix_cores(mi, oracle, devices, D, m):
    """A shape K3m takes (8-dimensional sub-quantizers, D = 64) and one K3mk takes (D = 256, 16 x 16: the chunked form with LDS-DMA, and
    the coarse stage's wave-per-query selection): every shard's pass B -- under the all-reduced thresholds, answers as sorted partial
    lists for the owners -- runs the matrix-core bound and its verification; overlapping cells so that far probes are scanned, every
    vector twice so that ties straddle shards.  The single queue's answer, with the bound and without."""
    C_, ks, w, k = 16, 256, 7, 20
    rng = np.random.default_rng(3 + len(devices))
    mu = 0.5 * rng.standard_normal((C_, D))
    half = mu[rng.integers(0, C_, 6000)] + rng.standard_normal((6000, D))
    base = np.concatenate([half, half])[rng.permutation(12000)]
    ds = D // m
    pq = np.stack([synth.kmeans((mu[rng.integers(0, C_, 3000)] - base[:3000])[:, s * ds:(s + 1) * ds], ks, iters=2, seed=s) for s in range(m)])
    p = {"coarse": mu, "pq": pq}
    ref = make_ref(oracle, p, D, m, ks, C_, w)
    ref.add_vectors(base)
    ix = make_sharded(mi, p, D, m, ks, C_, w, len(base), devices)
    ix.indexVectors([str(i) for i in range(len(base))], base)
    Q = np.concatenate([0.5 * (base[:30] + base[100:130]), base[:30] + 0.01 * rng.standard_normal((30, D))])
    want = ref.search_batch(Q, k)
    for off in (0, 1):
        ix.set_option("no_mfma", off)
        ix.set_profiling(True)
        assert_same(ix.search_batch(k, Q), want)
        st = ix.get_stats()
        assert (st["mfma_survivors"] > 0) == (off == 0)
    ix.close()


@pytest.mark.parametrize("devices", DEVS, ids=["rccl1", "virt2", "virt3"])
def test_native_sharded_pass_a_on_the_matrix_cores(mi, oracle, devices):
    """K3ma on every shard (option "passa_mfma" = 1; by default a shard takes it from 8 queries per list of the whole round's batch --
    the regime of the 8-GPU configuration): each shard sorts the queries whose nearest list it holds by cell, runs the two sweeps and
    the exact verification over them, the thresholds are MIN-reduced as after K3h.  Cells of ~1500 vectors, 300 queries (19 per list),
    every vector twice so that ties straddle shards; k + 1 = 101.  The single queue's answer, with K3ma and without."""
    D, m, C_, ks, w, k = 64, 8, 16, 256, 5, 100
    rng = np.random.default_rng(11 + len(devices))
    mu = 0.5 * rng.standard_normal((C_, D))
    half = mu[rng.integers(0, C_, 12000)] + rng.standard_normal((12000, D))
    base = np.concatenate([half, half])[rng.permutation(24000)]
    ds = D // m
    pq = np.stack([synth.kmeans((mu[rng.integers(0, C_, 3000)] - base[:3000])[:, s * ds:(s + 1) * ds], ks, iters=2, seed=s) for s in range(m)])
    p = {"coarse": mu, "pq": pq}
    ref = make_ref(oracle, p, D, m, ks, C_, w)
    ref.add_vectors(base)
    ix = make_sharded(mi, p, D, m, ks, C_, w, len(base), devices)
    ix.indexVectors([str(i) for i in range(len(base))], base)
    Q = np.concatenate([0.5 * (base[:40] + base[100:140]), base[200:440] + 0.01 * rng.standard_normal((240, D)), rng.standard_normal((20, D))])
    want = ref.search_batch(Q, k)
    for force in (1, 0):
        ix.set_option("passa_mfma", force)
        ix.set_profiling(True)
        assert_same(ix.search_batch(k, Q), want)
        st = ix.get_stats()
        assert (st["passa_mfma_launches"] > 0) == (force == 1)
    ix.close()
