"""Debug aid: pass A's thresholds through K3ma against the exact kernels' (mmidx_shard_pass_a_device exports T after pass A),
and against the exact (k+1)-th smallest ADC distance of the nearest list from the oracle."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch

torch.cuda.init()
import synth

mi = importlib.import_module("multimedia-indexing_amd")
nat = importlib.import_module("multimedia-indexing_amd._native")
from oracle import oracle as o

D, m, C, n, w, k, ks, nq = 128, 16, 6, 30000, 3, 100, 256, 90
rng = np.random.default_rng(7 * D + m + k)
mu = 0.5 * rng.standard_normal((C, D))
base = mu[rng.integers(0, C, n)] + rng.standard_normal((n, D))
ds = D // m
pq = np.stack([synth.kmeans((mu[rng.integers(0, C, 3000)] - base[:3000])[:, s * ds:(s + 1) * ds], 256, iters=2, seed=s) for s in range(m)])
ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
ix.loadCoarseQuantizer(mu)
ix.loadProductQuantizer(pq)
ix.setW(w)
ix.indexVectors([str(i) for i in range(n)], base)
Q = base[200:200 + nq] + 0.01 * rng.standard_normal((nq, D))
dQ = torch.tensor(np.ascontiguousarray(Q), dtype=torch.float64, device="cuda")
cells = torch.empty(nq, w, dtype=torch.int32, device="cuda")
cd = torch.empty(nq, w, dtype=torch.float64, device="cuda")
nat.check(mi.lib().mmidx_coarse_device(ix._h, nq, dQ.data_ptr(), cells.data_ptr(), cd.data_ptr(), None))
torch.cuda.synchronize()
Ts = {}
for force, sub in ((0, 0), (1, 0), (1, 8192)):
    ix.set_option("passa_mfma", force)
    ix.set_option("mfma_sub", sub)
    T = torch.empty(nq, dtype=torch.float64, device="cuda")
    nat.check(mi.lib().mmidx_shard_pass_a_device(ix._h, k, nq, dQ.data_ptr(), cells.data_ptr(), T.data_ptr(), None))
    torch.cuda.synchronize()
    Ts[(force, sub)] = T.cpu().numpy()
# exact (k+1)-th smallest distance of the nearest list: oracle with w = 1, k + 1 results
ref = o.OracleIndex(o.KIND_IVFPQ, D, m, ks, C)
ref.set_coarse(mu)
ref.set_pq(pq)
ref.set_w(1)
ref.add_vectors(base)
_, rd, rc = ref.search_batch(Q, k + 1)
exact = rd[:, k]
t0 = Ts[(0, 0)]
for key in ((1, 0), (1, 8192)):
    t1 = Ts[key]
    bad = np.sum(t1 < exact)
    print(key, "T(K3ma) < exact K1-th:", bad, "of", nq, "| inf:", np.sum(~np.isfinite(t1)), "| median T/exact:", np.median(t1 / exact), "| exact kernels' T/exact:", np.median(t0 / exact))
    if bad:
        i = np.where(t1 < exact)[0][:8]
        print("  e.g. q", i, "T", t1[i], "exact", exact[i], "ratio", t1[i] / exact[i])

# which results are missing with K3ma forced?  (cells of the missing ids' lists by probe rank)
ref.set_w(w)
want = ref.search_batch(Q, k)
cl = cells.cpu().numpy()
from collections import Counter
for opts in ((("passa_mfma", 1), ("mfma_sub", 0)), (("passa_mfma", 1), ("mfma_sub", 0), ("no_mfma", 1)), (("passa_mfma", 1), ("mfma_sub", 8192), ("no_mfma", 0))):
    for a, b in opts:
        ix.set_option(a, b)
    got = ix.search_batch(k, Q)
    nbad = int(np.sum(got[2] != want[2]))
    print(opts, "queries with a wrong count:", nbad, "| ids equal:", np.array_equal(got[0], want[0]))
    # per missing id: which probe rank does its cell have?
    cell_of = ref.assign(base) if hasattr(ref, "assign") else None
    miss_rank = Counter()
    d2c = ((base[:, None, :] - mu[None, :, :]) ** 2).sum(-1).argmin(1)
    for qi in range(nq):
        g = set(got[0][qi, :got[2][qi]].tolist())
        for iid in want[0][qi, :want[2][qi]]:
            if iid not in g:
                r = np.where(cl[qi] == d2c[iid])[0]
                miss_rank[int(r[0]) if len(r) else -1] += 1
    print("   missing ids by probe rank of their cell:", dict(miss_rank))

ix.set_option("mfma_sub", 0)
ix.set_option("no_mfma", 0)
ix.set_option("passa_mfma", 1)
ix.set_w(1) if hasattr(ix, "set_w") else None
# expected survivors: codes of the nearest list with exact distance <= T (w = 1 oracle, k large)
ref.set_w(1)
_, rd1, rc1 = ref.search_batch(Q, 600)
t1 = Ts[(1, 0)]
exp = int(sum(np.sum(rd1[i, :rc1[i]] <= t1[i]) for i in range(nq)))
ix.set_profiling(True)
got = ix.search_batch(k, Q)
st = ix.get_stats()
print("expected codes <= T in the nearest lists:", exp, "| verified_codes:", st["verified_codes"], "| mfma_survivors:", st["mfma_survivors"], "| redo:", st["mfma_redo_queries"], "| K3ma launches:", st["passa_mfma_launches"])

# positions of the missing ids inside their list (arrival order = iid order within a cell)
pos_in_list = np.zeros(n, np.int64)
for c in range(C):
    idx = np.where(d2c == c)[0]
    pos_in_list[idx] = np.arange(len(idx))
miss_pos = []
have_pos = []
row_of_q = {}
for qi in range(nq):
    g = set(got[0][qi, :got[2][qi]].tolist())
    for iid in want[0][qi, :want[2][qi]]:
        if d2c[iid] != cl[qi][0]:
            continue
        (have_pos if iid in g else miss_pos).append((qi, int(pos_in_list[iid])))
mp = np.array([p for _, p in miss_pos]); hp = np.array([p for _, p in have_pos])
print("missing: pos mod 32 histogram", np.bincount(mp % 32, minlength=32))
print("present: pos mod 32 histogram", np.bincount(hp % 32, minlength=32))
print("missing: (pos // 16) mod 8", np.bincount((mp // 16) % 8, minlength=8), " present:", np.bincount((hp // 16) % 8, minlength=8))
print("missing by query:", np.bincount([q for q, _ in miss_pos], minlength=nq))
# distances of the present ones equal?
same = 0; diff = 0
for qi in range(nq):
    wd = dict(zip(want[0][qi, :want[2][qi]].tolist(), want[1][qi, :want[2][qi]].tolist()))
    for iid, dd in zip(got[0][qi, :got[2][qi]].tolist(), got[1][qi, :got[2][qi]].tolist()):
        if iid in wd:
            same += wd[iid] == dd; diff += wd[iid] != dd
print("present ids: distances equal", same, "different", diff)
