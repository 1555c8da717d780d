import importlib, sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, synth
from oracle import oracle as o
try:
    import torch; torch.cuda.init()
except Exception: pass
mi = importlib.import_module("multimedia-indexing_amd")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-160
D, m, C, n, w, k, ks = 64, 8, 6, 12000, 6, 20, 256
rng = np.random.default_rng(11)
mu = 0.5 * rng.standard_normal((C, D))
base = mu[rng.integers(0, C, n)] + rng.standard_normal((n, D))
ds = D // m
pq = np.stack([synth.kmeans((mu[rng.integers(0, C, 3000)] - base[:3000])[:, s * ds:(s + 1) * ds], ks, iters=2, seed=s) for s in range(m)])
mu, base, pq = mu * scale, base * scale, pq * scale
ref = o.OracleIndex(o.KIND_IVFPQ, D, m, ks, C); ref.set_coarse(mu); ref.set_pq(pq); ref.set_w(w); ref.add_vectors(base)
Q = np.concatenate([0.5 * (base[:24] + base[100:124]), base[:24] + 0.01 * scale * rng.standard_normal((24, D))])
want = ref.search_batch(Q, k)
for opts in ([], [("passa_hist", 0)], [("no_grp", 1)], [("no_bound", 1)], [("no_filter", 1)], [("exact_coarse", 1)], [("passa_hist", 0), ("no_filter", 1), ("no_bound", 1)], [("passa_hist", 0), ("no_filter", 1), ("no_bound", 1), ("exact_coarse", 1)]):
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512); ix.loadCoarseQuantizer(mu); ix.loadProductQuantizer(pq); ix.setW(w)
    for a, b in opts: ix.set_option(a, b)
    ix.indexVectors([str(i) for i in range(n)], base)
    got = ix.search_batch(k, Q)
    cells_ok = np.array_equal(ix.listSizes(), ref.list_sizes())
    print(opts, "ids", np.array_equal(got[0], want[0]), "dist", np.array_equal(got[1], want[1]), "lists", cells_ok, "bad rows", int((got[0] != want[0]).any(1).sum()))
    ix.close()
