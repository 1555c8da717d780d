import importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from oracle import oracle
mi = importlib.import_module("multimedia-indexing_amd")
m, D, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ks, C, w, n = 256, 3, 3, 36000
p = synth.make_ivfpq_problem(n=6000, D=D, C=C, m=m, ks=ks, nq=8, seed=m + k)
base, _ = synth.mixture(n, D, C, sigma=0.3, seed=m)
ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
ix.loadCoarseQuantizer(p["coarse"]); ix.loadProductQuantizer(p["pq"]); ix.setW(w)
for a in sys.argv[4:]:
    kk, v = a.split("="); ix.set_option(kk, int(v))
ref = oracle.OracleIndex(oracle.KIND_IVFPQ, D, m, ks, C)
ref.set_coarse(p["coarse"]); ref.set_pq(p["pq"]); ref.set_w(w)
ix.indexVectors([str(i) for i in range(n)], base); ref.add_vectors(base)
rng = np.random.default_rng(1)
q = base[rng.choice(n, 10, replace=False)] + 0.01 * rng.standard_normal((10, D))
got = ix.search_batch(k, q); exp = ref.search_batch(q, k)
print(ix.get_dispatch())
print("pq absmax", np.abs(p["pq"]).max())
for i in range(10):
    same = np.array_equal(got[0][i], exp[0][i])
    print(i, "ok" if same else "DIFF", got[2][i], exp[2][i], got[1][i][:3], exp[1][i][:3], (got[0][i] != exp[0][i]).sum())

# each query alone (forced K3q) and stats
ix.set_option("passa_q", 1)
for i in range(10):
    g = ix.search_batch(k, q[i:i + 1])
    print("alone", i, "ok" if np.array_equal(g[0][0], exp[0][i]) else "DIFF", ix.get_dispatch()["pass_a"])
try:
    print(ix.get_stats())
except Exception as e:
    print("no stats", e)
