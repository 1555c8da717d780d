import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth
from oracle import oracle as o
mi = importlib.import_module("multimedia-indexing_amd")
D, C, m, ks, n, w, k = [int(x) for x in (sys.argv[1:8] if len(sys.argv) > 7 else (32, 16, 8, 256, 5000, 4, 10))]
p = synth.make_ivfpq_problem(n=n, D=D, C=C, m=m, ks=ks, nq=64, seed=D + C)
ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
ix.loadCoarseQuantizer(p["coarse"]); ix.loadProductQuantizer(p["pq"]); ix.setW(w)
ix.indexVectors([str(i) for i in range(n)], p["base"])
ref = o.OracleIndex(o.KIND_IVFPQ, D, m, ks, C)
ref.set_coarse(p["coarse"]); ref.set_pq(p["pq"]); ref.set_w(w); ref.add_vectors(p["base"])
res = ix.search_batch(k, p["queries"])
rr = ref.search_batch(p["queries"], k)
print("ids equal", np.array_equal(res[0], rr[0]), "dist equal", np.array_equal(res[1], rr[1]), "counts", np.array_equal(res[2], rr[2]))
if not np.array_equal(res[0], rr[0]):
    bad = np.where((res[0] != rr[0]).any(1))[0]
    print("bad queries", bad[:10], len(bad))
    q = bad[0]; print(res[0][q], rr[0][q]); print(res[1][q], rr[1][q])
