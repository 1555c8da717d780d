import importlib, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import synth
mi = importlib.import_module("multimedia-indexing_amd")
D,C,m,ks,n,w,k = 128,64,16,256,20000,8,100
p = synth.make_ivfpq_problem(n=n, D=D, C=C, m=m, ks=ks, nq=64, seed=D + C)
ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
ix.loadCoarseQuantizer(p["coarse"]); ix.loadProductQuantizer(p["pq"]); ix.setW(w)
print("encode", flush=True)
cells, codes = ix.encode(p["base"][:1500])
print("index", flush=True)
ix.indexVectors([str(i) for i in range(n)], p["base"])
print("sizes", ix.listSizes()[:8], flush=True)
for kk in (10, 100):
    print("search k", kk, flush=True)
    r = ix.search_batch(kk, p["queries"])
    print("ok", r[2][:4], flush=True)
print("single", flush=True)
a = ix.computeNearestNeighbors(k, p["queries"][0])
print("done", len(a.getIds()))
