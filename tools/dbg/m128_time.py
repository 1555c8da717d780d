"""the m = 128 shape of tests/bench_configs.py (100 k x 1024-d, 128 x 256, 128 cells, w = 8, k = 30): stage times and dispatch"""
import importlib, os, sys, time
sys.path[:0] = [".", "tests"]
import numpy as np, torch
import synth
mi = importlib.import_module("multimedia-indexing_amd"); nat = importlib.import_module("multimedia-indexing_amd._native")
L = mi.lib(); dev = torch.device("cuda:0"); rng = np.random.default_rng(0)
N8, D8, m8, C8, w8, k8, ks = int(os.environ.get("N8", 100_000)), 1024, 128, int(os.environ.get("C8", 128)), 8, 30, 256
base8, mu8 = synth.mixture(N8, D8, C8, sigma=0.3, seed=5)
ix = mi.IVFPQ(D8, N8, False, "", m8, ks, 0, C8, 512)
ix.loadCoarseQuantizer(mu8)
cell = ((base8[:20000] * base8[:20000]).sum(1)[:, None] - 2 * base8[:20000] @ mu8.T + (mu8 * mu8).sum(1)[None]).argmin(1)
resid = mu8[cell] - base8[:20000]
pq8 = np.stack([synth.kmeans(resid[:, s * 8:(s + 1) * 8], ks, iters=2, seed=s) for s in range(m8)])
ix.loadProductQuantizer(pq8); ix.setW(w8)
ix.indexVectors(list(range(N8)), base8)
for o in sys.argv[1:]:
    kk, vv = o.split("="); ix.set_option(kk, int(vv))
B = int(os.environ.get("B8", 1024))
qi = rng.choice(N8, B, replace=False)
Q8 = base8[qi] + 0.01 * rng.standard_normal((B, D8))
dQ = torch.tensor(Q8, dtype=torch.float64, device=dev)
iid = torch.empty(B, k8, dtype=torch.int32, device=dev); dd = torch.empty(B, k8, dtype=torch.float64, device=dev); cc = torch.empty(B, dtype=torch.int32, device=dev)
def go():
    nat.check(L.mmidx_search_device(ix._h, k8, B, dQ.data_ptr(), iid.data_ptr(), dd.data_ptr(), cc.data_ptr(), None))
for _ in range(2): go()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): go()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("ms per batch of %d: %.3f  (%.3f M q/s)  dispatch %s" % (B, dt * 1e3, B / dt / 1e6, ix.get_dispatch()))
ix.set_profiling(True); go(); st = ix.get_stats()
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items() if v})
if os.environ.get("PARITY", "1") == "1":
    from oracle import oracle as o
    ref = o.OracleIndex(o.KIND_IVFPQ, D8, m8, ks, C8); ref.set_coarse(mu8); ref.set_pq(pq8); ref.set_w(w8)
    off, ids_e, codes_e = ix.export(); ref.load_lists(off, ids_e, codes_e)
    rid, rd, rc = ref.search_batch(Q8[:48], k8, nthreads=16)
    print("parity", bool(np.array_equal(iid.cpu().numpy()[:48], rid)), float(np.max(np.abs(dd.cpu().numpy()[:48] - rd))))
