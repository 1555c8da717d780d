#!/usr/bin/env python3
"""cfg2 (flat PQ 1M x 128, m = 8, k = 100, 4096 queries) with different chunk sizes / options: stage times and q/s"""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
torch.cuda.init()
import synth
mi = importlib.import_module("multimedia-indexing_amd"); nat = importlib.import_module("multimedia-indexing_amd._native"); L = mi.lib()
rng = np.random.default_rng(0)
N, D, k, m, ks = 1_000_000, 128, 100, 8, 256
base = rng.standard_normal((N, D))
pq = np.stack([synth.kmeans(base[:30000, s * 16:(s + 1) * 16], ks, iters=5, seed=s) for s in range(m)])
ix = mi.PQ(D, N, False, "", m, ks, 0, 512); ix.loadProductQuantizer(pq); ix.indexVectors(list(range(N)), base)
NQ = int(os.environ.get("NQ", "4096"))
Q = torch.tensor(rng.standard_normal((NQ, D)), dtype=torch.float64, device="cuda")
iid = torch.empty(NQ, k, dtype=torch.int32, device="cuda"); dd = torch.empty(NQ, k, dtype=torch.float64, device="cuda"); cc = torch.empty(NQ, dtype=torch.int32, device="cuda")
ref = None
for opts in sys.argv[1:] or ["flat_chunk=0"]:
    for o in opts.split(","):
        n_, v_ = o.split("="); ix.set_option(n_, int(v_))
    f = lambda: nat.check(L.mmidx_search_device(ix._h, k, NQ, Q.data_ptr(), iid.data_ptr(), dd.data_ptr(), cc.data_ptr(), None))
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    ix.set_profiling(True)
    for _ in range(4): f()
    torch.cuda.synchronize(); st = ix.get_stats(); ix.set_profiling(False)
    r = (iid.cpu().numpy().copy(), dd.cpu().numpy().copy())
    same = True if ref is None else bool(np.array_equal(r[0], ref[0]) and np.array_equal(r[1], ref[1]))
    ref = ref or r
    print(opts, "qps %.0f ms %.3f  passA %.3f passB %.3f merge %.3f verified/q %.0f same=%s" % (NQ / dt, dt * 1e3, st["passa_ms"] / 4, (st["scan_ms"] - st["passa_ms"]) / 4, st["merge_ms"] / 4, st["verified_codes"] / 4 / NQ, same))
