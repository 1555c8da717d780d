#!/usr/bin/env python3
"""cfg2 (flat PQ 1M x 128, m = 8, k = 100, 4096 queries) alone: qps, stage times and K3g statistics; run under rocprofv3 for kernels"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synth  # noqa: E402

try:
    import torch

    torch.cuda.init()
except Exception:
    pass
mi = importlib.import_module("multimedia-indexing_amd")
rng = np.random.default_rng(0)
N, D, k, m, ks = 1_000_000, 128, 100, 8, 256
base = rng.standard_normal((N, D))
pq = np.stack([synth.kmeans(base[:30000, s * 16:(s + 1) * 16], ks, iters=5, seed=s) for s in range(m)])
ix = mi.PQ(D, N, False, "", m, ks, 0, 512)
ix.loadProductQuantizer(pq)
ix.indexVectors(list(range(N)), base)
for o_ in sys.argv[1:]:
    a, b = o_.split("=")
    ix.set_option(a, int(b))
Q = rng.standard_normal((4096, D))
for _ in range(2):
    ix.search_batch(k, Q)
t0 = time.time()
for _ in range(5):
    ix.search_batch(k, Q)
dt = (time.time() - t0) / 5
print(f"cfg2: {4096 / dt:.0f} q/s, {dt * 1e3:.2f} ms per 4096 queries (host buffers)")
ix.set_profiling(1)
ix.search_batch(k, Q)
print(ix.get_stats())
ix.close()
