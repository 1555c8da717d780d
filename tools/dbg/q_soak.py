"""q_soak.py -- scratch: K3q against K3h (option passa_q 1 / 0) on a mid-size index with long lists, many batches, bit for bit.
Run on the GPU box:  python tools/dbg/q_soak.py [batches] [queries per batch]"""
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
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
D, m, ks, C, w, k, n = 128, 16, 256, 128, 8, 100, 2_000_000
rng = np.random.default_rng(12)
mu = rng.standard_normal((C, D))
lab = rng.integers(0, C, n)
base = mu[lab] + 0.35 * rng.standard_normal((n, D))
ds = D // m
res = (mu[lab[:20000]] - base[:20000])
pq = np.stack([synth.kmeans(res[:, s * ds:(s + 1) * ds], ks, iters=3, seed=s) for s in range(m)])
ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
ix.loadCoarseQuantizer(mu)
ix.loadProductQuantizer(pq)
ix.setW(w)
ix.set_option("passa_mfma", int(os.environ.get("Q_SOAK_MFMA", "0")))  # 0: the comparison kernel is K3h; -1: K3ma where its gate takes the batch
t0 = time.time()
for i0 in range(0, n, 500000):
    ix.indexVectors([str(i) for i in range(i0, min(n, i0 + 500000))], base[i0:i0 + 500000])
print(f"index of {n} vectors built in {time.time() - t0:.1f} s", flush=True)
bad = 0
for b in range(nb):
    kind = b % 4
    if kind == 0:
        Q = base[rng.integers(0, n, nq)] + 0.01 * rng.standard_normal((nq, D))
    elif kind == 1:
        Q = 0.5 * (base[rng.integers(0, n, nq)] + base[rng.integers(0, n, nq)])
    elif kind == 2:
        Q = mu[rng.integers(0, C, nq)] + 0.35 * rng.standard_normal((nq, D))
    else:
        Q = mu[rng.integers(0, 4, nq)] + 0.2 * rng.standard_normal((nq, D))  # four lists take the whole batch: hundreds of groups per list
    ix.set_option("passa_q", 1)
    a = ix.search_batch(k, Q)
    da = ix.get_dispatch()["pass_a"]
    ix.set_option("passa_q", 0)
    c = ix.search_batch(k, Q)
    dc = ix.get_dispatch()["pass_a"]
    ok = all(np.array_equal(x, y) for x, y in zip(a, c))
    if not ok:
        bad += 1
        diff = np.nonzero(np.any(a[0] != c[0], axis=1) | np.any(a[1] != c[1], axis=1))[0]
        print(f"batch {b} kind {kind}: {len(diff)} queries differ, first {diff[:5]}", flush=True)
    if b == 0:
        print(f"dispatch: {da} against {dc}", flush=True)
print(f"{nb} batches of {nq} queries, {bad} with differences")
sys.exit(1 if bad else 0)
