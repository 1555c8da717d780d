#!/usr/bin/env python3
"""Micro-benchmark of the config-5 front end on one MI355X (inputs resident in HBM):
PCA projection (f64 MFMA, 8192 -> 128) and VLAD aggregation (SURF-64, 128 centroids)."""
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mi = importlib.import_module("multimedia-indexing_amd")
nat = importlib.import_module("multimedia-indexing_amd._native")
L = mi.lib()
dev = torch.device("cuda", 0)
f64 = torch.float64
out = {}

# ---- PCA: n x 8192 -> 128, whitening
n, ss, nc = 131072, 8192, 128
g = torch.Generator(device=dev)
g.manual_seed(1)
X = torch.randn(n, ss, generator=g, device=dev, dtype=f64) / 90.0
Vt = torch.linalg.qr(torch.randn(ss, nc, generator=g, device=dev, dtype=f64))[0].T.contiguous()
mu = 0.01 * torch.randn(ss, generator=g, device=dev, dtype=f64)
eig = torch.linspace(4.0, 0.5, nc, dtype=f64)
h = C.c_void_p()
mu_h, eig_h, Vt_h = mu.cpu().numpy(), eig.numpy(), Vt.cpu().numpy()  # keep the host arrays alive across the call
nat.check(L.mmidx_pca_create(nc, ss, 1, mu_h.ctypes.data, eig_h.ctypes.data, Vt_h.ctypes.data, 0, C.byref(h)))
Y = torch.empty(n, nc, device=dev, dtype=f64)
for _ in range(2):
    nat.check(L.mmidx_pca_project_device(h, n, X.data_ptr(), Y.data_ptr(), None))
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 5
for _ in range(R):
    nat.check(L.mmidx_pca_project_device(h, n, X.data_ptr(), Y.data_ptr(), None))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / R
flops = 2.0 * n * ss * nc
# torch reference (rocBLAS f64) for a cross-check of the numbers, not of the product path
Vw = Vt * (eig.to(dev) ** -0.5)[:, None]
ref = (X - mu) @ Vw.T
ref = ref / ref.norm(dim=1, keepdim=True)
err = float((Y - ref).abs().max())
t1 = time.perf_counter()
for _ in range(R):
    ref = (X - mu) @ Vw.T
torch.cuda.synchronize()
dt_ref = (time.perf_counter() - t1) / R
out["pca"] = {"n": n, "ss": ss, "nc": nc, "ms": dt * 1e3, "tflops_f64": flops / dt / 1e12, "images_per_s": n / dt,
              "hbm_GBps": n * ss * 8 / dt / 1e9, "max_abs_err_vs_torch": err, "torch_rocblas_ms": dt_ref * 1e3}
nat.check(L.mmidx_pca_destroy(h))
del X, Y, ref

# ---- VLAD: nimg images, U[200,800] SURF-64 descriptors, 128 centroids, power + L2
nimg, dl, ncent = 20000, 64, 128
rng = np.random.default_rng(2)
nd = rng.integers(200, 801, size=nimg)
off = np.zeros(nimg + 1, np.int64)
off[1:] = np.cumsum(nd)
tot = int(off[-1])
D = torch.randn(tot, dl, generator=g, device=dev, dtype=f64)
D = D / D.norm(dim=1, keepdim=True)
cb = torch.randn(ncent, dl, generator=g, device=dev, dtype=f64) / 8.0
hv = C.c_void_p()
nca = np.array([ncent], np.int32)
cb_h = cb.cpu().numpy()
nat.check(L.mmidx_vlad_create(1, nca.ctypes.data, dl, cb_h.ctypes.data, 1, 0, C.byref(hv)))
d_off = torch.tensor(off, device=dev)
V = torch.empty(nimg, ncent * dl, device=dev, dtype=f64)
for _ in range(2):
    nat.check(L.mmidx_vlad_aggregate_device(hv, nimg, d_off.data_ptr(), D.data_ptr(), int(nd.max()), V.data_ptr(), None))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(R):
    nat.check(L.mmidx_vlad_aggregate_device(hv, nimg, d_off.data_ptr(), D.data_ptr(), int(nd.max()), V.data_ptr(), None))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / R
out["vlad"] = {"images": nimg, "descriptors": tot, "ms": dt * 1e3, "images_per_s": nimg / dt,
               "f64_triples_per_s": tot * ncent * dl / dt}
nat.check(L.mmidx_vlad_destroy(hv))
print(json.dumps(out))
