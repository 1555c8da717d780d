#!/usr/bin/env python3
"""Latency of one computeNearestNeighbors call as a function of the number of queries handed over at once
(diagnostic, not a bench line).  The reference's API is one query per call (ASS.computeNearestNeighbors), so nq = 1 with
host buffers is what a JNI caller that does not batch would see.

Builds the cfg4 index (synthetic, codebook quality irrelevant here) and times, for every batch size, `mmidx_search_device`
(queries / results resident in HBM, synchronised after every call) and `mmidx_search` (host pointers: H2D of the queries,
D2H of the results included).  Prints one JSON line.

  python tools/bench_latency.py [--n 100000000] [--reps 40]
"""
import argparse
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
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100_000_000)
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--chunk", type=int, default=2_000_000)
ap.add_argument("--sizes", type=str, default="1,2,4,8,16,64,256,1024,4096,16384")
ap.add_argument("--opt", action="append", default=[], metavar="NAME=INT")
ap.add_argument("--threads", type=str, default="1,4,16,64,256", help="caller threads of the one-query-per-call test")
args = ap.parse_args()
nat = importlib.import_module("multimedia-indexing_amd._native")
torch.cuda.init()
L = nat.lib()
dev = torch.device("cuda", 0)
N, D, Cc, w, m, ks, k = args.n, 128, 8192, 32, 16, 256, 100
sizes = [int(s) for s in args.sizes.split(",")]
B = max(sizes)
f64 = torch.float64
st = torch.cuda.current_stream().cuda_stream
g0 = torch.Generator(device=dev)
g0.manual_seed(1234)
mu = torch.randn(Cc, D, generator=g0, device=dev, dtype=f64)
coarse_h = mu.cpu().numpy()
pq_h = (0.15 * torch.randn(m, ks, D // m, generator=g0, device=dev, dtype=f64)).cpu().numpy()
h = C.c_void_p()
nat.check(L.mmidx_create(nat.KIND_IVFPQ, D, m, ks, Cc, 0, None, None, 0, C.byref(h)))
nat.check(L.mmidx_set_coarse(h, coarse_h.ctypes.data))
nat.check(L.mmidx_set_pq(h, pq_h.ctypes.data))
nat.check(L.mmidx_set_w(h, w))
for o in args.opt:
    name, val = o.split("=")
    nat.check(L.mmidx_set_option(h, name.encode(), int(val)))
gq = torch.Generator(device=dev)
gq.manual_seed(4321)
qsrc = torch.randint(0, N, (B,), generator=gq, device=dev)
Q = torch.zeros(B, D, device=dev, dtype=f64)
for c0 in range(0, N, args.chunk):
    n = min(args.chunk, N - c0)
    gc = torch.Generator(device=dev)
    gc.manual_seed(10_000 + c0 // args.chunk)
    X = mu[torch.randint(0, Cc, (n,), generator=gc, device=dev)]
    X += 0.15 * torch.randn(n, D, generator=gc, device=dev, dtype=f64)
    sel = (qsrc >= c0) & (qsrc < c0 + n)
    if sel.any():
        Q[sel] = X[qsrc[sel] - c0]
    torch.cuda.synchronize()
    nat.check(L.mmidx_add_vectors_device(h, n, X.data_ptr(), None, c0, st))
    torch.cuda.synchronize()
    del X
nat.check(L.mmidx_sync_index(h))
Q += 0.01 * torch.randn(B, D, generator=gq, device=dev, dtype=f64)
Qh = Q.cpu().numpy()
d_iid = torch.empty(B, k, dtype=torch.int32, device=dev)
d_dist = torch.empty(B, k, dtype=f64, device=dev)
d_cnt = torch.empty(B, dtype=torch.int32, device=dev)
h_iid = np.empty((B, k), np.int32)
h_dist = np.empty((B, k), np.float64)
h_cnt = np.empty(B, np.int32)
rows = []
for nq in sizes:
    r = {"nq": nq}
    reps = args.reps if nq <= 4096 else max(5, args.reps // 4)
    for kind in ("device", "host"):
        ts = []
        for i in range(reps + 3):
            q0 = (i * nq) % max(1, B - nq + 1)  # a different slice every call
            if kind == "device":
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                nat.check(L.mmidx_search_device(h, k, nq, Q[q0:].data_ptr(), d_iid.data_ptr(), d_dist.data_ptr(), d_cnt.data_ptr(), st))
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            else:
                t0 = time.perf_counter()
                nat.check(L.mmidx_search(h, k, nq, Qh[q0:].ctypes.data, h_iid.ctypes.data, h_dist.ctypes.data, h_cnt.ctypes.data))
                t1 = time.perf_counter()
            if i >= 3:
                ts.append((t1 - t0) * 1e3)
        ts.sort()
        r[kind + "_ms_median"] = round(ts[len(ts) // 2], 4)
        r[kind + "_ms_min"] = round(ts[0], 4)
        r[kind + "_qps"] = round(nq / (ts[len(ts) // 2] * 1e-3), 1)
    # the two paths return the same answers
    nat.check(L.mmidx_search_device(h, k, nq, Q.data_ptr(), d_iid.data_ptr(), d_dist.data_ptr(), d_cnt.data_ptr(), st))
    torch.cuda.synchronize()
    nat.check(L.mmidx_search(h, k, nq, Qh.ctypes.data, h_iid.ctypes.data, h_dist.ctypes.data, h_cnt.ctypes.data))
    r["same"] = bool((d_iid[:nq].cpu().numpy() == h_iid[:nq]).all() and (d_dist[:nq].cpu().numpy() == h_dist[:nq]).all())
    rows.append(r)
    print(r, file=sys.stderr, flush=True)
# ---- many caller threads, one query per call (the reference's usage: reader threads on one index) ----------------
import subprocess
import tempfile

tmp = tempfile.mkdtemp()
so = os.path.join(tmp, "callers_harness.so")
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-pthread", "-o", so, os.path.join(ROOT, "tools", "callers_harness.c")])
H = C.CDLL(so)
H.run_callers.restype = C.c_double
H.run_callers.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int)]
fn = C.cast(L.mmidx_search, C.c_void_p)
trows = []
for combine in (1, 0):
    nat.check(L.mmidx_set_option(h, b"combine", combine))
    for T in [int(t) for t in args.threads.split(",")]:
        calls = max(20, min(1000, (40000 if combine else 8000) // T))
        errs = C.c_int(0)
        H.run_callers(fn, h, k, D, Qh.ctypes.data, B, T, 20, C.byref(errs))  # warm-up
        sec = H.run_callers(fn, h, k, D, Qh.ctypes.data, B, T, calls, C.byref(errs))
        r = {"combine": combine, "threads": T, "calls": T * calls, "qps": round(T * calls / sec, 1),
             "ms_per_call": round(sec / calls * 1e3, 4), "errors": errs.value}
        trows.append(r)
        print(r, file=sys.stderr, flush=True)
nat.check(L.mmidx_set_option(h, b"combine", 1))
# ---- the same callers on BASELINE config 1 (Linear, 10k x 128, k = 10) --------------------------------------
lrows = []
rng = np.random.default_rng(5)
X1 = rng.standard_normal((10000, D))
Q1 = np.ascontiguousarray(X1[rng.choice(10000, 1000, replace=False)] + 0.05 * rng.standard_normal((1000, D)))
lh = C.c_void_p()
nat.check(L.mmidx_linear_create(D, 10000, 0, C.byref(lh)))
nat.check(L.mmidx_linear_add(lh, 10000, X1.ctypes.data))
lfn = C.cast(L.mmidx_linear_search, C.c_void_p)
for T in [int(t) for t in args.threads.split(",")]:
    calls = max(20, min(1000, 40000 // T))
    errs = C.c_int(0)
    H.run_callers(lfn, lh, 10, D, Q1.ctypes.data, 1000, T, 20, C.byref(errs))
    sec = H.run_callers(lfn, lh, 10, D, Q1.ctypes.data, 1000, T, calls, C.byref(errs))
    r = {"threads": T, "calls": T * calls, "qps": round(T * calls / sec, 1), "ms_per_call": round(sec / calls * 1e3, 4), "errors": errs.value}
    lrows.append(r)
    print("linear", r, file=sys.stderr, flush=True)
nat.check(L.mmidx_linear_destroy(lh))
print(json.dumps({"workload": f"IVFPQ {N}x{D}, C={Cc}, w={w}, m={m}x{ks}, k={k}", "latency": rows, "single_query_callers": trows, "linear_10k_single_query_callers": lrows}))
L.mmidx_destroy(h)
