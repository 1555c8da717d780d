#!/usr/bin/env python3
"""Side measurements bench.py publishes next to the headline (each in its own JSON object):

  host_path   the boundary's own entry point on the headline index: mmidx_search with HOST buffers (what the JNI shim calls,
              ASS.computeNearestNeighbors ASS:281-291) at 1 / 64 / 16384 queries per call, and native caller threads that
              issue one query per call (tools/callers_harness.c) with and without the library combining them
  cfg5        BASELINE config 5's front end: PCA 8192 -> 128 on the f64 matrix cores against the MEASURED f64-MFMA peak,
              VLAD aggregation, the fused descriptors -> 128-d call, and the pipeline end to end on 1 M synthetic images
  probes      the measured ceilings: f64 MFMA peak, LDS gather rate at pass A's access pattern (mmidx_probe.hip)
"""
import ctypes as C
import importlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "examples")):
    if p not in sys.path:
        sys.path.insert(0, p)


def probes(L, nat, device=0):
    out = {}
    v = (C.c_double * 4)()
    nat.check(L.mmidx_probe_f64_mfma(device, v))
    out["f64_mfma_peak_tflops_measured"] = round(v[0], 2)
    for m, ch in ((16, 3), (16, 1), (8, 3), (32, 1)):
        nat.check(L.mmidx_probe_lds_gather(device, m, ch, v))
        out[f"lds_gather_m{m}_chains{ch}"] = {"wave_gathers_per_s": round(v[0], 1), "algorithmic_GBps": round(v[1], 1), "blocks_per_cu": int(v[2])}
    return out


def host_path(L, nat, h, Qh, k, threads=(64, 128), calls_per_thread=200):
    """h: the cfg4 index; Qh: host queries [nq][D] (float64, C order)"""
    nq_all, D = Qh.shape
    out = {"note": "mmidx_search with host buffers (PCIe both ways, synchronous): the call the JNI shim makes"}
    oi = np.empty((nq_all, k), np.int32)
    od = np.empty((nq_all, k), np.float64)
    oc = np.empty(nq_all, np.int32)
    for nq in (1, 64, min(16384, nq_all)):
        reps = 200 if nq == 1 else (50 if nq <= 64 else 6)
        for i in range(3):
            nat.check(L.mmidx_search(h, k, nq, Qh[i:].ctypes.data, oi.ctypes.data, od.ctypes.data, oc.ctypes.data))
        ts = []
        for i in range(reps):
            q0 = (i * nq) % max(1, nq_all - nq + 1)
            t0 = time.perf_counter()
            nat.check(L.mmidx_search(h, k, nq, Qh[q0:].ctypes.data, oi.ctypes.data, od.ctypes.data, oc.ctypes.data))
            ts.append(time.perf_counter() - t0)
        ts.sort()
        med = ts[len(ts) // 2]
        out[f"nq{nq}"] = {"ms_per_call_median": round(med * 1e3, 4), "queries_per_s": round(nq / med, 1)}
    # two and three reader threads, every one with requests of 16384 queries (ctypes drops the interpreter lock inside the call):
    # one caller's copies run under another caller's kernels (search_host_big)
    import threading

    nq = min(16384, nq_all)
    ref_i, ref_d = oi[:nq].copy(), od[:nq].copy()
    nat.check(L.mmidx_search(h, k, nq, Qh.ctypes.data, ref_i.ctypes.data, ref_d.ctypes.data, oc.ctypes.data))
    for T in (2, 3):
        bufs = [(np.empty((nq, k), np.int32), np.empty((nq, k), np.float64), np.empty(nq, np.int32)) for _ in range(T)]
        calls, errs = 8, []

        def work(t):
            bi, bd, bc = bufs[t]
            for _ in range(calls):
                rc = L.mmidx_search(h, k, nq, Qh.ctypes.data, bi.ctypes.data, bd.ctypes.data, bc.ctypes.data)
                if rc:
                    errs.append(rc)

        for rep in range(2):  # (the first round warms the slots' buffers)
            th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            t0 = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            sec = time.perf_counter() - t0
        same = all(np.array_equal(b[0], ref_i) and np.array_equal(b[1], ref_d) for b in bufs)
        out[f"nq{nq}_callers{T}"] = {"queries_per_s": round(T * calls * nq / sec, 1), "ms_per_call_per_caller": round(sec / calls * 1e3, 4),
                                     "errors": len(errs), "answers_equal_single_caller": bool(same)}
    # native caller threads, one query per call (no interpreter lock in the way)
    try:
        tmp = tempfile.mkdtemp()
        so = os.path.join(tmp, "callers_harness.so")
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-pthread", "-o", so, os.path.join(ROOT, "tools", "callers_harness.c")])
        H = C.CDLL(so)
        H.run_callers.restype = C.c_double
        H.run_callers.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int)]
        fn = C.cast(L.mmidx_search, C.c_void_p)
        rows = []
        for combine in (1, 0):
            nat.check(L.mmidx_set_option(h, b"combine", combine))
            for T in threads:
                calls = calls_per_thread if combine else max(20, calls_per_thread // 8)
                errs = C.c_int(0)
                H.run_callers(fn, h, k, D, Qh.ctypes.data, nq_all, T, 10, C.byref(errs))
                sec = H.run_callers(fn, h, k, D, Qh.ctypes.data, nq_all, T, calls, C.byref(errs))
                rows.append({"combine": combine, "threads": T, "calls": T * calls, "queries_per_s": round(T * calls / sec, 1),
                             "ms_per_call": round(sec / calls * 1e3, 4), "errors": errs.value})
        nat.check(L.mmidx_set_option(h, b"combine", 1))
        out["single_query_caller_threads"] = rows
    except Exception as e:  # noqa: BLE001 (no compiler on the box: the per-call figures above stand alone)
        out["single_query_caller_threads"] = {"error": repr(e)}
    return out


def cfg5(L, nat, mi, images_e2e=1_000_000, device=0):
    """PCA / VLAD / fused front end (inputs resident in HBM) and the whole pipeline on images_e2e synthetic images"""
    import torch

    dev = torch.device("cuda", device)
    f64 = torch.float64
    out = {}
    v = (C.c_double * 4)()
    nat.check(L.mmidx_probe_f64_mfma(device, v))
    peak = v[0]
    # ---- PCA: n x 8192 -> 128, whitening (PCA.sampleToEigenSpace, PCA.java:188-208)
    n, ss, nc = 131072, 8192, 128
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    X = torch.randn(n, ss, generator=g, device=dev, dtype=f64) / 90.0
    Vt = torch.linalg.qr(torch.randn(ss, nc, generator=g, device=dev, dtype=f64))[0].T.contiguous()
    mu = 0.01 * torch.randn(ss, generator=g, device=dev, dtype=f64)
    eig = torch.linspace(4.0, 0.5, nc, dtype=f64)
    hp = C.c_void_p()
    mu_h, eig_h, Vt_h = mu.cpu().numpy(), eig.numpy(), Vt.cpu().numpy()
    nat.check(L.mmidx_pca_create(nc, ss, 1, mu_h.ctypes.data, eig_h.ctypes.data, Vt_h.ctypes.data, device, C.byref(hp)))
    Y = torch.empty(n, nc, device=dev, dtype=f64)
    for _ in range(2):
        nat.check(L.mmidx_pca_project_device(hp, n, X.data_ptr(), Y.data_ptr(), None))
    torch.cuda.synchronize()
    R = 5
    t0 = time.perf_counter()
    for _ in range(R):
        nat.check(L.mmidx_pca_project_device(hp, n, X.data_ptr(), Y.data_ptr(), None))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    flops = 2.0 * n * ss * nc
    Vw = Vt * (eig.to(dev) ** -0.5)[:, None]
    ref = (X - mu) @ Vw.T
    ref = ref / ref.norm(dim=1, keepdim=True)
    err = float((Y - ref).abs().max())
    tf = flops / dt / 1e12
    out["pca_8192_to_128"] = {"samples": n, "ms": round(dt * 1e3, 3), "tflops_f64": round(tf, 2), "images_per_s": round(n / dt, 1),
                              "hbm_GBps_of_input": round(n * ss * 8 / dt / 1e9, 1), "max_abs_err_vs_torch": err,
                              "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": round(peak, 2), "unit": "TFLOP/s",
                                           "frac": round(tf / peak, 4) if peak > 0 else None,
                                           "peak_source": "mmidx_probe_f64_mfma (measured in this run; MI355X_MICROARCH.md quotes 78.6 TF fp64 matrix)",
                                           "hbm_frac_of_8TBps": round(n * ss * 8 / dt / 8e12, 4),
                                           "note": "128 output columns per 8192-d row: 32 flops per input byte -- at 8 TB/s the input alone caps this at 256 TF, so the "
                                                   "kernel is priced against the matrix cores"}}
    nat.check(L.mmidx_pca_destroy(hp))
    del X, Y, ref
    # ---- VLAD and the fused call
    nimg, dl, ncent = 20000, 64, 128
    rng = np.random.default_rng(2)
    nd = rng.integers(200, 801, size=nimg)
    off = np.zeros(nimg + 1, np.int64)
    off[1:] = np.cumsum(nd)
    tot = int(off[-1])
    Dd = torch.randn(tot, dl, generator=g, device=dev, dtype=f64)
    Dd = Dd / Dd.norm(dim=1, keepdim=True)
    cb = torch.randn(ncent, dl, generator=g, device=dev, dtype=f64) / 8.0
    hv = C.c_void_p()
    nca = np.array([ncent], np.int32)
    cb_h = cb.cpu().numpy()
    nat.check(L.mmidx_vlad_create(1, nca.ctypes.data, dl, cb_h.ctypes.data, 1, device, C.byref(hv)))
    d_off = torch.tensor(off, device=dev)
    V = torch.empty(nimg, ncent * dl, device=dev, dtype=f64)
    for _ in range(2):
        nat.check(L.mmidx_vlad_aggregate_device(hv, nimg, d_off.data_ptr(), Dd.data_ptr(), int(nd.max()), V.data_ptr(), None))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        nat.check(L.mmidx_vlad_aggregate_device(hv, nimg, d_off.data_ptr(), Dd.data_ptr(), int(nd.max()), V.data_ptr(), None))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    vb = float(tot) * dl * 8
    try:
        vtr = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "hbm_traffic.json"))).get("vlad", {})
    except Exception:  # noqa: BLE001
        vtr = {}
    out["vlad_surf64_128_centroids"] = {"images": nimg, "descriptors": tot, "ms": round(dt * 1e3, 3), "images_per_s": round(nimg / dt, 1),
                                        "f64_triples_per_s": round(tot * ncent * dl / dt, 1),
                                        "roofline": {"bound": "hbm", "kernel": "k_vlad_fused (K8'': a block per image -- nearest centroid of its descriptors by the certified bf16-split MFMA "
                                                                             "argmin, flagged ones redone in fp64 in the block, ordered accumulation, power + L2)",
                                                     "achieved": round(vb / dt / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(vb / dt / 8e12, 4),
                                                     "algorithmic_bytes_per_launch": vb,
                                                     "traffic": vtr.get("k_vlad_fused_fetch_bytes_per_launch") if vtr.get("descriptors") == tot else None,
                                                     "traffic_written": vtr.get("k_vlad_fused_write_bytes_per_launch") if vtr.get("descriptors") == tot else None,
                                                     "traffic_source": "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE x 2 / WRITE_SIZE, separate pass: profiles/r05b_vlad_pmc_kernels.txt)",
                                                     "note": "algorithmic bytes = the descriptors once (n x 64 x 8).  One kernel and no host synchronisation; the block reads its "
                                                             "image's rows twice (assignment, then the ordered accumulation into the waves' register sums) and the second read reaches "
                                                             "the fabric again; the vector is written once.  Latency-bound phases inside a block at three blocks per CU "
                                                             "(DESIGN.md 5.5).  Round 4's two kernels: 4.8 M images/s; round 3's fp64 brute force: 0.94 M"}}
    hp2 = C.c_void_p()
    mean_v = V.mean(0).cpu().numpy()
    nat.check(L.mmidx_pca_create(nc, ss, 1, mean_v.ctypes.data, eig_h.ctypes.data, Vt_h.ctypes.data, device, C.byref(hp2)))
    Yv = torch.empty(nimg, nc, device=dev, dtype=f64)
    for _ in range(2):
        nat.check(L.mmidx_vectorize_device(hv, hp2, nimg, d_off.data_ptr(), Dd.data_ptr(), int(nd.max()), Yv.data_ptr(), None))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        nat.check(L.mmidx_vectorize_device(hv, hp2, nimg, d_off.data_ptr(), Dd.data_ptr(), int(nd.max()), Yv.data_ptr(), None))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    out["fused_descriptors_to_128d"] = {"images": nimg, "ms": round(dt * 1e3, 3), "images_per_s": round(nimg / dt, 1)}
    nat.check(L.mmidx_pca_destroy(hp2))
    nat.check(L.mmidx_vlad_destroy(hv))
    del V, Dd, Yv
    torch.cuda.empty_cache()
    # ---- end to end at the stated size (config5_pipeline.run_device: synthesis + vectorize on the device, IVFPQ cfg3-style)
    if images_e2e > 0:
        import config5_pipeline as c5

        t0 = time.time()
        summ, _ = c5.run_device(n_images=images_e2e, n_queries=1024, k=10, cells=1024, w=8, chunk=16384)
        summ["wall_s"] = round(time.time() - t0, 1)
        out["end_to_end"] = summ
    return out
