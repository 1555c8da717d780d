#!/usr/bin/env python3
"""The reference's own flagship shape at its stated size (bench.py publishes the dict as `yfcc`):
YFCC100MExample.java:85-90 -- vectorLength 1024, 64 x 256 sub-quantizers, 8192 coarse centroids, RandomPermutation;
:58-59 / Example.java:96-97 -- probes w in {2, 64}; :155 -- k = 30; :93-99 -- 95,213,780 vectors in one index.  It is the one
configuration for which the reference states a latency ("less than 1 sec" per query, one thread).

Synthetic data of that shape (no dataset in the image): the Gaussian mixture of SURVEY 8d with 8192 means in 1024 dimensions
and mixture noise `sigma` -- 1.0 by default, cluster radius 32 against 45 between the means, so the cells overlap as real
VLAD + PCA vectors' cells do and the far probes are really scanned (with the 0.15 of the headline workload the exact coarse
bound would drop all of them and w = 64 would cost what w = 1 costs).  Codebooks by the library's own k-means, index built on
the device, queries = self-perturbed base vectors.  Reported per w: queries/s at `batch` queries per call (device buffers), the
latency of a single one-query call with host buffers (the reference's call shape, ASS:281-291), stage times, and ids +
distance bits against the CPU oracle loaded with the device's own codes (mmidx_export).
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
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def log(*a):
    print("[yfcc]", *a, file=sys.stderr, flush=True)


def mem_available_gb():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable"):
                return int(ln.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def run(n=95_213_780, D=1024, m=64, ks=256, Cc=8192, k=30, ws=(2, 64), batch=4096, sigma=1.0, steps=6, parity_queries=512,
        chunk=250_000, device=0, cpu_threads=None, opts=()):
    mi = importlib.import_module("multimedia-indexing_amd")
    nat = importlib.import_module("multimedia-indexing_amd._native")
    L, chk = mi.lib(), nat.check
    dev = torch.device("cuda", device)
    torch.cuda.set_device(dev)
    f64 = torch.float64
    dsub = D // m
    out = {"workload": f"IVFPQ {n}x{D}-d, {Cc} coarse cells, m={m}x{ks}, RandomPermutation, k={k}, batch {batch} (YFCC100MExample.java:85-99, :155)",
           "mixture_sigma": sigma}

    def kmeans(X, kk, iters, seed=1, init=None, plus_plus=False):
        nn, d = X.shape
        cent = np.full((kk, d), 1000.0)
        kout, it = C.c_int32(0), C.c_int32(0)
        ini = None if init is None else np.ascontiguousarray(init, np.float64)
        chk(L.mmidx_kmeans_device(device, nn, d, kk, iters, seed, 1 if plus_plus else 0, X.data_ptr(), ini.ctypes.data if ini is not None else None,
                                  cent.ctypes.data, None, None, C.addressof(it), C.addressof(kout), None))
        return cent

    # ---- codebooks: coarse = Lloyd from the mixture means, PQ = k-means++ per sub-space on residuals (centroid - vector)
    t0 = time.time()
    g0 = torch.Generator(device=dev)
    g0.manual_seed(1234)
    mu = torch.randn(Cc, D, generator=g0, device=dev, dtype=f64)
    ns = min(n, 1 << 18)
    gs = torch.randint(0, Cc, (ns,), generator=g0, device=dev)
    Xs = mu[gs] + sigma * torch.randn(ns, D, generator=g0, device=dev, dtype=f64)
    torch.cuda.synchronize()
    coarse_h = kmeans(Xs, Cc, 2, init=mu.cpu().numpy())
    coarse = torch.from_numpy(coarse_h).to(dev)
    hq = C.c_void_p()
    chk(L.mmidx_create(nat.KIND_IVFPQ, D, 1, 2, Cc, 0, None, None, device, C.byref(hq)))
    chk(L.mmidx_set_coarse(hq, coarse_h.ctypes.data))
    nr = min(ns, 1 << 17)
    cell_s = torch.empty(nr, dtype=torch.int32, device=dev)
    chk(L.mmidx_assign_device(hq, nr, Xs.data_ptr(), cell_s.data_ptr(), None))
    torch.cuda.synchronize()
    chk(L.mmidx_destroy(hq))
    resid = coarse[cell_s.long()] - Xs[:nr]
    del Xs
    # the index permutes the residual before it is split into sub-vectors (IVFPQ.java:318-323): the codebooks are learned in
    # the permuted space, as ProductQuantizationLearning.java:176-178 does with the same seed
    perm = np.asarray(mi.quantization.RandomPermutation(1, D).randomlyPermutatedIndices, np.int64)
    resid = resid[:, torch.from_numpy(perm).to(dev)]
    pq_h = np.empty((m, ks, dsub))
    for s in range(m):
        sub = resid[:, s * dsub:(s + 1) * dsub].contiguous()
        torch.cuda.synchronize()
        pq_h[s] = kmeans(sub, ks, 6, seed=s + 1, plus_plus=True)
    del resid
    out["codebooks_s"] = round(time.time() - t0, 1)
    log(f"codebooks in {out['codebooks_s']}s")

    # ---- index: encode + append on the device, chunk by chunk
    h = C.c_void_p()
    chk(L.mmidx_create(nat.KIND_IVFPQ, D, m, ks, Cc, nat.TR_PERMUTATION, None, None, device, C.byref(h)))
    chk(L.mmidx_set_coarse(h, coarse_h.ctypes.data))
    chk(L.mmidx_set_pq(h, pq_h.ctypes.data))
    for name_, val_ in opts:
        chk(L.mmidx_set_option(h, name_.encode(), int(val_)))
    nq_total = batch * 2
    gq = torch.Generator(device=dev)
    gq.manual_seed(4321)
    qsrc = torch.randint(0, n, (nq_total,), generator=gq, device=dev)
    Qsrc = torch.zeros(nq_total, D, device=dev, dtype=f64)
    t0 = time.time()
    t_enc = 0.0
    for c0 in range(0, n, chunk):
        nn = min(chunk, n - c0)
        gc = torch.Generator(device=dev)
        gc.manual_seed(20_000 + c0 // chunk)
        g = torch.randint(0, Cc, (nn,), generator=gc, device=dev)
        X = mu[g]
        X += sigma * torch.randn(nn, D, generator=gc, device=dev, dtype=f64)
        sel = (qsrc >= c0) & (qsrc < c0 + nn)
        if sel.any():
            Qsrc[sel] = X[qsrc[sel] - c0]
        torch.cuda.synchronize()
        te = time.time()
        chk(L.mmidx_add_vectors_device(h, nn, X.data_ptr(), None, c0, None))
        t_enc += time.time() - te
        del X
    chk(L.mmidx_sync_index(h))
    out["index_build_s"] = round(time.time() - t0, 1)
    out["encode_append_s"] = round(t_enc, 1)
    log(f"index: {n} vectors in {out['index_build_s']}s (encode + append {out['encode_append_s']}s)")
    Q = Qsrc + 0.01 * torch.randn(nq_total, D, generator=gq, device=dev, dtype=f64)
    Qb = [Q[i * batch:(i + 1) * batch].contiguous() for i in range(2)]
    iid = torch.empty(batch, k, dtype=torch.int32, device=dev)
    dd = torch.empty(batch, k, dtype=f64, device=dev)
    cc = torch.empty(batch, dtype=torch.int32, device=dev)
    Qh = Qb[0].cpu().numpy()

    # ---- the oracle with the device's own codes (only when the host can hold two copies of the index)
    from oracle import oracle as o  # (checker: the parity gate and the CPU figure below, nothing that is measured)

    ref = None
    need_gb = 2.5 * n * (m + 4) / 1e9 + 4
    if parity_queries > 0 and mem_available_gb() > need_gb:
        off = np.zeros(Cc + 1, np.int64)
        chk(L.mmidx_export(h, off.ctypes.data, None, None))
        e_iids = np.empty(int(off[-1]), np.int32)
        e_codes = np.empty((int(off[-1]), m), np.int8)
        chk(L.mmidx_export(h, off.ctypes.data, e_iids.ctypes.data, e_codes.ctypes.data))
        ref = o.OracleIndex(o.KIND_IVFPQ, D, m, ks, Cc, transform=2)
        ref.set_coarse(coarse_h)
        ref.set_pq(pq_h)
        ref.load_lists(off, e_iids, e_codes)
        del e_iids, e_codes
    elif parity_queries > 0:
        out["parity_skipped"] = f"host has {mem_available_gb():.0f} GB available, the oracle's copy of the index needs {need_gb:.0f}"
    if cpu_threads is None:
        cpu_threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:
            qv, pv = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if qv != "max":
                cpu_threads = max(1, min(cpu_threads, int(float(qv) / float(pv) + 0.999)))
        except (OSError, ValueError):
            pass

    # a second query set that defeats the pair pre-filter K3s: midpoints of two indexed vectors -- a query BETWEEN clusters sees its
    # probed cells at similar distances, far pairs stay alive after their Smin bound and pass B really scans far lists with the
    # m = 64 instance of K3g (the reference offers every probed code, IVFPQ.java:429-446; Example.java:96-97 probes 64 lists)
    Qmid = 0.5 * (Qsrc + Qsrc.roll(1, 0))
    Qmb = [Qmid[i * batch:(i + 1) * batch].contiguous() for i in range(2)]
    legs = [(w, Qb, f"w{w}", True) for w in ws] + [(max(ws), Qmb, f"w{max(ws)}_between_clusters", False)]
    st = nat.Stats()
    for w, Qb, tag, self_queries in legs:
        Qh = Qb[0].cpu().numpy()
        chk(L.mmidx_set_w(h, w))
        step = lambda Qx: chk(L.mmidx_search_device(h, k, batch, Qx.data_ptr(), iid.data_ptr(), dd.data_ptr(), cc.data_ptr(), None))
        for i in range(2):
            step(Qb[i % 2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(Qb[i % 2])
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        chk(L.mmidx_set_profiling(h, 1))
        nd = 2
        for i in range(nd):
            step(Qb[i % 2])
        torch.cuda.synchronize()
        chk(L.mmidx_get_stats(h, C.byref(st)))
        chk(L.mmidx_set_profiling(h, 0))
        step(Qb[0])
        torch.cuda.synchronize()
        g_iid, g_dd = iid.cpu().numpy().copy(), dd.cpu().numpy().copy()
        recall = float(np.mean(g_iid[:, 0] == qsrc[:batch].cpu().numpy())) if self_queries else None
        # one query per call with host buffers: the reference's own call shape
        one_i, one_d, one_c = np.empty((1, k), np.int32), np.empty((1, k)), np.empty(1, np.int32)
        for i in range(3):
            chk(L.mmidx_search(h, k, 1, Qh[i:i + 1].ctypes.data, one_i.ctypes.data, one_d.ctypes.data, one_c.ctypes.data))
        nl = 40
        t0 = time.perf_counter()
        for i in range(nl):
            chk(L.mmidx_search(h, k, 1, Qh[i:i + 1].ctypes.data, one_i.ctypes.data, one_d.ctypes.data, one_c.ctypes.data))
        lat_ms = (time.perf_counter() - t0) / nl * 1e3
        codes_q = st.scan_codes / nd / batch
        r = {"queries_per_s": round(batch * steps / el, 1), "ms_per_step": round(el / steps * 1e3, 3), "steps": steps,
             "single_query_call_ms_host_buffers": round(lat_ms, 3), "recall_at_1": recall,
             "stage_ms_per_step": {"coarse": round(st.coarse_ms / nd, 3), "pass_a": round(st.passa_ms / nd, 3),
                                   "pass_b": round((st.scan_ms - st.passa_ms) / nd, 3), "merge": round(st.merge_ms / nd, 3)},
             "probed_codes_per_query": round(codes_q, 1), "algorithmic_bytes_per_query": round(codes_q * m, 1),
             "algorithmic_GBps": round(codes_q * m * batch * steps / el / 1e9, 1),
             "far_pairs_scanned_per_query": round(int(st.passb_items_last) / batch, 3),
             "verified_codes_per_query": round(st.verified_codes / nd / batch, 2)}
        # pass A's kernel (k_scan_hist<64, 256, 1024>: the exact fp64 scan of every query's nearest list, one 1024-thread block per CU
        # over a 128 KiB table): algorithmic bytes = m x the codes of the nearest lists / its launch time (HIP events of the profiled steps)
        pa_ms = st.passa_ms / max(1, st.passa_launches)
        pa_bytes = float(m) * st.passa_codes / max(1, st.passa_launches)
        try:  # physical HBM bytes per launch: separate rocprofv3 --pmc FETCH_SIZE passes of this workload (x 2 on gfx950), profiles/
            tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get("yfcc", {})
        except Exception:  # noqa: BLE001
            tj = {}
        same_wl = (n, D, m, Cc, batch) == tuple(tj.get("workload", ())) if tj else False
        if pa_ms > 0:
            gbps = pa_bytes / (pa_ms * 1e-3) / 1e9
            r["roofline"] = {"bound": "hbm", "kernel": "k_scan_hist<64, 256, 1024> (pass A)", "achieved": round(gbps, 1), "peak": 8000.0, "unit": "GB/s",
                             "frac": round(gbps / 8000.0, 4), "algorithmic_bytes_per_launch": pa_bytes, "avg_launch_ms": round(pa_ms, 4),
                             "traffic": tj.get("k_scan_hist_fetch_bytes_per_launch") if same_wl else None,
                             "traffic_source": "profiles/hbm_traffic.json (FETCH_SIZE x 2, separate pass)" if same_wl else None,
                             "note": "bound by the LDS gather of the exact fp64 table (64 random 8-byte reads per code: 5 LDS cycles per wave-read with its "
                                     "bank conflicts = 0.4 of the HBM peak at 64-byte codes); the table itself comes prebuilt from k_lut_pre (128 KiB per "
                                     "query copied into LDS: 3 us of a 64 us item), not by HBM: DESIGN.md 5.2"}
        pb_ms_ = (st.scan_ms - st.passa_ms) / nd
        if int(st.passb_items_last) > 0 and pb_ms_ > 0:
            far_codes = (st.scan_codes - st.passa_codes) / nd  # codes of every probed far list (scanned or not)
            r["pass_b"] = {"kernel": "k_pair_smin_* (K3s) + k_scan_grp<64, 4, 16> (K3g) + hand-back", "ms": round(pb_ms_, 3),
                           "far_pairs_scanned_per_query": round(int(st.passb_items_last) / batch, 3),
                           "algorithmic_GBps_over_all_far_probes": round(far_codes * m / (pb_ms_ * 1e-3) / 1e9, 1)}
            if int(st.mfma_launches) > 0:  # K3mk (mmidx_scan_mfma_kc.h): the matrix-core bound over the pairs K3s left
                ml = int(st.mfma_launches)
                scan_ms = st.mfma_scan_ms / ml
                # the matrix-core work of the scan: 2 D flops per (kept pair, code of its list); the kept pairs' codes are taken as the
                # far probes' codes x the fraction of far pairs kept (exact when all are kept, as on the between-clusters leg)
                kept = int(st.passb_items_last) / max(1.0, batch * (w - 1.0))
                tf = 2.0 * D * far_codes * kept / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
                r["pass_b"]["roofline"] = {"bound": "mfma", "kernel": "k_scan_mfma_kc2 (K3mk)", "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s",
                                           "frac": round(tf / 2500.0, 4), "avg_launch_ms": round(scan_ms, 3),
                                           "traffic": tj.get("k_scan_mfma_kc2_fetch_bytes_per_launch") if (same_wl and w == 64 and kept > 0.9) else None,
                                           "note": "fp16 MFMA lower bound over all codes of the kept pairs' lists; co-limited by the LDS gather of the "
                                                   "decoded codebook rows (16 bytes per code, 8 dimensions and 32 queries; ~3-way bank conflicts "
                                                   "of random 16-byte rows), DESIGN.md 5.3"}
                r["pass_b"].update({"kernel": "k_pair_smin_* (K3s) + k_scan_mfma_kc2 (K3mk) + k_mfma_verify", "mfma_scan_launch_ms": round(scan_ms, 3),
                                    "mfma_verify_launch_ms": round(st.mfma_verify_ms / ml, 3),
                                    "mfma_survivors_per_query": round(st.mfma_survivors / nd / batch, 1),
                                    "mfma_redo_queries_per_step": round(st.mfma_redo_queries / nd, 1)})
        if ref is not None:
            ref.set_w(w)
            nsq = min(batch, parity_queries)
            t0 = time.perf_counter()
            rid, rd, rc = ref.search_batch(Qh[:nsq], k, nthreads=cpu_threads)
            ct = time.perf_counter() - t0
            fin = np.isfinite(rd)
            r["parity"] = {"queries": int(nsq), "ids_match": bool(np.array_equal(g_iid[:nsq], rid)),
                           "max_abs_ddist": float(np.max(np.abs(g_dd[:nsq][fin] - rd[fin]), initial=0.0))}
            r["cpu_port"] = {"queries_per_s": round(nsq / ct, 2), "threads": cpu_threads,
                             "one_query_ms_per_thread": round(ct / nsq * cpu_threads * 1e3, 1)}
        out[tag] = r
        log(f"{tag}: {r}")
    chk(L.mmidx_destroy(h))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=95_213_780)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--parity", type=int, default=512)
    ap.add_argument("--cells", type=int, default=8192)
    ap.add_argument("--w", type=int, nargs="*", default=[2, 64])
    ap.add_argument("--opt", action="append", default=[])
    a = ap.parse_args()
    print(json.dumps(run(n=a.n, Cc=a.cells, batch=a.batch, sigma=a.sigma, steps=a.steps, parity_queries=a.parity, ws=tuple(a.w),
                         opts=[tuple(x.split("=")) for x in a.opt])))
