#!/usr/bin/env python3
"""bench.py -- headline benchmark: queries/sec of IVFPQ search (IVFADC) on MI355X.

Workload (BASELINE.json configs[3], the configuration the metric is quoted on): IVFPQ over a
synthetic 100M x 128-d Gaussian-mixture base, 8192 coarse cells, nprobe w = 32, m = 16 x 256
sub-quantizers, k = 100.  One "step" = one pass of the hot path (coarse top-w -> residual LUTs ->
list scan -> top-k merge) over one batch of queries already resident in HBM.

  python bench.py --gpus 1 --steps K --warmup W            (defaults finish in a few minutes)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU; inverted lists are partitioned whole-list across ranks (cell mod N),
queries are replicated, the coarse assignment is split across ranks and all-gathered, the
per-shard top-(k+1) lists are sent to the query's owner rank (RCCL all-to-all over xGMI) and merged
there.  The index is the same 100M vectors for every N; the query batch per step is `--batch` x N
(each rank scans its 1/N of the lists for N x as many queries), so the work per GPU per step is
fixed and the scaling reported is "weak"; `--global-batch B` pins the batch instead ("strong").

The JSON line carries `roofline` (dominant kernels = the list scans, algorithmic bytes = m x scanned codes,
timed with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle -- a C restatement
of the Java reference, kind "port" -- timed on this host's cores on a bounded query sample).
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def gpu_kmeans(nat, L, dev_index, X, k, iters, seed=1, init=None, plus_plus=False):
    """codebook learning with the library's own k-means (csrc/mmidx_learn.hip; the reference uses Weka offline,
    J/quantization/*): X = fp64 CUDA tensor [n][d]; returns a [k][d] numpy array (empty clusters -> all-1000 rows,
    ProductQuantizationLearning.java:285-302)"""
    n, d = X.shape
    out = np.full((k, d), 1000.0)
    kout, it = C.c_int32(0), C.c_int32(0)
    ini = None if init is None else np.ascontiguousarray(init, np.float64)
    nat.check(L.mmidx_kmeans_device(dev_index, n, d, k, iters, seed, 1 if plus_plus else 0, X.data_ptr(),
                                    ini.ctypes.data if ini is not None else None, out.ctypes.data, None, None, C.addressof(it),
                                    C.addressof(kout), None))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=100_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--cells", type=int, default=8192)
    ap.add_argument("--w", type=int, default=32)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="queries per step over all ranks (default: --batch x n_gpus = fixed work per GPU)")
    ap.add_argument("--nbatches", type=int, default=4)
    ap.add_argument("--chunk", type=int, default=2_000_000)
    ap.add_argument("--gt", type=int, default=1024, help="queries with exact ground truth (recall@1)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--sigma", type=float, default=0.15, help="mixture noise (0.15 = SURVEY 8d's generator)")
    ap.add_argument("--settle", type=int, default=24)
    ap.add_argument("--exhaustive-steps", type=int, default=3,
                    help="extra untimed-for-value steps with pruning off, reported as roofline_exhaustive (0 = skip)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=INT",
                    help="mmidx_set_option(NAME, INT) on the index before the timed steps (kernel A/B switches)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU code path (encode -> owner filter -> add_codes, two-phase shard search, "
                         "merge) even with one rank: exercises it on a single GPU")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: everything else that writes to fd 1 (RCCL prints its
    # version banner there) is routed to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch

    mi = importlib.import_module("multimedia-indexing_amd")
    nat = importlib.import_module("multimedia-indexing_amd._native")
    L = mi.lib()
    if not torch.cuda.is_available() or L.mmidx_device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: libmmidx_hip has no CPU fallback")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or (args.force_sharded and "MASTER_ADDR" in os.environ):
        # (--force-sharded under torch.distributed.run with one rank: the RCCL calls run on a 1-rank group)
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=dev)

    N, D, Cc, w, m, k = args.n, args.dim, args.cells, args.w, args.m, args.k
    B = args.global_batch if args.global_batch > 0 else args.batch * world
    scaling = "strong" if args.global_batch > 0 else "weak"
    ks, dsub, K1 = 256, D // m, args.k + 1
    f64 = torch.float64
    stream = torch.cuda.current_stream().cuda_stream

    def chk(st):
        nat.check(st)

    # ---------------------------------------------------------------- codebooks (offline learning)
    t0 = time.time()
    g0 = torch.Generator(device=dev)
    g0.manual_seed(1234)
    mu = torch.randn(Cc, D, generator=g0, device=dev, dtype=f64)
    ns = min(N, 1 << 20)
    gs = torch.randint(0, Cc, (ns,), generator=g0, device=dev)
    Xs = mu[gs] + args.sigma * torch.randn(ns, D, generator=g0, device=dev, dtype=f64)
    torch.cuda.synchronize()
    # coarse quantizer: Lloyd from the mixture means (2 iterations); residual PQ codebooks: k-means++ per sub-space on
    # centroid - vector (ResidualVectorComputation.java:34)
    coarse_h0 = gpu_kmeans(nat, L, local, Xs, Cc, 2, init=mu.cpu().numpy())
    coarse = torch.from_numpy(coarse_h0).to(dev)
    hq = C.c_void_p()
    chk(L.mmidx_create(nat.KIND_IVFPQ, D, 1, 2, Cc, 0, None, None, local, C.byref(hq)))
    chk(L.mmidx_set_coarse(hq, coarse_h0.ctypes.data))
    nr = min(ns, 1 << 18)
    cell_s = torch.empty(nr, dtype=torch.int32, device=dev)
    chk(L.mmidx_assign_device(hq, nr, Xs.data_ptr(), cell_s.data_ptr(), stream))
    torch.cuda.synchronize()
    chk(L.mmidx_destroy(hq))
    resid = coarse[cell_s.long()] - Xs[:nr]
    pq = torch.empty(m, ks, dsub, device=dev, dtype=f64)
    for s in range(m):
        sub = resid[:, s * dsub:(s + 1) * dsub].contiguous()
        torch.cuda.synchronize()
        pq[s] = torch.from_numpy(gpu_kmeans(nat, L, local, sub, ks, 8, seed=s + 1, plus_plus=True)).to(dev)
    del Xs, resid
    log(f"codebooks learned in {time.time() - t0:.1f}s")

    # ---------------------------------------------------------------- index
    h = C.c_void_p()
    chk(L.mmidx_create(nat.KIND_IVFPQ, D, m, ks, Cc, 0, None, None, local, C.byref(h)))
    coarse_h = coarse.cpu().numpy().copy()
    pq_h = pq.cpu().numpy()
    chk(L.mmidx_set_coarse(h, coarse_h.ctypes.data))
    chk(L.mmidx_set_pq(h, pq_h.ctypes.data))
    chk(L.mmidx_set_w(h, w))
    for o_ in args.opt:
        name_, val_ = o_.split("=")
        chk(L.mmidx_set_option(h, name_.encode(), int(val_)))

    sh_owner = importlib.import_module("multimedia-indexing_amd.sharded").owner_of_cell
    nq_total = B * args.nbatches
    gq = torch.Generator(device=dev)
    gq.manual_seed(4321)
    qsrc = torch.randint(0, N, (nq_total,), generator=gq, device=dev)
    Qsrc = torch.zeros(nq_total, D, device=dev, dtype=f64)
    ngt = min(args.gt, B)
    gt_best = torch.full((ngt,), float("inf"), device=dev, dtype=f64)
    gt_arg = torch.full((ngt,), -1, device=dev, dtype=torch.long)

    t0 = time.time()
    t_enc = 0.0
    for c0 in range(0, N, args.chunk):
        n = min(args.chunk, N - c0)
        gc = torch.Generator(device=dev)
        gc.manual_seed(10_000 + c0 // args.chunk)
        g = torch.randint(0, Cc, (n,), generator=gc, device=dev)
        X = mu[g]
        X += args.sigma * torch.randn(n, D, generator=gc, device=dev, dtype=f64)
        sel = (qsrc >= c0) & (qsrc < c0 + n)
        if sel.any():
            Qsrc[sel] = X[qsrc[sel] - c0]
        torch.cuda.synchronize()
        te = time.time()
        if world == 1 and not args.force_sharded:
            chk(L.mmidx_add_vectors_device(h, n, X.data_ptr(), None, c0, stream))
        else:
            cells = torch.empty(n, dtype=torch.int32, device=dev)
            codes = torch.empty(n, m, dtype=torch.int8, device=dev)
            chk(L.mmidx_encode_device(h, n, X.data_ptr(), cells.data_ptr(), codes.data_ptr(), stream))
            own = sh_owner(cells, world) == rank
            iids = (torch.arange(n, device=dev, dtype=torch.int32) + c0)[own].contiguous()
            oc = cells[own].contiguous()
            ok = codes[own].contiguous()
            torch.cuda.synchronize()
            chk(L.mmidx_add_codes_device(h, iids.numel(), iids.data_ptr(), oc.data_ptr(), ok.data_ptr(), stream))
        torch.cuda.synchronize()
        t_enc += time.time() - te
        del g, X
    chk(L.mmidx_sync_index(h))
    torch.cuda.synchronize()
    log(f"index built: {N} vectors in {time.time() - t0:.1f}s (encode+append {t_enc:.1f}s)")

    Q = Qsrc + 0.01 * torch.randn(nq_total, D, generator=gq, device=dev, dtype=f64)
    Qb = [Q[i * B:(i + 1) * B].contiguous() for i in range(args.nbatches)]

    # exact fp64 brute-force ground truth (Linear semantics) for recall@1: regenerate the chunks
    t0 = time.time()
    if rank == 0 and ngt > 0:
        Qg = Qb[0][:ngt]
        qn = (Qg * Qg).sum(1)
        for c0 in range(0, N, args.chunk):
            n = min(args.chunk, N - c0)
            gc = torch.Generator(device=dev)
            gc.manual_seed(10_000 + c0 // args.chunk)
            g = torch.randint(0, Cc, (n,), generator=gc, device=dev)
            X = mu[g]
            X += args.sigma * torch.randn(n, D, generator=gc, device=dev, dtype=f64)
            dm = (X * X).sum(1)[None, :] - 2.0 * (Qg @ X.T) + qn[:, None]
            bv, bi = dm.min(1)
            upd = bv < gt_best
            gt_best[upd] = bv[upd]
            gt_arg[upd] = bi[upd] + c0
            del X, dm, g
        log(f"ground truth for {ngt} queries in {time.time() - t0:.1f}s")

    # ---------------------------------------------------------------- the step
    iid_out = torch.empty(B, k, dtype=torch.int32, device=dev)
    dist_out = torch.empty(B, k, dtype=f64, device=dev)
    cnt_out = torch.empty(B, dtype=torch.int32, device=dev)
    sharded = None
    if world > 1 or args.force_sharded:
        sh = importlib.import_module("multimedia-indexing_amd.sharded")
        sharded = sh.ShardedIVFPQ(sh.HipShardEngine(h, D, w, local), rank, world, dist=dist, force_collectives=args.force_sharded)

    def step(Qx):
        if sharded is None:
            chk(L.mmidx_search_device(h, k, B, Qx.data_ptr(), iid_out.data_ptr(), dist_out.data_ptr(), cnt_out.data_ptr(), stream))
            return
        # results stay on the rank that owns the query slice (rank r: queries [r*B/N, (r+1)*B/N))
        i_, d_, c_ = sharded.search(k, Qx, gather=False)
        iid_out[:i_.shape[0]].copy_(i_)
        dist_out[:i_.shape[0]].copy_(d_)
        cnt_out[:i_.shape[0]].copy_(c_)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # settle the device before the W warm-up steps the caller asked for (clocks, caches, the pass-B launch hint): with a
    # small W the first timed steps otherwise run ~10 % slower than the steady state
    for i in range(args.settle):
        step(Qb[i % args.nbatches])
    barrier()
    for i in range(args.warmup):
        step(Qb[i % args.nbatches])
    barrier()
    # the timed region carries only the two HIP events around pass A of every step (the dominant kernel's launch time for
    # the roofline object; resolved after the region): every event record is a ~5 us bubble in the stream
    chk(L.mmidx_set_profiling(h, 2))
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(Qb[i % args.nbatches])
    barrier()
    elapsed = time.perf_counter() - t0
    st_light = nat.Stats()
    chk(L.mmidx_get_stats(h, C.byref(st_light)))
    # stage times and code counters: the same batches once more with full profiling, outside the timed region
    chk(L.mmidx_set_profiling(h, 1))
    detail_steps = max(args.nbatches, min(args.steps, 8))
    for i in range(detail_steps):
        step(Qb[i % args.nbatches])
    barrier()
    st = nat.Stats()
    chk(L.mmidx_get_stats(h, C.byref(st)))
    chk(L.mmidx_set_profiling(h, 0))
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=f64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    qps = B * args.steps / elapsed

    # the same steps with every exact shortcut switched off (each probed code read and summed in fp64):
    # the configuration on which the scan kernel's HBM roofline fraction is a meaningful figure
    exhaustive = None
    if sharded is None and args.exhaustive_steps > 0:
        chk(L.mmidx_set_option(h, b"exhaustive", 1))
        step(Qb[0])
        barrier()
        chk(L.mmidx_set_profiling(h, 1))
        t0 = time.perf_counter()
        for i in range(args.exhaustive_steps):
            step(Qb[i % args.nbatches])
        barrier()
        ex_t = time.perf_counter() - t0
        ex = nat.Stats()
        chk(L.mmidx_get_stats(h, C.byref(ex)))
        chk(L.mmidx_set_profiling(h, 0))
        chk(L.mmidx_set_option(h, b"exhaustive", 0))
        ex_bytes = float(m) * ex.scan_codes
        ex_ach = ex_bytes / (ex.scan_ms * 1e-3) / 1e9 if ex.scan_ms > 0 else 0.0
        exhaustive = {"steps": args.exhaustive_steps, "queries_per_s": round(B * args.exhaustive_steps / ex_t, 1),
                      "scan_ms_per_step": round(ex.scan_ms / args.exhaustive_steps, 4), "achieved": round(ex_ach, 1), "peak": 8000.0,
                      "unit": "GB/s", "frac": round(ex_ach / 8000.0, 4), "frac_of_measured_copy_ceiling": round(ex_ach / 6290.0, 4), "bound": "lds (bank conflicts of the fp64 gather), see DESIGN.md 5.1"}

    # recall@1 and results of batch 0 (for the parity gate)
    step(Qb[0])
    torch.cuda.synchronize()
    res_iid = iid_out.cpu().numpy().copy()
    res_dist = dist_out.cpu().numpy().copy()
    recall1 = None
    if rank == 0 and ngt > 0:
        recall1 = float((iid_out[:ngt, 0].long() == gt_arg).double().mean().item())

    launches = max(1, st.scan_launches)
    scan_ms = st.scan_ms / launches
    alg_bytes = float(m) * st.scan_codes / launches
    achieved = alg_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    # physical HBM bytes per launch: PMC FETCH_SIZE (x2 on gfx950) from the committed profile of this
    # exact workload; PMC collection cannot run inside the timed region, so the figure is per query
    traffic = None
    traffic_passa = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        wl = tj["workload"]
        if world == 1 and (wl["n"], wl["dim"], wl["cells"], wl["nprobe"], wl["m"], wl["k"]) == (N, D, Cc, w, m, k):
            traffic = tj["hbm_bytes_per_query"] * B / max(1, launches / max(1, detail_steps))
            traffic_passa = tj["k_scan_hist_fetch_kib_per_step"] * 1024.0 * 2.0 * B / wl["batch"]
    except Exception:
        traffic = None
    # The dominant kernel is pass A (k_scan_hist: every query's nearest list, read and summed exactly: ~62 % of the step).
    # Its algorithmic bytes are m x the codes of those lists -- no pruning is involved, so this is a plain HBM roofline
    # fraction: the kernel is bound by the LDS gather and the VALU work around it, not by HBM.
    pa_ms = st_light.passa_ms / max(1, st_light.passa_launches)  # (timed region)
    pa_bytes = float(m) * st.passa_codes / max(1, st.passa_launches)  # (per launch; the detail run covers the same batches)
    pa_ach = pa_bytes / (pa_ms * 1e-3) / 1e9 if pa_ms > 0 else 0.0
    if sharded is None and st.passa_launches > 0:
        roofline = {"bound": "hbm", "kernel": "k_scan_hist (pass A: the exact scan of every query's nearest list; the launch also "
                                               "carries its empty hand-back launch)",
                    "achieved": round(pa_ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(pa_ach / 8000.0, 4),
                    "frac_of_measured_copy_ceiling": round(pa_ach / 6290.0, 4), "traffic": traffic_passa,
                    "algorithmic_bytes_per_launch": pa_bytes, "avg_launch_ms": round(pa_ms, 4), "launches": int(st_light.passa_launches),
                    "note": "limited by the LDS gather (16 fp64 table entries per code, ~60 % of its LDS cycles are bank conflicts) "
                            "and the VALU work around it (both pipes ~75 % busy, profiles/r01p_pmc_kernels.txt), not by HBM"}
    else:
        roofline = None
    # the whole search in algorithmic bytes (every probed list counted, although the coarse bound and the lower-bound filter
    # keep almost all of them from being read): how far exact pruning takes the path beyond what HBM could stream
    whole = {"bound": "hbm", "kernel": "all scan launches: k_scan_hist (pass A) + k_scan_filt (pass B)", "achieved": round(achieved, 1),
             "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
             "note": "achieved = algorithmic bytes (m x probed codes) / scan-kernel time; exact pruning (coarse bound, "
                     "Smin >= T) and L2 reuse make it exceed the physical HBM rate: see traffic and DESIGN.md section 7",
             "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(scan_ms, 4),
             "bytes_per_query": alg_bytes / B, "scan_launches": int(st.scan_launches),
             "coarse_ms_per_step": round(st.coarse_ms / max(1, detail_steps), 4),
             "merge_ms_per_step": round(st.merge_ms / max(1, detail_steps), 4),
             "measured": f"{detail_steps} steps with full profiling after the timed region"}
    if roofline is None:
        roofline = whole

    # ---------------------------------------------------------------- CPU baseline + parity gate
    cpu_baseline, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import oracle as o

        t0 = time.time()
        off = np.zeros(Cc + 1, np.int64)
        chk(L.mmidx_export(h, off.ctypes.data, None, None))
        n_exp = int(off[-1])
        iids = np.empty(n_exp, np.int32)
        codes = np.empty((n_exp, m), np.int8)
        chk(L.mmidx_export(h, off.ctypes.data, iids.ctypes.data, codes.ctypes.data))
        ref = o.OracleIndex(o.KIND_IVFPQ, D, m, ks, Cc)
        ref.set_coarse(coarse_h)
        ref.set_pq(pq_h)
        ref.set_w(w)
        ref.load_lists(off, iids, codes)
        del iids, codes
        # threads = the CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota
        # (a 256-thread box with a 16-CPU quota throttles 256 runnable threads to 16 CPUs' worth of time)
        logical = os.cpu_count() or 1
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else logical
        quota = None
        try:
            qv, pv = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if qv != "max":
                quota = float(qv) / float(pv)
        except (OSError, ValueError):
            try:
                qv = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
                pv = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if qv > 0:
                    quota = qv / pv
            except (OSError, ValueError):
                pass
        if quota is not None:
            cores = max(1, min(cores, int(quota + 0.999)))
        Qh = Qb[0].cpu().numpy()
        tc = time.perf_counter()
        ref.search_batch(Qh[:cores], k, nthreads=cores)  # calibration
        per_round = max(time.perf_counter() - tc, 1e-4)
        nsamp = int(min(B, max(cores, cores * int(args.cpu_seconds / per_round))))
        tc = time.perf_counter()
        rid, rd, rc = ref.search_batch(Qh[:nsamp], k, nthreads=cores)
        cpu_t = time.perf_counter() - tc
        # one reader thread (the reference's single call), a couple of seconds of work
        tc = time.perf_counter()
        ref.search_batch(Qh[:2], k, nthreads=1)
        per_q = max((time.perf_counter() - tc) / 2, 1e-4)
        n1 = int(min(nsamp, max(2, 2.0 / per_q)))
        tc = time.perf_counter()
        r1 = ref.search_batch(Qh[:n1], k, nthreads=1)
        one_t = time.perf_counter() - tc
        one_ok = bool(np.array_equal(r1[0], rid[:n1]))
        cpu_model = "unknown"
        try:
            for ln in open("/proc/cpuinfo"):
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        cpu_baseline = {"value": round(nsamp / cpu_t, 2), "unit": "queries/s", "cores": cores, "kind": "port",
                        "sample": f"{nsamp} queries of batch 0 (same index, k={k}, w={w}), C restatement of "
                                  f"IVFPQ.computeKnnIVFADC, {cores} concurrent reader threads "
                                  f"({logical} logical CPUs, cgroup quota {quota if quota is not None else 'none'}), {cpu_t:.1f}s",
                        "cpu_model": cpu_model,
                        "one_thread": {"value": round(n1 / one_t, 2), "unit": "queries/s", "queries": n1,
                                       "same_results_as_threaded": one_ok}}
        ids_match = bool(np.array_equal(res_iid[:nsamp], rid))
        fin = np.isfinite(rd)
        maxd = float(np.max(np.abs(res_dist[:nsamp][fin] - rd[fin]), initial=0.0))
        parity = {"queries": nsamp, "ids_match": ids_match, "max_abs_ddist": maxd}
        log(f"cpu baseline + parity in {time.time() - t0:.1f}s: {cpu_baseline['value']} q/s on {cores} cores; parity {parity}")

    if rank == 0:
        out = {
            "metric": "queries/sec @ recall@1, IVFPQ 100Mx128-d nprobe=32",
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"IVFPQ {N}x{D}-d, {Cc} coarse cells, nprobe w={w}, m={m}x{ks}, k={k}, batch {B} queries/step",
                       "n": N, "dim": D, "cells": Cc, "nprobe": w, "m": m, "ks": ks, "k": k, "batch": B, "batch_per_gpu": B // world,
                       "sharding": "single GPU" if world == 1 else f"whole inverted lists, cell mod {world}; RCCL: all-gather probe cells, MIN all-reduce thresholds, "
                                                                      f"all-to-all partial top-k to the query's owner rank, merge there"},
            "recall_at_1": recall1, "recall_queries": ngt, "sigma": args.sigma, "passb_items_last": int(st.passb_items_last), "verified_codes_per_step": int(st.verified_codes) // max(1, detail_steps),
            "roofline": roofline, "roofline_whole_search": whole, "roofline_exhaustive": exhaustive,
            "cpu_baseline": cpu_baseline, "parity": parity,
        }
        print(json.dumps(out), file=json_out, flush=True)
    chk(L.mmidx_destroy(h))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
