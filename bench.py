#!/usr/bin/env python3
"""bench.py -- headline benchmark: queries/sec of IVFPQ search (IVFADC) on MI355X.

Workload (BASELINE.json configs[3], the configuration the metric is quoted on): IVFPQ over a
synthetic 100M x 128-d Gaussian-mixture base, 8192 coarse cells, nprobe w = 32, m = 16 x 256
sub-quantizers, k = 100.  One "step" = one pass of the hot path (coarse top-w -> residual LUTs ->
list scan -> top-k merge) over one batch of queries already resident in HBM.

  python bench.py --gpus 1 --steps K --warmup W            (defaults finish in a few minutes)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

The JSON line carries, next to the headline (SURVEY 8d's generator: mixture noise 0.15, on which the exact coarse
bound removes 31 of the 32 probes):
  `roofline`      the dominant kernel of the headline step (pass A, k_scan_hist), HIP events on the launch stream;
  `hard`          the same engine on data that defeats the coarse bound (mixture noise --hard-sigma: clusters overlap,
                  every one of the 32 probes goes through the filtered scan k_scan_grp), with its own qps, roofline
                  object, recall and an oracle parity gate over >= 1024 queries of the 100M index;
  `cpu_baseline`  the CPU oracle (a C restatement of the Java reference, kind "port") on this host's cores;
  `other_configs` BASELINE configs 1-3 at their stated sizes (tests/bench_configs.py), parity-gated.

N > 1: ONE process drives all N GPUs through the library's own sharded handle (mmidx_create_sharded, include/mmidx.h: one
host thread and one RCCL communicator per device inside libmmidx_hip.so) -- the shape of the reference's caller, a single
JVM holding the whole index (YFCC100MExample.java:93-99).  Under `torch.distributed.run --nproc-per-node N` rank 0 is that
process and the other ranks wait for it; should the native handle fail to come up, all ranks fall back to the older
one-process-per-GPU path over torch.distributed (`--torch-dist` selects it outright), and the JSON line says which ran.
Inverted lists are partitioned whole-list across shards (cell mod N), every shard owns 1/N of the step's queries (their
vectors and probe cells are all-gathered, thresholds MIN-all-reduced, per-shard top-(k+1) lists stored into the owner's HBM
over xGMI and merged there).  The index is the same 100M vectors for every N; the query batch per step is `--batch` x N
(each GPU scans its 1/N of the lists for N x as many queries), so the work per GPU per step is fixed and the scaling
reported is "weak"; `--global-batch B` pins the batch instead ("strong").  `--native-sharded` runs the sharded handle with
N = 1 too (one shard, a 1-rank RCCL communicator): what the multi-GPU machinery costs on one device.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import shutil
import sys
import time

if "OMP_NUM_THREADS" not in os.environ:  # (the box shows 256 logical CPUs and grants 16: an unbounded BLAS pool is throttled to a crawl)
    try:
        _q, _p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        _n = int(float(_q) / float(_p) + 0.999) if _q != "max" else (os.cpu_count() or 1)
    except (OSError, ValueError):
        _n = os.cpu_count() or 1
    for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(_v, str(max(1, min(_n, 64))))

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


class Ctx:
    """what every stage needs: the library, the device, the process group, the arguments"""


def gpu_kmeans(cx, X, k, iters, seed=1, init=None, plus_plus=False):
    """codebook learning with the library's own k-means (csrc/mmidx_learn.hip; the reference uses Weka offline,
    J/quantization/*): X = fp64 CUDA tensor [n][d]; returns a [k][d] numpy array (empty clusters -> all-1000 rows,
    ProductQuantizationLearning.java:285-302)"""
    n, d = X.shape
    out = np.full((k, d), 1000.0)
    kout, it = C.c_int32(0), C.c_int32(0)
    ini = None if init is None else np.ascontiguousarray(init, np.float64)
    cx.chk(cx.L.mmidx_kmeans_device(cx.local, n, d, k, iters, seed, 1 if plus_plus else 0, X.data_ptr(),
                                    ini.ctypes.data if ini is not None else None, out.ctypes.data, None, None, C.addressof(it),
                                    C.addressof(kout), None))
    return out


def same_codebooks(cx, coarse_h, pq_h):
    """every rank learns the codebooks from the same seeds; rank 0's copy is broadcast so that the shards agree bit for bit
    on the quantizers (list ownership = cell mod N) whatever the reduction order of a rank's k-means was"""
    if cx.dist is None:
        return coarse_h, pq_h
    torch = cx.torch
    tc, tp = torch.from_numpy(coarse_h).to(cx.dev), torch.from_numpy(pq_h).to(cx.dev)
    cx.dist.broadcast(tc, src=0)
    cx.dist.broadcast(tp, src=0)
    return tc.cpu().numpy(), tp.cpu().numpy()


def learn_codebooks(cx, sigma, iid=False):
    """mixture means, coarse quantizer (Lloyd from the means, 2 iterations) and residual PQ codebooks (k-means++ per
    sub-space on centroid - vector, ResidualVectorComputation.java:34).  iid: no mixture at all -- every vector is
    sigma N(0, I) (all "means" zero), the cells are plain k-means cells (Lloyd from sample points, 4 iterations)"""
    torch, a = cx.torch, cx.args
    N, D, Cc, m = a.n, a.dim, a.cells, a.m
    ks, dsub = 256, D // m
    t0 = time.time()
    g0 = torch.Generator(device=cx.dev)
    g0.manual_seed(1234)
    mu = torch.randn(Cc, D, generator=g0, device=cx.dev, dtype=torch.float64)
    if iid:
        mu.zero_()
    ns = min(N, 1 << 20)
    gs = torch.randint(0, Cc, (ns,), generator=g0, device=cx.dev)
    Xs = mu[gs] + sigma * torch.randn(ns, D, generator=g0, device=cx.dev, dtype=torch.float64)
    torch.cuda.synchronize()
    coarse_h = gpu_kmeans(cx, Xs, Cc, 4, init=Xs[:Cc].cpu().numpy()) if iid else gpu_kmeans(cx, Xs, Cc, 2, init=mu.cpu().numpy())
    coarse = torch.from_numpy(coarse_h).to(cx.dev)
    hq = C.c_void_p()
    cx.chk(cx.L.mmidx_create(cx.nat.KIND_IVFPQ, D, 1, 2, Cc, 0, None, None, cx.local, C.byref(hq)))
    cx.chk(cx.L.mmidx_set_coarse(hq, coarse_h.ctypes.data))
    nr = min(ns, 1 << 18)
    cell_s = torch.empty(nr, dtype=torch.int32, device=cx.dev)
    cx.chk(cx.L.mmidx_assign_device(hq, nr, Xs.data_ptr(), cell_s.data_ptr(), cx.stream))
    torch.cuda.synchronize()
    cx.chk(cx.L.mmidx_destroy(hq))
    resid = coarse[cell_s.long()] - Xs[:nr]
    pq = torch.empty(m, ks, dsub, device=cx.dev, dtype=torch.float64)
    for s in range(m):
        sub = resid[:, s * dsub:(s + 1) * dsub].contiguous()
        torch.cuda.synchronize()
        pq[s] = torch.from_numpy(gpu_kmeans(cx, sub, ks, 8, seed=s + 1, plus_plus=True)).to(cx.dev)
    log(f"codebooks (sigma {sigma}) learned in {time.time() - t0:.1f}s")
    return mu, coarse_h.copy(), pq.cpu().numpy()


def gen_chunk(cx, mu, sigma, c0, n, dev=None):
    """base vectors [c0, c0 + n): component g ~ U(cells), vector = mu_g + sigma N(0, I) (regenerable from the chunk seed;
    the Philox stream of a seed is the same on every device)"""
    torch = cx.torch
    dev = cx.dev if dev is None else dev
    gc = torch.Generator(device=dev)
    gc.manual_seed(10_000 + c0 // cx.args.chunk)
    g = torch.randint(0, cx.args.cells, (n,), generator=gc, device=dev)
    X = mu[g]
    X += sigma * torch.randn(n, cx.args.dim, generator=gc, device=dev, dtype=torch.float64)
    return X


def shard_device(r):
    """device of shard r of the native sharded handle: r, or 0 for every shard under MMIDX_BENCH_VIRTUAL_SHARDS=1 (virtual shards:
    the functional run of the N > 1 native path on a one-GPU box, tests/test_gpu_two_ranks.py -- never a measurement)"""
    return 0 if os.environ.get("MMIDX_BENCH_VIRTUAL_SHARDS") == "1" else r


def build_index_native(cx, mu, sigma, coarse_h, pq_h, nq_total, ndev):
    """The sharded handle over devices 0 .. ndev-1, built from this one process: in every round device r generates chunk
    round0 + r and hands it in as slice r (mmidx_add_vectors_sliced_device: encoded where it lies, records routed to the shard
    that owns their list, iids in batch order -> the same index as the single-GPU build).  Returns (handle, queries on device 0)."""
    torch, a, L, nat = cx.torch, cx.args, cx.L, cx.nat
    N, D, Cc, m = a.n, a.dim, a.cells, a.m
    nat.preload_rccl()
    h = C.c_void_p()
    devs = (C.c_int * ndev)(*[shard_device(r) for r in range(ndev)])
    cx.chk(L.mmidx_create_sharded(nat.KIND_IVFPQ, D, m, 256, Cc, 0, None, None, ndev, devs, C.byref(h)))
    cx.chk(L.mmidx_set_coarse(h, coarse_h.ctypes.data))
    cx.chk(L.mmidx_set_pq(h, pq_h.ctypes.data))
    cx.chk(L.mmidx_set_w(h, a.w))
    for o_ in a.opt:
        name_, val_ = o_.split("=")
        cx.chk(L.mmidx_set_option(h, name_.encode(), int(val_)))
    gq = torch.Generator(device=cx.dev)
    gq.manual_seed(4321)
    qsrc = torch.randint(0, N, (nq_total,), generator=gq, device=cx.dev)
    Qsrc = torch.zeros(nq_total, D, device=cx.dev, dtype=torch.float64)
    mus = [mu.to(torch.device("cuda", shard_device(r))) for r in range(ndev)]
    t0 = time.time()
    t_enc = 0.0
    nchunks = (N + a.chunk - 1) // a.chunk
    for round0 in range(0, nchunks, ndev):
        Xs, ns = [], []
        for r in range(ndev):
            ci = round0 + r
            c0 = ci * a.chunk
            n = max(0, min(a.chunk, N - c0)) if ci < nchunks else 0
            dv = torch.device("cuda", shard_device(r))
            X = gen_chunk(cx, mus[r], sigma, c0, n, dv) if n > 0 else torch.empty(0, D, device=dv, dtype=torch.float64)
            if n > 0:
                sel = (qsrc >= c0) & (qsrc < c0 + n)
                if sel.any():
                    Qsrc[sel] = X[(qsrc[sel] - c0).to(dv)].to(cx.dev)
            Xs.append(X)
            ns.append(n)
        for r in range(ndev):
            torch.cuda.synchronize(shard_device(r))
        te = time.time()
        cx.chk(L.mmidx_add_vectors_sliced_device(h, (C.c_int64 * ndev)(*ns), (C.c_void_p * ndev)(*[x.data_ptr() for x in Xs]),
                                                 round0 * a.chunk))
        t_enc += time.time() - te
        del Xs
    cx.chk(L.mmidx_sync_index(h))
    log(f"index (sigma {sigma}) built on {ndev} device(s) through the sharded handle: {N} vectors in {time.time() - t0:.1f}s "
        f"(encode + route + append {t_enc:.1f}s)")
    Q = Qsrc + 0.01 * torch.randn(nq_total, D, generator=gq, device=cx.dev, dtype=torch.float64)
    return h, Q


def build_index(cx, mu, sigma, coarse_h, pq_h, nq_total, sharded_build, independent_queries=False):
    """encode + append the N base vectors on the device; returns (handle, queries [nq_total][D]).
    Queries are self-perturbed base vectors: base[i] + 0.01 N(0, I); independent_queries: fresh draws from the
    base distribution instead (nobody's copy: the true neighbour is wherever it is)."""
    torch, a, L, nat = cx.torch, cx.args, cx.L, cx.nat
    N, D, Cc, m = a.n, a.dim, a.cells, a.m
    h = C.c_void_p()
    cx.chk(L.mmidx_create(nat.KIND_IVFPQ, D, m, 256, Cc, 0, None, None, cx.local, C.byref(h)))
    cx.chk(L.mmidx_set_coarse(h, coarse_h.ctypes.data))
    cx.chk(L.mmidx_set_pq(h, pq_h.ctypes.data))
    cx.chk(L.mmidx_set_w(h, a.w))
    for o_ in a.opt:
        name_, val_ = o_.split("=")
        cx.chk(L.mmidx_set_option(h, name_.encode(), int(val_)))
    sh_owner = importlib.import_module("multimedia-indexing_amd.sharded").owner_of_cell
    gq = torch.Generator(device=cx.dev)
    gq.manual_seed(4321)
    qsrc = torch.randint(0, N, (nq_total,), generator=gq, device=cx.dev)
    Qsrc = torch.zeros(nq_total, D, device=cx.dev, dtype=torch.float64)
    t0 = time.time()
    t_enc = 0.0
    W = cx.world if sharded_build else 1
    coll = sharded_build and cx.dist is not None
    nchunks = (N + a.chunk - 1) // a.chunk
    for round0 in range(0, nchunks, W):
        # sharded build: in every round rank r encodes chunk round0 + r (1/W of the vectors per rank, not all of them) and the
        # records travel to the rank that owns their inverted list (cell mod W) in one variable-size all-to-all per array
        ci = round0 + (cx.rank if sharded_build else 0)
        c0 = ci * a.chunk
        n = max(0, min(a.chunk, N - c0)) if ci < nchunks else 0
        X = gen_chunk(cx, mu, sigma, c0, n) if n > 0 else torch.empty(0, D, device=cx.dev, dtype=torch.float64)
        if n > 0:
            sel = (qsrc >= c0) & (qsrc < c0 + n)
            if sel.any():
                Qsrc[sel] = X[qsrc[sel] - c0]
        torch.cuda.synchronize()
        te = time.time()
        if not sharded_build:
            cx.chk(L.mmidx_add_vectors_device(h, n, X.data_ptr(), None, c0, cx.stream))
        else:
            cells = torch.empty(n, dtype=torch.int32, device=cx.dev)
            codes = torch.empty(n, m, dtype=torch.int8, device=cx.dev)
            if n > 0:
                cx.chk(L.mmidx_encode_device(h, n, X.data_ptr(), cells.data_ptr(), codes.data_ptr(), cx.stream))
            iids = torch.arange(n, device=cx.dev, dtype=torch.int32) + c0
            owner = sh_owner(cells, cx.world).long()
            order = torch.argsort(owner, stable=True)  # destination-major, arrival (iid) order kept inside a destination
            iids, cells, codes = iids[order].contiguous(), cells[order].contiguous(), codes[order].contiguous()
            send = torch.bincount(owner, minlength=cx.world)
            if coll:
                recv = torch.empty_like(send)
                cx.dist.all_to_all_single(recv, send)
                ssz, rsz = send.cpu().tolist(), recv.cpu().tolist()
                nr = int(sum(rsz))
                r_i = torch.empty(nr, dtype=torch.int32, device=cx.dev)
                r_c = torch.empty(nr, dtype=torch.int32, device=cx.dev)
                r_k = torch.empty(nr, m, dtype=torch.int8, device=cx.dev)
                cx.dist.all_to_all_single(r_i, iids, rsz, ssz)
                cx.dist.all_to_all_single(r_c, cells, rsz, ssz)
                cx.dist.all_to_all_single(r_k, codes, rsz, ssz)
                # chunks of one round arrive source-rank-major = ascending chunk index = ascending iid: arrival order kept
                iids, cells, codes = r_i, r_c, r_k
            torch.cuda.synchronize()
            if iids.numel():
                cx.chk(L.mmidx_add_codes_device(h, iids.numel(), iids.data_ptr(), cells.data_ptr(), codes.data_ptr(), cx.stream))
        torch.cuda.synchronize()
        t_enc += time.time() - te
        del X
    if coll:
        cx.dist.all_reduce(Qsrc)  # (every query's source vector was generated on exactly one rank)
    cx.chk(L.mmidx_sync_index(h))
    torch.cuda.synchronize()
    log(f"index (sigma {sigma}) built: {N} vectors in {time.time() - t0:.1f}s (encode+append {t_enc:.1f}s)")
    if independent_queries:
        gi = torch.randint(0, a.cells, (nq_total,), generator=gq, device=cx.dev)
        Q = mu[gi] + sigma * torch.randn(nq_total, D, generator=gq, device=cx.dev, dtype=torch.float64)
    else:
        Q = Qsrc + 0.01 * torch.randn(nq_total, D, generator=gq, device=cx.dev, dtype=torch.float64)
    return h, Q


def ground_truth(cx, mu, sigma, Qg):
    """exact fp64 brute-force nearest neighbour (Linear semantics) of every row of Qg: the chunks are regenerated"""
    torch, a = cx.torch, cx.args
    t0 = time.time()
    ngt = Qg.shape[0]
    best = torch.full((ngt,), float("inf"), device=cx.dev, dtype=torch.float64)
    arg = torch.full((ngt,), -1, device=cx.dev, dtype=torch.long)
    qn = (Qg * Qg).sum(1)
    for c0 in range(0, a.n, a.chunk):
        n = min(a.chunk, a.n - c0)
        X = gen_chunk(cx, mu, sigma, c0, n)
        dm = (X * X).sum(1)[None, :] - 2.0 * (Qg @ X.T) + qn[:, None]
        bv, bi = dm.min(1)
        upd = bv < best
        best[upd] = bv[upd]
        arg[upd] = bi[upd] + c0
        del X, dm
    torch.cuda.synchronize()
    log(f"ground truth for {ngt} queries in {time.time() - t0:.1f}s")
    return arg


def usable_cpus():
    """threads = the CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota
    (a 256-thread box with a 16-CPU quota throttles 256 runnable threads to 16 CPUs' worth of time)"""
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
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return cores, logical, quota, model


def oracle_of_index(cx, h, coarse_h, pq_h):
    """the CPU oracle loaded with the very codes the device index holds (mmidx_export = what loadIndexInMemory builds)"""
    from oracle import oracle as o

    a = cx.args
    off = np.zeros(a.cells + 1, np.int64)
    cx.chk(cx.L.mmidx_export(h, off.ctypes.data, None, None))
    n_exp = int(off[-1])
    iids = np.empty(n_exp, np.int32)
    codes = np.empty((n_exp, a.m), np.int8)
    cx.chk(cx.L.mmidx_export(h, off.ctypes.data, iids.ctypes.data, codes.ctypes.data))
    ref = o.OracleIndex(o.KIND_IVFPQ, a.dim, a.m, 256, a.cells)
    ref.set_coarse(coarse_h)
    ref.set_pq(pq_h)
    ref.set_w(a.w)
    ref.load_lists(off, iids, codes)
    # the duplicate-code rate of every generated index (a sample of 64 lists, exact within them)
    try:
        from tie_census import duplicate_code_rate

        rs = np.random.default_rng(7)
        lists = rs.choice(a.cells, size=min(64, a.cells), replace=False)
        rate, seen = duplicate_code_rate(off, codes, lists)
        ref.dup_info = {"duplicate_code_rate": rate, "codes_examined": seen, "lists_examined": int(len(lists))}
    except Exception as e:  # noqa: BLE001
        ref.dup_info = {"error": repr(e)}
    return ref


def parity_of(res_iid, res_dist, rid, rd):
    n = rid.shape[0]
    fin = np.isfinite(rd)
    return {"queries": int(n), "ids_match": bool(np.array_equal(res_iid[:n], rid)),
            "max_abs_ddist": float(np.max(np.abs(res_dist[:n][fin] - rd[fin]), initial=0.0))}


def java_probe():
    """is there a JVM on this box?  If there is, and MMIDX_REFERENCE_CLASSPATH names the reference jar and its four
    dependencies (LingPipe / Trove / BDB-JE / EJML: not in the image), tools/java_crosscheck/run.sh drives the REAL classes with the
    committed fixtures and compares ids and distance bits with the oracle's answers -- the run that would turn "parity unpinned"
    into "pinned".  Its verdict is recorded here; with no JDK nothing runs and the baseline stays the C restatement (kind 'port')."""
    import subprocess

    j, jc = shutil.which("java"), shutil.which("javac")
    out = {"java": j or "absent", "javac": jc or "absent"}
    if not (j and jc):
        out["note"] = "no JDK / reference jars on this box: the baseline is the C restatement (kind 'port'), never the Java classes"
        return out
    try:
        r = subprocess.run([os.path.join(ROOT, "tools", "java_crosscheck", "run.sh")], capture_output=True, text=True, timeout=600)
        out["crosscheck_rc"] = r.returncode
        out["crosscheck"] = (r.stdout + r.stderr)[-600:]
        out["note"] = {0: "JDK + reference jars present: the oracle's fixtures agree with the reference classes bit for bit",
                       2: "JDK present, reference jars not on MMIDX_REFERENCE_CLASSPATH: cross-check not run (tools/java_crosscheck/README.md)"}.get(
            r.returncode, "JDK present: the cross-check ran and FAILED -- see `crosscheck`")
    except Exception as e:  # noqa: BLE001
        out["note"] = f"JDK present, cross-check could not be started: {e!r}"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", "--vectors", dest="n", type=int, default=100_000_000)  # (--vectors: `--n` is an ambiguous prefix for torch.distributed.run)
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
    ap.add_argument("--sigma", type=float, default=0.15, help="mixture noise of the headline workload (0.15 = SURVEY 8d's generator)")
    ap.add_argument("--hard-sigma", type=float, default=1.0,
                    help="mixture noise of the `hard` object: cluster radius 11 against an inter-mean distance of 16, the coarse bound "
                         "removes no probe")
    ap.add_argument("--hard-steps", type=int, default=10, help="timed steps of the `hard` object (0 = skip it)")
    ap.add_argument("--hard-parity", type=int, default=2048, help="queries of the hard workload checked against the oracle")
    ap.add_argument("--spread-steps", type=int, default=6, help="timed steps of the `spread` object: iid Gaussian base, independent queries (0 = skip it)")
    ap.add_argument("--spread-parity", type=int, default=1024, help="queries of the spread workload checked against the oracle")
    ap.add_argument("--other-configs", type=int, default=1, help="1: also run BASELINE configs 1-3 (tests/bench_configs.py)")
    ap.add_argument("--extras", type=int, default=1,
                    help="1: also publish host_path, cfg5, yfcc and the measured ceilings (tests/bench_extras.py, tests/bench_yfcc.py)")
    ap.add_argument("--yfcc-n", type=int, default=95_213_780, help="vectors of the `yfcc` object (0 = skip it)")
    ap.add_argument("--cfg5-images", type=int, default=1_000_000, help="images of cfg5's end-to-end run (0 = front-end kernels only)")
    ap.add_argument("--big-batch", type=int, default=131072,
                    help="queries per step of the `batch_131072` object (same index, one GPU: >= 8 queries per nearest list, pass A through K3ma; 0 = skip it)")
    ap.add_argument("--big-steps", type=int, default=10)
    ap.add_argument("--dry-run-shards", type=int, default=8,
                    help="N = 1 only: also run the `--gpus N` path with N in-process shards on this one device (`sharded_dry_run`; 0 = skip it)")
    ap.add_argument("--settle", type=int, default=24)
    ap.add_argument("--exhaustive-steps", type=int, default=3,
                    help="extra untimed-for-value steps with pruning off, reported as roofline_exhaustive (0 = skip)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=INT",
                    help="mmidx_set_option(NAME, INT) on the index before the timed steps (kernel A/B switches)")
    ap.add_argument("--extra-out", default=os.path.join(ROOT, "bench_extra.json"),
                    help="file that receives every side measurement (hard / spread / other configs / cfg5 / yfcc / host path ...); the JSON line names it")
    ap.add_argument("--dump", default="", metavar="PREFIX", help="write batch 0's answers of every rank to PREFIX.rank<r>.npz")
    ap.add_argument("--sharded-self-test", action="store_true",
                    help="(internal) build a small index on the sharded handle over --gpus devices and on a plain handle, compare the answers, exit 0 / 1")
    ap.add_argument("--native-sharded", action="store_true",
                    help="drive the library's own sharded handle (mmidx_create_sharded: worker threads + RCCL inside libmmidx_hip.so) "
                         "from this one process; the default for --gpus > 1, with --gpus 1 it measures one shard on a 1-rank communicator")
    ap.add_argument("--torch-dist", action="store_true",
                    help="--gpus > 1: the one-process-per-GPU path over torch.distributed (multimedia-indexing_amd/sharded.py) instead")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU code path (encode -> owner filter -> add_codes, two-phase shard search, "
                         "merge) even with one rank: exercises it on a single GPU")
    args = ap.parse_args()
    if args.sharded_self_test:
        try:
            import torch

            torch.cuda.init()
        except Exception:  # noqa: BLE001
            pass
        raise SystemExit(sharded_self_test(args.gpus))

    # stdout carries exactly one JSON line: everything else that writes to fd 1 (RCCL prints its
    # version banner there) is routed to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch

    cx = Ctx()
    cx.torch, cx.args = torch, args
    cx.mi = importlib.import_module("multimedia-indexing_amd")
    cx.nat = nat = importlib.import_module("multimedia-indexing_amd._native")
    cx.L = L = cx.mi.lib()
    cx.chk = chk = nat.check
    if not torch.cuda.is_available() or L.mmidx_device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: libmmidx_hip has no CPU fallback")

    cx.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    cx.rank = rank = int(os.environ.get("RANK", "0"))
    # MMIDX_BENCH_ONE_GPU=1 (tests/test_gpu_two_ranks.py): every rank on GPU 0, gloo collectives staged through the host --
    # a functional run of the N > 1 path on a one-GPU box (RCCL refuses two ranks on one device); never a measurement
    one_gpu = os.environ.get("MMIDX_BENCH_ONE_GPU") == "1"
    cx.local = local = 0 if one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    # N > 1: rank 0 drives all N devices through the library's sharded handle; the other ranks wait for its verdict
    # (a plain `python bench.py --gpus N` without torch.distributed.run works as well: there is only that one process)
    native = (args.native_sharded or args.gpus > 1) and not args.torch_dist and not args.force_sharded and not one_gpu
    if args.gpus > 1 and world != args.gpus and not (native and world == 1):
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    ndev = args.gpus if native else 1
    fallback_reason = None
    if native and torch.cuda.device_count() < ndev and os.environ.get("MMIDX_BENCH_VIRTUAL_SHARDS") != "1":
        # (every rank sees the same count and takes the same turn: one process cannot reach all the devices, so one process per GPU)
        if world == ndev:
            fallback_reason = f"this process sees {torch.cuda.device_count()} of the {ndev} devices: one process per GPU instead"
            native, ndev = False, 1
        else:
            raise SystemExit(f"--gpus {ndev} through the native sharded handle needs {ndev} visible devices, this process sees {torch.cuda.device_count()}")
    dist = None
    if world > 1 or (args.force_sharded and "MASTER_ADDR" in os.environ):
        # (--force-sharded under torch.distributed.run with one rank: the RCCL calls run on a 1-rank group)
        import torch.distributed as dist

        if one_gpu:
            dist.init_process_group("gloo")
            dist = importlib.import_module("multimedia-indexing_amd.sharded").HostStagedDist(dist)
        elif native:
            # gloo for the verdict below (CPU tensors); RCCL comes up lazily, only if the ranks have to fall back
            dist.init_process_group("cpu:gloo,cuda:nccl")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    virtual = os.environ.get("MMIDX_BENCH_VIRTUAL_SHARDS") == "1"
    if native and world == 1 and ndev > 1 and not virtual:
        why = run_sharded_self_test(ndev)
        if why:
            raise SystemExit(f"{why}; run under torch.distributed.run for the one-process-per-GPU path")
    if native and world > 1:
        verdict = torch.zeros(1, dtype=torch.int32)
        if rank == 0:
            try:
                why = None if virtual else run_sharded_self_test(ndev)
                if why:
                    raise RuntimeError(why)
                run(cx, args, json_out, native=True, ndev=ndev, dist=None, rank=0, world=1, local=0)
                verdict[0] = 1
            except BaseException as e:  # noqa: BLE001 -- whatever went wrong, the other ranks must hear about it
                fallback_reason = repr(e)
                log(f"native sharded handle failed ({fallback_reason}): falling back to the torch.distributed path")
        dist.broadcast(verdict, src=0)  # (a CPU tensor: gloo -- RCCL is not brought up between the ranks unless they fall back)
        if int(verdict[0]) == 1:
            dist.destroy_process_group()
            return
        native = False
    run(cx, args, json_out, native=native, ndev=ndev, dist=dist, rank=rank, world=world, local=local, fallback_reason=fallback_reason)


def sharded_self_test(ndev):
    """The first thing a multi-GPU run does (in a subprocess, under a timeout, so that a hang in a collective ends as a reported
    fallback reason): one small IVFPQ index on the sharded handle over `ndev` physical devices and on a plain handle, the same
    queries through both -- ids, distance bits and counts must be identical (ties included: every vector is indexed twice)."""
    import numpy as np

    mi = importlib.import_module("multimedia-indexing_amd")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth

    D, C_, m, ks, w, k = 32, 64, 8, 256, 9, 20
    p = synth.make_ivfpq_problem(n=6000, D=D, C=C_, m=m, ks=ks, nq=96, seed=7, iters=3)
    base = np.concatenate([p["base"], p["base"][:3000]])
    out = []
    for devs in (None, list(range(ndev))):
        ix = mi.IVFPQ(D, len(base), False, "", m, ks, 0, C_, 512, **({} if devs is None else {"devices": devs}))
        ix.loadCoarseQuantizer(p["coarse"])
        ix.loadProductQuantizer(p["pq"])
        ix.setW(w)
        ix.indexVectors([str(i) for i in range(len(base))], base)
        if devs is not None:
            ix.set_option("tie_slots", 4)
        out.append([ix.search_batch(k, p["queries"]), ix.search_batch(1, p["queries"][:1])])
        ix.close()
    for a, b in zip(out[0], out[1]):
        for x, y in zip(a, b):
            if not np.array_equal(x, y):
                print("sharded self-test: answers differ from the plain handle's", file=sys.stderr)
                return 1
    print(f"sharded self-test over {ndev} devices: identical to the plain handle", file=sys.stderr)
    return 0


def run_sharded_self_test(ndev, timeout_s=240):
    """None when the sharded handle over ndev devices passes its self-test, else the reason (a string)"""
    import subprocess

    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--sharded-self-test", "--gpus", str(ndev)], timeout=timeout_s,
                           capture_output=True, text=True)
    except subprocess.TimeoutExpired:
        return f"sharded self-test over {ndev} devices did not finish in {timeout_s}s (a collective hung?)"
    if r.returncode != 0:
        return f"sharded self-test over {ndev} devices failed (rc {r.returncode}): {(r.stderr or '').strip()[-300:]}"
    return None


def run(cx, args, json_out, native, ndev, dist, rank, world, local, fallback_reason=None):
    torch, L, nat, chk = cx.torch, cx.L, cx.nat, cx.chk
    one_gpu = os.environ.get("MMIDX_BENCH_ONE_GPU") == "1"
    cx.rank, cx.world, cx.local = rank, world, local
    if native:
        world = ndev  # (one process, ndev shards: B, per_rank and the roofline are per device all the same)
    torch.cuda.set_device(local)
    cx.dev = dev = torch.device("cuda", local)
    cx.dist = dist

    N, D, Cc, w, m, k = args.n, args.dim, args.cells, args.w, args.m, args.k
    B = args.global_batch if args.global_batch > 0 else args.batch * world
    scaling = "strong" if args.global_batch > 0 else "weak"
    ks = 256
    f64 = torch.float64
    cx.stream = stream = torch.cuda.current_stream().cuda_stream
    single = world == 1 and not args.force_sharded and not native

    # measured ceilings (a few tens of ms): what the LDS pipe delivers for pass A's gather pattern, the f64 matrix peak
    probes = None
    if rank == 0 and args.extras:
        try:
            probes = importlib.import_module("bench_extras").probes(L, nat, local)
            log(f"probes: {probes}")
        except Exception as e:  # noqa: BLE001
            probes = {"error": repr(e)}

    # ---------------------------------------------------------------- headline workload
    mu, coarse_h, pq_h = learn_codebooks(cx, args.sigma)
    coarse_h, pq_h = same_codebooks(cx, coarse_h, pq_h)
    nq_total = B * args.nbatches
    big_B = args.big_batch if (world == 1 and not native and not args.force_sharded and args.big_batch > B) else 0
    nq_total = max(nq_total, big_B)
    rccl_ranks = None
    if native:
        h, Q = build_index_native(cx, mu, args.sigma, coarse_h, pq_h, nq_total, ndev)
        # the first thing a multi-GPU record should prove: how many shards the handle has, on which devices, and whether its
        # collectives run on RCCL (one rank per device inside the library) -- straight from mmidx_shard_info
        try:
            nsh = C.c_int(0)
            chk(L.mmidx_shard_count(h, C.byref(nsh)))
            devs_seen, rccl_flag = [], 0
            for r_ in range(nsh.value):
                dv, sz, ur = C.c_int(0), C.c_int64(0), C.c_int(0)
                chk(L.mmidx_shard_info(h, r_, C.byref(dv), C.byref(sz), C.byref(ur)))
                devs_seen.append(dv.value)
                rccl_flag = max(rccl_flag, ur.value)
            rccl_ranks = {"shards": nsh.value, "devices": devs_seen, "distinct_devices": len(set(devs_seen)), "uses_rccl": bool(rccl_flag),
                          "rccl_ranks": nsh.value if rccl_flag else 0}
            log(f"native sharded handle: {rccl_ranks}")
        except Exception as e:  # noqa: BLE001
            rccl_ranks = {"error": repr(e)}
    else:
        h, Q = build_index(cx, mu, args.sigma, coarse_h, pq_h, nq_total, sharded_build=not single)
    Qb = [Q[i * B:(i + 1) * B].contiguous() for i in range(args.nbatches)]
    ngt = min(args.gt, B)
    gt_arg = ground_truth(cx, mu, args.sigma, Qb[0][:ngt]) if rank == 0 and ngt > 0 else None

    iid_out = torch.empty(B, k, dtype=torch.int32, device=dev)
    dist_out = torch.empty(B, k, dtype=f64, device=dev)
    cnt_out = torch.empty(B, dtype=torch.int32, device=dev)
    sharded = None
    per_rank = B // world
    nslices = None
    if native:
        # every shard's slice of every batch, and its answers, resident in that shard's HBM before the clock starts
        dvs = [torch.device("cuda", shard_device(r)) for r in range(ndev)]
        nslices = [[Qx[r * per_rank:(r + 1) * per_rank].to(dvs[r]).contiguous() for r in range(ndev)] for Qx in Qb]
        n_iid = [torch.empty(per_rank, k, dtype=torch.int32, device=dvs[r]) for r in range(ndev)]
        n_dist = [torch.empty(per_rank, k, dtype=f64, device=dvs[r]) for r in range(ndev)]
        n_cnt = [torch.empty(per_rank, dtype=torch.int32, device=dvs[r]) for r in range(ndev)]
        parr = lambda ts: (C.c_void_p * ndev)(*[t.data_ptr() for t in ts])
        n_args = (parr(n_iid), parr(n_dist), parr(n_cnt))
        n_q = {id(Qx): parr(sl) for Qx, sl in zip(Qb, nslices)}
    if not single and not native:
        sh = importlib.import_module("multimedia-indexing_amd.sharded")
        sharded = sh.ShardedIVFPQ(sh.HipShardEngine(h, D, w, local), rank, world, dist=dist, force_collectives=args.force_sharded,
                                  tie_slots=int(os.environ.get("MMIDX_SHARD_TIE_SLOTS", "32")),
                                  pipeline={"0": False, "1": True}.get(os.environ.get("MMIDX_SHARD_PIPELINE", ""), None))

    def step(Qx, hh=None):
        if native and hh is None:
            chk(L.mmidx_search_sliced_device(h, k, per_rank, n_q[id(Qx)], n_args[0], n_args[1], n_args[2]))
            return
        if sharded is None or hh is not None:
            chk(L.mmidx_search_device(hh if hh is not None else h, k, B, Qx.data_ptr(), iid_out.data_ptr(), dist_out.data_ptr(),
                                      cnt_out.data_ptr(), stream))
            return
        # every rank hands in the B / N queries it owns (rank r: rows [r*B/N, (r+1)*B/N) of the step's batch); the query
        # vectors are exchanged INSIDE the step (all-gather), and the answers stay with the owner
        i_, d_, c_ = sharded.search_owned(k, Qx[rank * per_rank:(rank + 1) * per_rank])
        iid_out[:i_.shape[0]].copy_(i_)
        dist_out[:i_.shape[0]].copy_(d_)
        cnt_out[:i_.shape[0]].copy_(c_)

    def barrier():
        for r in range(ndev if native else 1):
            torch.cuda.synchronize(shard_device(r) if native else None)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def gather_native():
        """the native handle's answers (slices on the shards' devices) into the result tensors on device 0"""
        if native:
            for r in range(ndev):
                iid_out[r * per_rank:(r + 1) * per_rank].copy_(n_iid[r])
                dist_out[r * per_rank:(r + 1) * per_rank].copy_(n_dist[r])
                cnt_out[r * per_rank:(r + 1) * per_rank].copy_(n_cnt[r])
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
    if st.passa_mfma_launches > 0:  # K3ma served pass A (batches of >= 8 queries per list): its stages, HIP events of the detail steps
        nl = st.passa_mfma_launches
        log(f"K3ma pass A per step: sweep 1 {st.passa_mfma_sweep1_ms / nl:.3f} ms, select {st.passa_mfma_select_ms / nl:.3f}, sweep 2 "
            f"{st.passa_mfma_sweep2_ms / nl:.3f}, verify {st.passa_mfma_verify_ms / nl:.3f}; {st.verified_codes / max(1, detail_steps) / B:.1f} codes verified per query, "
            f"{st.mfma_redo_queries / max(1, detail_steps):.1f} queries per step handed to the exact kernels")
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=f64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    qps = B * args.steps / elapsed
    if args.dump:  # the answers of batch 0 as this rank holds them (its B / N queries): compared across world sizes by the tests
        step(Qb[0])
        barrier()
        gather_native()
        n_own = B if sharded is None else per_rank
        np.savez(f"{args.dump}.rank{rank}.npz", iid=iid_out[:n_own].cpu().numpy(), dist=dist_out[:n_own].cpu().numpy(),
                 cnt=cnt_out[:n_own].cpu().numpy())

    # the same steps with every exact shortcut switched off (each probed code read and summed in fp64):
    # the configuration on which the exact scan kernel's HBM roofline fraction is a meaningful figure
    exhaustive = None
    if single and args.exhaustive_steps > 0:
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
                      "unit": "GB/s", "frac": round(ex_ach / 8000.0, 4), "frac_of_measured_copy_ceiling": round(ex_ach / 6290.0, 4),
                      "bound": "lds (bank conflicts of the fp64 gather), see DESIGN.md 5.1"}

    # recall@1 and results of batch 0 (for the parity gate)
    step(Qb[0])
    barrier()
    gather_native()
    res_iid = iid_out.cpu().numpy().copy()
    res_dist = dist_out.cpu().numpy().copy()
    recall1 = None
    if rank == 0 and ngt > 0:
        recall1 = float((iid_out[:ngt, 0].long() == gt_arg).double().mean().item())

    launches = max(1, st.scan_launches)
    scan_ms = st.scan_ms / launches
    alg_bytes = float(m) * st.scan_codes / launches
    achieved = alg_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    # physical HBM bytes per launch: PMC FETCH_SIZE (x2 on gfx950) from the committed profile of this exact workload --
    # PMC collection cannot run inside the timed region, so the figure is read from profiles/, not measured in this run
    traffic = None
    traffic_passa = None
    traffic_source = None
    # which kernel family served pass A of the timed steps (mmidx_get_dispatch; K3q since round 6 from 1.25 queries per non-empty list)
    passa_kernel = "?"
    try:
        buf = C.create_string_buffer(512)
        chk(L.mmidx_get_dispatch(h, buf, 512))
        passa_kernel = dict(kv.split("=", 1) for kv in buf.value.decode().split(";")).get("pass_a", "?")
    except Exception:  # noqa: BLE001
        pass
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        wl = tj["workload"]
        if world == 1 and (wl["n"], wl["dim"], wl["cells"], wl["nprobe"], wl["m"], wl["k"]) == (N, D, Cc, w, m, k) and args.sigma == 0.15:
            pa = tj.get("pass_a", {}).get(passa_kernel)  # {fetch_kib_per_step, write_kib_per_step, kernel, source}
            if pa is not None:
                traffic_passa = (pa["fetch_kib_per_step"] * 2.0 + pa.get("write_kib_per_step", 0.0)) * 1024.0 * B / wl["batch"]
                traffic = traffic_passa
                traffic_source = ("profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE of %s on this workload, "
                                  "separate pass (%s) -- not measured in this run" % (pa.get("kernel", passa_kernel), pa.get("source", "profiles/")))
    except Exception:
        traffic = None
    # The dominant kernel of the headline step is pass A (k_scan_hist: every query's nearest list, read and summed exactly:
    # ~65 % of the step).  Its algorithmic bytes are m x the codes of those lists -- no pruning is involved, so this is a plain
    # HBM roofline fraction: the kernel is bound by the LDS gather and the VALU work around it, not by HBM.
    pa_ms = st_light.passa_ms / max(1, st_light.passa_launches)  # (timed region)
    pa_bytes = float(m) * st.passa_codes / max(1, st.passa_launches)  # (per launch; the detail run covers the same batches)
    if native:
        pa_bytes /= ndev  # (stats of a sharded handle: times = the slowest shard, code counts = the sum over shards -> per device)
    pa_ach = pa_bytes / (pa_ms * 1e-3) / 1e9 if pa_ms > 0 else 0.0
    # the LDS roofline of the same kernel: its gather alone (mmidx_probe_lds_gather: m random ds_read_b64 per code over 2 KiB
    # rows + the fp64 adds, three codes in flight per lane, four 256-thread blocks per CU as K3h runs) measured in this run
    lds_roof = None
    if isinstance(probes, dict) and f"lds_gather_m{m}_chains3" in probes:
        pk = probes[f"lds_gather_m{m}_chains3"]["algorithmic_GBps"]
        lds_roof = {"bound": "lds", "achieved": round(pa_ach, 1), "peak": pk, "unit": "GB/s of codes (m random 8-byte table reads per code)",
                    "frac": round(pa_ach / pk, 4) if pk > 0 else None,
                    "peak_source": "mmidx_probe_lds_gather(m, 3 chains): the scan loop with only the gather and the adds left, measured in this run",
                    "wave_gathers_per_s_peak": probes[f"lds_gather_m{m}_chains3"]["wave_gathers_per_s"]}
    if sharded is None and st.passa_launches > 0:
        kdesc = {"K3q": "k_scan_q (K3q, pass A: every query's nearest list, four queries of a list per block -- packed u16 tables, integer "
                        "top-6 per lane, exact fp64 sums for ~K1 + 10 candidates per query; the launch carries its pair sort and its empty hand-back launch)",
                 "K3h": "k_scan_hist (K3h, pass A: the exact scan of every query's nearest list, a block per query; the launch also carries its empty "
                        "hand-back launch)",
                 "K3ma": "k_scan_mfma<.., 1/2> + k_a1_* (K3ma, pass A on the matrix cores)"}.get(passa_kernel, passa_kernel)
        knote = {"K3q": "algorithmic bytes = m x the codes of EVERY query's nearest list (SURVEY 8d); K3q reads a list from HBM once per block of up to "
                        "four queries, so the physical traffic (`traffic`) is below the algorithmic figure; the scan is bound by the LDS gather of the packed "
                        "table (one random 8-byte read per code and sub-quantizer for four queries) and the VALU work of the sorted insertions, DESIGN.md 5.2",
                 "K3h": "limited by the LDS gather (16 fp64 table entries per code, ~60 % of its LDS cycles are bank conflicts) and the VALU work around it "
                        "(both pipes ~75 % busy, profiles/), not by HBM"}.get(passa_kernel, "")
        roofline = {"bound": "hbm", "kernel": kdesc, "pass_a_kernel": passa_kernel,
                    "achieved": round(pa_ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(pa_ach / 8000.0, 4),
                    "frac_of_measured_copy_ceiling": round(pa_ach / 6290.0, 4), "traffic": traffic_passa, "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": pa_bytes, "avg_launch_ms": round(pa_ms, 4), "launches": int(st_light.passa_launches),
                    "lds": lds_roof if passa_kernel == "K3h" else None,
                    "note": knote}
    else:
        roofline = None
    # the whole search in algorithmic bytes (every probed list counted, although the coarse bound and the lower-bound filter
    # keep almost all of them from being read): how far exact pruning takes the path beyond what HBM could stream
    step_bytes = float(m) * st.scan_codes / max(1, detail_steps)  # (per step: every launch of a step)
    whole = {"bound": "hbm", "kernel": "all scan launches of a step: k_scan_hist (pass A) + k_scan_grp / k_scan_filt (pass B)",
             "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
             "traffic_source": traffic_source,
             "note": "achieved = algorithmic bytes (m x probed codes) / scan-kernel time; exact pruning (coarse bound, "
                     "Smin >= T) and L2 reuse make it exceed the physical HBM rate: see traffic and DESIGN.md section 7",
             "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(scan_ms, 4),
             "algorithmic_bytes_per_step": step_bytes, "bytes_per_query": step_bytes / B, "scan_launches": int(st.scan_launches),
             "coarse_ms_per_step": round(st.coarse_ms / max(1, detail_steps), 4),
             "merge_ms_per_step": round(st.merge_ms / max(1, detail_steps), 4),
             "passb_pairs_per_query": round(int(st.passb_items_last) / B, 3),
             "measured": f"{detail_steps} steps with full profiling after the timed region"}
    if roofline is None:
        roofline = whole

    # ---------------------------------------------------------------- the same index at a batch of 131072 (K3ma)
    def big_batch(ref):
        """One GPU at >= 8 queries per nearest list (16 at 131072 / 8192): what every shard of the 8-GPU configuration sees per round.
        Pass A goes through K3ma (csrc/mmidx_scan_mfma_a.h): the list-major matrix-core bound in two sweeps, exact fp64 for ~K1 + 10 %
        codes per query.  Same index, same engine, same answers: the first `parity` queries are checked against the oracle."""
        Qx = Q[:big_B].contiguous()
        bi = torch.empty(big_B, k, dtype=torch.int32, device=dev)
        bd = torch.empty(big_B, k, dtype=f64, device=dev)
        bc = torch.empty(big_B, dtype=torch.int32, device=dev)

        def bstep():
            chk(L.mmidx_search_device(h, k, big_B, Qx.data_ptr(), bi.data_ptr(), bd.data_ptr(), bc.data_ptr(), stream))

        for _ in range(4):
            bstep()
        torch.cuda.synchronize()
        chk(L.mmidx_set_profiling(h, 2))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.big_steps):
            bstep()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        sl = nat.Stats()
        chk(L.mmidx_get_stats(h, C.byref(sl)))
        chk(L.mmidx_set_profiling(h, 1))
        nd = 4
        for _ in range(nd):
            bstep()
        torch.cuda.synchronize()
        sd = nat.Stats()
        chk(L.mmidx_get_stats(h, C.byref(sd)))
        chk(L.mmidx_set_profiling(h, 0))
        nl = max(1, sd.passa_mfma_launches)
        pa_ms_b = sl.passa_ms / max(1, sl.passa_launches)
        codes = float(sd.passa_codes) / max(1, sd.passa_launches)  # codes of the nearest lists, summed over the queries, per step
        sweeps_ms = (sd.passa_mfma_sweep1_ms + sd.passa_mfma_sweep2_ms) / nl
        flops = 2.0 * (2.0 * D * codes)  # two sweeps, 2 D flops per (query, code of its nearest list)
        tf = flops / (sweeps_ms * 1e-3) / 1e12 if sweeps_ms > 0 else 0.0
        alg = float(m) * codes
        par = None
        if ref is not None:
            npar = min(2048, big_B)
            rid, rd, rc = ref.search_batch(Qx[:npar].cpu().numpy(), k, nthreads=cores)
            par = parity_of(bi[:npar].cpu().numpy(), bd[:npar].cpu().numpy(), rid, rd)
        rec1 = float((bi[:ngt, 0].long() == gt_arg).double().mean().item()) if (ngt > 0 and gt_arg is not None) else None
        tr = None
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get("batch_131072")
        except Exception:  # noqa: BLE001
            pass
        k3ma = sd.passa_mfma_launches > 0
        if k3ma:
            roof = {"bound": "mfma", "kernel": "k_scan_mfma<.., 1> + k_scan_mfma<.., 2> (K3ma's two sweeps over every query's nearest list: fp16 MFMA "
                                              "bound, list-major, a code decoded once per <= 64 queries)",
                    "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
                    "flops_per_step": flops, "sweeps_ms": round(sweeps_ms, 4),
                    "algorithmic_bytes_per_step": alg, "pass_a_algorithmic_GBps": round(alg / (pa_ms_b * 1e-3) / 1e9, 1) if pa_ms_b > 0 else None,
                    "pass_a_frac_of_hbm_peak_algorithmic": round(alg / (pa_ms_b * 1e-3) / 8e12, 4) if pa_ms_b > 0 else None,
                    "traffic": tr.get("sweeps_fetch_bytes_per_step") if isinstance(tr, dict) else None,
                    "traffic_source": "profiles/hbm_traffic.json: FETCH_SIZE x 2 of the two sweep kernels, separate rocprofv3 --pmc pass" if isinstance(tr, dict) else None,
                    "note": "16 queries per list fill ONE 16-row tile: per tile of 16 codes 4 matrix instructions stand against 4 random 16-byte "
                            "decode gathers and ~50-65 vector instructions (slot updates / compares, addresses), which bound the sweeps "
                            "(profiles/r05b_b131k_pmc_kernels.txt); the algorithmic byte rate exceeds the HBM rate because a list is read once for "
                            "all the queries that probe it (physical traffic: `traffic`)"}
        else:  # K3q (the default since round 6): four queries of a list per block
            # at 16 queries per list the per-QUERY byte count (m x the codes of the query's nearest list) counts a list sixteen times; what a
            # launch has to read once is the lists the batch touches, once per block of four queries -- `achieved` is priced on that
            # (bytes the blocks request = algorithmic / 4 queries per block), the per-query figure stays beside it
            blk = alg / 4.0
            ach = blk / (pa_ms_b * 1e-3) / 1e9 if pa_ms_b > 0 else 0.0
            kq = tr.get("k_scan_q") if isinstance(tr, dict) else None
            roof = {"bound": "hbm", "kernel": "k_scan_q (K3q: every query's nearest list, four queries of a list per block; pair sort and empty hand-back launch included)",
                    "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
                    "algorithmic_bytes_per_step": blk, "per_query_bytes_per_step": alg,
                    "per_query_GBps": round(alg / (pa_ms_b * 1e-3) / 1e9, 1) if pa_ms_b > 0 else None, "avg_launch_ms": round(pa_ms_b, 4),
                    "traffic": (kq["fetch_kib_per_step"] * 2048.0 + kq.get("write_kib_per_step", 0.0) * 1024.0) if isinstance(kq, dict) else None,
                    "traffic_source": "profiles/hbm_traffic.json batch_131072.k_scan_q: FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 --pmc pass" if isinstance(kq, dict) else None,
                    "note": "algorithmic bytes = m x the codes of every BLOCK's list (16 queries per list = four blocks per list, each streaming it, the "
                            "later ones mostly from L2 / MALL: `traffic`); per query (m x the codes of its nearest list, the headline's definition) the "
                            "rate is `per_query_GBps`, above the HBM peak because a list serves four queries per read.  The kernel is bound by its LDS "
                            "reads and vector work, not by HBM (DESIGN.md 5.2)"}
        res = {"value": round(big_B * args.big_steps / el, 1), "unit": "queries/s", "steps": args.big_steps, "batch": big_B,
               "ms_per_step": round(el / args.big_steps * 1e3, 4), "queries_per_nearest_list": round(big_B / Cc, 2),
               "pass_a_kernel": "K3ma" if k3ma else "K3q",
               "passa_mfma_launches_per_step": round(sd.passa_mfma_launches / nd, 2),
               "stage_ms_per_step": {"coarse": round(sd.coarse_ms / nd, 4), "pass_a": round(sd.passa_ms / nd, 4),
                                     "pass_a_timed_region": round(pa_ms_b, 4), "scan_all": round(sd.scan_ms / nd, 4), "merge": round(sd.merge_ms / nd, 4)},
               "pass_a_stages_ms": {"sweep1": round(sd.passa_mfma_sweep1_ms / nl, 4), "select": round(sd.passa_mfma_select_ms / nl, 4),
                                    "sweep2": round(sd.passa_mfma_sweep2_ms / nl, 4),
                                    "rows_records_verify": round(sd.passa_mfma_verify_ms / nl, 4)} if k3ma else None,
               "verified_codes_per_query": round(sd.verified_codes / nd / big_B, 2),
               "queries_handed_to_exact_kernels_per_step": round(sd.mfma_redo_queries / nd, 2),
               "recall_at_1": rec1, "recall_queries": ngt,
               "roofline": roof,
               "parity": par}
        log(f"batch {big_B}: {res['value'] / 1e6:.2f} M q/s, {res['ms_per_step']} ms per step, pass A {pa_ms_b:.3f} ms (sweeps {sweeps_ms:.3f}), parity {par}")
        return res

    big = None
    # ---------------------------------------------------------------- CPU baseline + parity gate (headline index)
    cpu_baseline, parity, ties = None, None, None
    cores, logical, quota, cpu_model = usable_cpus()
    if rank == 0 and (world == 1 or native) and not args.no_cpu:
        t0 = time.time()
        ref = oracle_of_index(cx, h, coarse_h, pq_h)
        Qh = Qb[0].cpu().numpy()
        tc = time.perf_counter()
        ref.search_batch(Qh[:cores], k, nthreads=cores)  # calibration
        per_round = max(time.perf_counter() - tc, 1e-4)
        # (the CPU baseline is an N = 1 figure; with more devices the oracle only serves the parity gate, on a short sample)
        nsamp = int(min(B, max(cores, cores * int((args.cpu_seconds if world == 1 else 3.0) / per_round))))
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
        cpu_baseline = {"value": round(nsamp / cpu_t, 2), "unit": "queries/s", "cores": cores, "kind": "port",
                        "sample": f"{nsamp} queries of batch 0 (same index, k={k}, w={w}), C restatement of "
                                  f"IVFPQ.computeKnnIVFADC, {cores} concurrent reader threads "
                                  f"({logical} logical CPUs, cgroup quota {quota if quota is not None else 'none'}), {cpu_t:.1f}s",
                        "cpu_model": cpu_model, "java_probe": java_probe(),
                        "one_thread": {"value": round(n1 / one_t, 2), "unit": "queries/s", "queries": n1,
                                       "same_results_as_threaded": one_ok}}
        parity = parity_of(res_iid, res_dist, rid, rd)
        # ... and the exact ties among the top-(k + 1) of a parity sample: does the gate lean on the LingPipe assumption A1?
        try:
            from tie_census import tie_census

            nt = int(min(nsamp, 2048))
            _, td, _ = ref.search_batch(Qh[:nt], k + 1, nthreads=cores)
            ties = dict(tie_census(td, k), **getattr(ref, "dup_info", {}))
            ties["engine_tie_replays_per_step"] = int(st.tie_fallbacks) // max(1, detail_steps)
            ties["note"] = ("oracle top-(k+1) of the first %d parity queries; a tie at k would make MEMBERSHIP depend on assumption A1, a tie inside "
                            "the ORDER of the tied ids (tests/test_queue_rules_cpu.py)" % nt)
        except Exception as e:  # noqa: BLE001
            ties = {"error": repr(e)}
        if big_B > 0:
            try:
                big = big_batch(ref)
            except Exception as e:  # (never lose the headline line to a side measurement)
                big = {"error": repr(e)}
        del ref
        log(f"cpu baseline + parity in {time.time() - t0:.1f}s: {cpu_baseline['value']} q/s on {cores} cores; parity {parity}")

    if big is None and big_B > 0 and rank == 0:
        try:
            big = big_batch(None)
        except Exception as e:  # noqa: BLE001
            big = {"error": repr(e)}
    # ---------------------------------------------------------------- how even would the 8-GPU partition be?  
    # whole inverted lists go to shard `cell mod n`: a round's work per shard = the (query, probed list) pairs whose list lives there.
    # Counted from this index's own coarse stage -- a measurement of the partition, not of a multi-GPU run.
    balance = None
    if rank == 0 and single and h is not None:
        try:
            balance = {}
            for nqb in sorted({B, big_B} - {0}):
                Qx = Q[:nqb].contiguous()
                cells_t = torch.empty(nqb, w, dtype=torch.int32, device=dev)
                chk(L.mmidx_coarse_device(h, nqb, Qx.data_ptr(), cells_t.data_ptr(), None, stream))
                torch.cuda.synchronize()
                cl = cells_t.cpu().numpy()
                row = {}
                for ns_ in (2, 4, 8):
                    near = np.bincount(cl[:, 0] % ns_, minlength=ns_).astype(np.float64)
                    allp = np.bincount((cl % ns_).ravel(), minlength=ns_).astype(np.float64)
                    row[f"shards_{ns_}"] = {"nearest_list_pairs_max_over_mean": round(float(near.max() / near.mean()), 4),
                                            "all_probed_pairs_max_over_mean": round(float(allp.max() / allp.mean()), 4)}
                balance[f"queries_{nqb}"] = row
            balance["note"] = ("pairs per shard under the `cell mod n` partition of whole lists, from this index's coarse stage on the headline queries: "
                               "pass A's load = nearest-list pairs (the coarse bound drops the far probes on this generator), pass B's = all probed pairs")
        except Exception as e:  # noqa: BLE001
            balance = {"error": repr(e)}
    # ---------------------------------------------------------------- restart through the native snapshot (ABI 8: mmidx_save / mmidx_load)
    snapshot = None
    if rank == 0 and single and args.extras and h is not None:
        import tempfile

        snap_path = os.path.join(tempfile.gettempdir(), f"mmidx_bench_{os.getpid()}.snap")
        h2 = C.c_void_p()
        try:
            t0 = time.time()
            chk(L.mmidx_save(h, snap_path.encode()))
            t_save = time.time() - t0
            nbytes = os.path.getsize(snap_path)
            chk(L.mmidx_create(nat.KIND_IVFPQ, D, m, ks, Cc, 0, None, None, local, C.byref(h2)))
            chk(L.mmidx_set_coarse(h2, coarse_h.ctypes.data))
            chk(L.mmidx_set_pq(h2, pq_h.ctypes.data))
            chk(L.mmidx_set_w(h2, w))
            t0 = time.time()
            chk(L.mmidx_load(h2, snap_path.encode()))
            t_load = time.time() - t0
            # the reloaded index answers as the first one (the oracle comparison at this size is the parity gate above, and
            # tests/test_gpu_parity.py::test_native_snapshot_reload_1m_against_the_oracle at 1 M)
            i2 = torch.empty(B, k, dtype=torch.int32, device=dev)
            d2 = torch.empty(B, k, dtype=f64, device=dev)
            c2 = torch.empty(B, dtype=torch.int32, device=dev)
            chk(L.mmidx_search_device(h2, k, B, Qb[0].data_ptr(), i2.data_ptr(), d2.data_ptr(), c2.data_ptr(), stream))
            torch.cuda.synchronize()
            same = bool(np.array_equal(i2.cpu().numpy(), res_iid) and np.array_equal(d2.cpu().numpy(), res_dist))
            snapshot = {"records": int(N), "file_bytes": int(nbytes), "save_s": round(t_save, 2), "load_s": round(t_load, 2),
                        "same_answers_as_the_saved_index": same,
                        "note": "mmidx_save: export + one sequential write; mmidx_load: read + mmidx_add_codes in pieces of 16 M records + the CSR build "
                                "(the path GpuIVFPQ.loadSnapshot takes instead of loadIndexInMemory's BDB cursor, IVFPQ.java:680-728)"}
            log(f"snapshot: {snapshot}")
        except Exception as e:  # noqa: BLE001
            snapshot = {"error": repr(e)}
        finally:
            if h2:
                L.mmidx_destroy(h2)
            try:
                os.remove(snap_path)
            except OSError:
                pass
            torch.cuda.empty_cache()
    # ---------------------------------------------------------------- the boundary's own path: host buffers, caller threads
    host = None
    if rank == 0 and single and args.extras:
        try:
            t0 = time.time()
            host = importlib.import_module("bench_extras").host_path(L, nat, h, np.ascontiguousarray(Qb[0].cpu().numpy()), k)
            log(f"host path in {time.time() - t0:.1f}s: {host}")
        except Exception as e:  # noqa: BLE001
            host = {"error": repr(e)}

    # ---------------------------------------------------------------- side workloads on the same engine, same index size
    def side_workload(sig, iid, steps_, parity_q, what):
        """builds another 100M index (mixture noise `sig`, or iid N(0, I) vectors with independent queries), times `steps_` steps,
        reports stage times, recall against exact ground truth and an oracle parity gate"""
        mu_h, coarse_hh, pq_hh = learn_codebooks(cx, sig, iid=iid)
        hh, Qh_all = build_index(cx, mu_h, sig, coarse_hh, pq_hh, B * 2, sharded_build=False, independent_queries=iid)
        Qhb = [Qh_all[i * B:(i + 1) * B].contiguous() for i in range(2)]
        ngh = min(args.gt, B, 256)
        gt_h = ground_truth(cx, mu_h, sig, Qhb[0][:ngh]) if ngh > 0 else None
        for i in range(4):
            step(Qhb[i % 2], hh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps_):
            step(Qhb[i % 2], hh)
        torch.cuda.synchronize()
        h_el = time.perf_counter() - t0
        chk(L.mmidx_set_profiling(hh, 1))
        hd = 4
        for i in range(hd):
            step(Qhb[i % 2], hh)
        torch.cuda.synchronize()
        hst = nat.Stats()
        chk(L.mmidx_get_stats(hh, C.byref(hst)))
        chk(L.mmidx_set_profiling(hh, 0))
        step(Qhb[0], hh)
        torch.cuda.synchronize()
        h_iid = iid_out.cpu().numpy().copy()
        h_dist = dist_out.cpu().numpy().copy()
        h_recall = float((iid_out[:ngh, 0].long() == gt_h).double().mean().item()) if ngh > 0 else None
        h_recall_k = float((iid_out[:ngh] == gt_h[:, None].to(torch.int32)).any(1).double().mean().item()) if ngh > 0 else None
        # pass B = everything between the end of pass A and the end of the scans (pair sort + k_scan_grp + hand-back launch)
        pb_ms = (hst.scan_ms - hst.passa_ms) / hd
        pb_bytes = float(m) * (hst.scan_codes - hst.passa_codes) / hd
        pb_ach = pb_bytes / (pb_ms * 1e-3) / 1e9 if pb_ms > 0 else 0.0
        h_parity, h_cpu = None, None
        if not args.no_cpu:
            t0 = time.time()
            ref = oracle_of_index(cx, hh, coarse_hh, pq_hh)
            nsh = int(min(B, parity_q))
            Qn = Qhb[0][:nsh].cpu().numpy()
            tc = time.perf_counter()
            rid, rd, rc = ref.search_batch(Qn, k, nthreads=cores)
            h_cpu_t = time.perf_counter() - tc
            h_parity = parity_of(h_iid, h_dist, rid, rd)
            h_cpu = {"value": round(nsh / h_cpu_t, 2), "unit": "queries/s", "cores": cores, "kind": "port",
                     "sample": f"{nsh} queries of this workload, {h_cpu_t:.1f}s"}
            del ref
            log(f"{what}: oracle parity in {time.time() - t0:.1f}s: {h_parity}")
        traffic_pb = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            traffic_pb = tj.get(f"{what}_pass_b_fetch_bytes_per_launch")
        except Exception:  # noqa: BLE001
            pass
        # pass B's dominant kernel.  K3m (k_scan_mfma): a GEMM-shaped certified lower bound on the matrix cores -- priced against the
        # dense fp16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF); algorithmic flops = 2 D per (query, probed far code) pair.  Its launch
        # duration comes from HIP events around the kernel in the profiled steps (mmidx_stats::mfma_scan_ms).  The physical HBM
        # figure next to it is a separate PMC pass (profiles/hbm_traffic.json): a list is read from HBM about once per batch.
        pairs_pb = float(hst.scan_codes - hst.passa_codes) / hd
        if hst.mfma_launches > 0 and hst.mfma_scan_ms > 0:
            mf_ms = hst.mfma_scan_ms / hst.mfma_launches
            vf_ms = hst.mfma_verify_ms / hst.mfma_launches
            flops = 2.0 * D * pairs_pb
            tf = flops / (mf_ms * 1e-3) / 1e12
            pb_roofline = {"bound": "mfma", "kernel": "k_scan_mfma (pass B: fp16 MFMA lower bound of every (query, far code) pair, list-major, "
                                                      "one decode of a code per 64 queries; survivors verified in fp64 by k_mfma_verify)",
                           "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
                           "flops_per_launch": flops, "avg_launch_ms": round(mf_ms, 4), "verify_launch_ms": round(vf_ms, 4),
                           "pass_b_ms_incl_pair_sort_verify_redo": round(pb_ms, 4),
                           "algorithmic_bytes_per_launch": pb_bytes, "algorithmic_GBps": round(pb_bytes / (mf_ms * 1e-3) / 1e9, 1),
                           "traffic": traffic_pb,
                           "hbm_frac_physical": round(traffic_pb / (mf_ms * 1e-3) / 8e12, 4) if traffic_pb else None,
                           "traffic_source": "profiles/hbm_traffic.json (separate rocprofv3 --pmc FETCH_SIZE pass of this workload, x 2 on gfx950)" if traffic_pb else None,
                           "note": "peak = dense fp16/bf16 MFMA (MI355X_MICROARCH.md); besides the matrix work a tile of 16 codes costs four random 16-byte "
                                   "LDS gathers (the decode) and ~50 vector instructions per wave (compares, addresses), which co-limit the kernel "
                                   "(DESIGN.md 5.3)"}
        else:
            pb_roofline = {"bound": "issue", "kernel": "k_scan_grp (pass B: grouped u8 lower-bound filter; vector-instruction issue and LDS bound, DESIGN.md 5.12)",
                           "achieved": round(pb_ach, 1), "peak": None, "unit": "GB/s of algorithmic bytes", "frac": None,
                           "algorithmic_bytes_per_launch": pb_bytes, "avg_launch_ms": round(pb_ms, 4), "traffic": traffic_pb,
                           "hbm_frac_physical": round(traffic_pb / (pb_ms * 1e-3) / 8e12, 4) if traffic_pb else None}
        res = {"value": round(B * steps_ / h_el, 1), "unit": "queries/s", "steps": steps_,
               "ms_per_step": round(h_el / steps_ * 1e3, 4), "batch": B,
               "survivors_per_query": round(int(hst.passb_items_last) / B, 3), "far_probes_per_query": w - 1,
               "verified_codes_per_query": round(hst.verified_codes / hd / B, 2),
               "mfma_survivors_per_query": round(hst.mfma_survivors / hd / B, 2), "mfma_redo_queries_per_step": round(hst.mfma_redo_queries / hd, 2),
               "recall_at_1": h_recall, "true_neighbour_in_top_k": h_recall_k, "recall_queries": ngh,
               "stage_ms_per_step": {"coarse": round(hst.coarse_ms / hd, 4), "pass_a": round(hst.passa_ms / hd, 4), "pass_b": round(pb_ms, 4),
                                     "merge": round(hst.merge_ms / hd, 4)},
               "roofline": pb_roofline,
               "parity": h_parity, "cpu_baseline": h_cpu}
        chk(L.mmidx_destroy(hh))
        del Qh_all, Qhb
        torch.cuda.empty_cache()
        return res

    hard, spread = None, None
    if rank == 0 and single and (args.hard_steps > 0 or args.spread_steps > 0):
        chk(L.mmidx_destroy(h))
        h = None
        del Q, Qb
        torch.cuda.empty_cache()
    if rank == 0 and single and args.hard_steps > 0:
        hs = args.hard_sigma
        hard = side_workload(hs, False, args.hard_steps, args.hard_parity, "hard")
        hard["data"] = (f"same generator and means, mixture noise sigma = {hs} (headline: {args.sigma}): cluster radius "
                        f"{hs * D ** 0.5:.1f} against an inter-mean distance of {(2 * D) ** 0.5:.1f}; IVFPQ.java:414-447 scans all w probes, and here "
                        f"so does the engine")
    if rank == 0 and single and args.spread_steps > 0:
        # no cluster structure at all and queries that are nobody's copy: the top-k of a query is spread over several of its
        # probed cells, far probes FEED the queue (IVFPQ.java:429-446 offers every probed code) and recall@1 is what 16-byte
        # codes give on such data -- reported, not required ("iid-Gaussian independent queries would not reach 0.9")
        spread = side_workload(1.0, True, args.spread_steps, args.spread_parity, "spread")
        spread["data"] = ("iid N(0, I) base vectors (no mixture), k-means cells, INDEPENDENT N(0, I) queries: neighbours sit in several probed "
                          "cells, the filter's survivors are verified exactly; recall@1 is against exact fp64 brute force")

    # ---------------------------------------------------------------- BASELINE configs 1-3 at their stated sizes
    other = None
    if rank == 0 and single and args.other_configs and not args.no_cpu:
        try:
            t0 = time.time()
            bc = importlib.import_module("bench_configs")
            other = bc.run_all()
            log(f"other configs in {time.time() - t0:.1f}s")
        except Exception as e:  # (never lose the headline line to a side measurement)
            other = {"error": repr(e)}

    # ---------------------------------------------------------------- cfg5 front end, the reference's flagship shape
    cfg5, yfcc = None, None
    if rank == 0 and single and args.extras:
        if h is not None:
            chk(L.mmidx_destroy(h))
            h = None
        torch.cuda.empty_cache()
        try:
            t0 = time.time()
            cfg5 = importlib.import_module("bench_extras").cfg5(L, nat, cx.mi, images_e2e=args.cfg5_images, device=local)
            log(f"cfg5 in {time.time() - t0:.1f}s")
        except Exception as e:  # noqa: BLE001
            cfg5 = {"error": repr(e)}
        torch.cuda.empty_cache()
        if args.yfcc_n > 0:
            try:
                t0 = time.time()
                yfcc = importlib.import_module("bench_yfcc").run(n=args.yfcc_n, device=local, parity_queries=0 if args.no_cpu else 512)
                log(f"yfcc in {time.time() - t0:.1f}s")
            except Exception as e:  # noqa: BLE001
                yfcc = {"error": repr(e)}
        if isinstance(other, dict) and "error" not in other:
            other["cfg5_vlad_pca_ivfpq"] = cfg5

    # ---------------------------------------------------------------- the 8-GPU configuration's code path on this one device
    dry = None
    if rank == 0 and single and args.dry_run_shards > 1 and os.environ.get("MMIDX_BENCH_VIRTUAL_SHARDS") != "1":
        # `bench.py --gpus 8` exactly as the driver runs it, but with eight IN-PROCESS shards on device 0 (no xGMI, no RCCL: in-process
        # collectives): what each rank of the 8-GPU run computes per round, and a parity-gated exercise of the sharded handle.  It is
        # NOT a scaling measurement -- nothing here has run on more than one physical GPU.
        try:
            import subprocess
            if h is not None:
                chk(L.mmidx_destroy(h))
                h = None
            torch.cuda.empty_cache()
            t0 = time.time()
            ns = args.dry_run_shards
            cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(ns), "--steps", "6", "--warmup", "2", "--settle", "3", "--hard-steps", "0",
                   "--spread-steps", "0", "--other-configs", "0", "--extras", "0", "--yfcc-n", "0", "--cfg5-images", "0", "--exhaustive-steps", "0",
                   "--cpu-seconds", "2", "--dry-run-shards", "0", "--big-batch", "0", "--vectors", str(N), "--batch", str(args.batch),
                   "--extra-out", args.extra_out + ".dry_run"]
            pr = subprocess.run(cmd, env=dict(os.environ, MMIDX_BENCH_VIRTUAL_SHARDS="1"), capture_output=True, text=True, timeout=420)
            jl = [l for l in pr.stdout.splitlines() if l.startswith("{")]
            if pr.returncode == 0 and jl:
                dj = json.loads(jl[-1])
                try:  # (the child's side objects live in its own extras file)
                    dx = json.load(open(args.extra_out + ".dry_run"))
                    os.remove(args.extra_out + ".dry_run")
                except (OSError, ValueError):
                    dx = {}
                k3 = [l for l in pr.stderr.splitlines() if "K3ma pass A per step" in l]
                dry = {"n_virtual_shards": ns, "queries_per_s": dj["value"], "queries_per_round": dj["config"]["batch"], "ms_per_round": dj["ms_per_step"],
                       "ms_per_shard_and_round": round(dj["ms_per_step"] / ns, 4), "parity": dj.get("parity"),
                       "pass_a": k3[-1].split("] ", 1)[-1] if k3 else "%s (the first shard's dispatch, mmidx_get_dispatch)" % (dj.get("roofline") or {}).get("pass_a_kernel", "?"),
                       "stage_ms_slowest_shard": {"coarse": dx.get("roofline_whole_search", {}).get("coarse_ms_per_step"),
                                                  "merge": dx.get("roofline_whole_search", {}).get("merge_ms_per_step")},
                       "shard_info": dx.get("shard_info"),
                       "path": dj["config"]["multi_gpu_path"], "seconds": round(time.time() - t0, 1),
                       "note": f"{ns} in-process shards on ONE device (MMIDX_BENCH_VIRTUAL_SHARDS=1): the kernels, the partition, the threshold exchange and the "
                               "owner-side merge of the 8-GPU configuration run, sharing one GPU -- per-shard work per round, not a scaling figure; "
                               "UNMEASURED on more than one physical GPU"}
            else:
                dry = {"error": f"rc {pr.returncode}", "stderr_tail": pr.stderr[-400:]}
            log(f"sharded dry run ({ns} virtual shards) in {time.time() - t0:.1f}s: {dry.get('ms_per_shard_and_round')} ms per shard and round")
        except Exception as e:  # noqa: BLE001
            dry = {"error": repr(e)}

    if rank == 0:
        out = {
            "metric": "queries/sec @ recall@1, IVFPQ 100Mx128-d nprobe=32",
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"IVFPQ {N}x{D}-d, {Cc} coarse cells, nprobe w={w}, m={m}x{ks}, k={k}, batch {B} queries/step",
                       "n": N, "dim": D, "cells": Cc, "nprobe": w, "m": m, "ks": ks, "k": k, "batch": B, "batch_per_gpu": B // world,
                       "mixture_sigma": args.sigma,
                       "multi_gpu_path": ("native sharded handle (mmidx_create_sharded): one process, one host thread + one RCCL communicator per device "
                                          "inside libmmidx_hip.so" if native else
                                          ("none" if world == 1 and not args.force_sharded else "torch.distributed, one process per GPU (multimedia-indexing_amd/sharded.py)")),
                       "native_fallback_reason": fallback_reason, "rccl_ranks": rccl_ranks,
                       "sharding": "single GPU" if world == 1 and not native else
                                   (f"whole inverted lists, cell mod {world}; every shard owns batch/{world} queries; per step: RCCL all-gather of query vectors + probe "
                                    f"cells, RCCL MIN all-reduce of thresholds, partial top-(k+1) lists stored into the owner's HBM over xGMI (peer access), merge + "
                                    f"cross-shard tie replay (RCCL all-reduces) there" if native else
                                    f"whole inverted lists, cell mod {world}; every rank owns batch/{world} queries; RCCL per step: all-gather "
                                    f"query vectors + probe cells, MIN all-reduce thresholds, one variable-size all-to-all of partial "
                                    f"top-k entries to the query's owner, merge + cross-shard tie replay there")},
            "recall_at_1": recall1, "recall_queries": ngt,
            "roofline": roofline, "roofline_whole_search": whole, "roofline_exhaustive": exhaustive,
            "cpu_baseline": cpu_baseline, "parity": parity, "batch_131072": big, "hard": hard, "spread": spread, "other_configs": other, "cfg5": cfg5, "yfcc": yfcc,
            "host_path": host, "measured_ceilings": probes, "sharded_dry_run": dry, "shard_balance": balance, "snapshot_restart": snapshot,
            # the figures a reader should see NEXT to the headline: what a Java caller reaches through the JNI shim
            # (mmidx_search: host arrays in, host arrays out, synchronous) and what the engine does when the far probes have to be scanned
            "ties": ties,
            "host_buffers_qps": (host or {}).get("nq16384", {}).get("queries_per_s") if isinstance(host, dict) else None,
            "host_buffers_qps_3_callers": (host or {}).get("nq16384_callers3", {}).get("queries_per_s") if isinstance(host, dict) else None,
            "hard_qps": hard.get("value") if isinstance(hard, dict) else None,
            "spread_qps": spread.get("value") if isinstance(spread, dict) else None,
            "batch_131072_qps": big.get("value") if isinstance(big, dict) else None,
            "sharded_dry_run_ms_per_shard": dry.get("ms_per_shard_and_round") if isinstance(dry, dict) else None,
        }
        # every published figure once more as short brace-free `key=value` text on stderr BEFORE the result line (tests/bench_emit.py):
        # the result line is the last thing either stream carries
        def g(o, *ks):
            for kk in ks:
                if not isinstance(o, dict) or kk not in o or o[kk] is None:
                    return None
                o = o[kk]
            return o

        def mq(v):
            return None if v is None else round(v / 1e6, 3)

        def ok(p_):
            return None if not isinstance(p_, dict) else bool(p_.get("ids_match") and p_.get("max_abs_ddist") == 0.0)

        summ = {"headline_Mqps": mq(qps), "ms": round(ms_per_step, 4), "passA_ms": g(roofline, "avg_launch_ms"), "passA_frac": g(roofline, "frac"),
                "recall1": recall1, "parity": ok(parity), "cpu_qps": g(cpu_baseline, "value"), "cores": cores,
                "b131072": {"Mqps": mq(g(big, "value")), "ms": g(big, "ms_per_step"), "passA_ms": g(big, "stage_ms_per_step", "pass_a_timed_region"),
                            "sweeps_ms": g(big, "roofline", "sweeps_ms"), "passA_frac": g(big, "roofline", "frac"), "parity": ok(g(big, "parity"))},
                "hard": {"Mqps": mq(g(hard, "value")), "passB_ms": g(hard, "stage_ms_per_step", "pass_b"), "mfma_frac": g(hard, "roofline", "frac"),
                         "parity": ok(g(hard, "parity"))},
                "spread": {"Mqps": mq(g(spread, "value")), "passB_ms": g(spread, "stage_ms_per_step", "pass_b"), "mfma_frac": g(spread, "roofline", "frac"),
                           "recall1": g(spread, "recall_at_1"), "parity": ok(g(spread, "parity"))},
                "exhaustive_frac": g(exhaustive, "frac"),
                "dry8": {"Mqps": mq(g(dry, "queries_per_s")), "ms_per_shard": g(dry, "ms_per_shard_and_round"), "parity": ok(g(dry, "parity"))},
                "cfg1_Mqps": mq(g(other, "cfg1_linear_10k", "qps_gpu_host_buffers")), "cfg2_Mqps": mq(g(other, "cfg2_pq_adc_1M", "qps_gpu")),
                "cfg3_Mqps": mq(g(other, "cfg3_ivfpq_1M", "qps_gpu")), "rot1M_Mqps": mq(g(other, "ivfpq_1M_random_rotation", "qps_gpu")),
                "m128_Mqps": mq(g(other, "ivfpq_100k_1024d_m128", "qps_gpu")),
                "cfgs_parity": None if not isinstance(other, dict) else all(v.get("ids_match", True) for v in other.values() if isinstance(v, dict) and "ids_match" in v),
                "cfg5": {"pca_TF": g(cfg5, "pca_8192_to_128", "tflops_f64"), "pca_frac": g(cfg5, "pca_8192_to_128", "roofline", "frac"),
                         "vlad_Mimg": mq(g(cfg5, "vlad_surf64_128_centroids", "images_per_s")), "vlad_frac": g(cfg5, "vlad_surf64_128_centroids", "roofline", "frac"),
                         "fused_Mimg": mq(g(cfg5, "fused_descriptors_to_128d", "images_per_s")), "e2e_self_hit": g(cfg5, "end_to_end", "self_hit_rate"),
                         "e2e_exact_self_hit": g(cfg5, "end_to_end", "exact_self_hit_rate")},
                "yfcc": {kk: {"Mqps": mq(g(yfcc, kk, "queries_per_s")), "passA_frac": g(yfcc, kk, "roofline", "frac"),
                              "passB_ms": g(yfcc, kk, "stage_ms_per_step", "pass_b"), "passB_mfma_frac": g(yfcc, kk, "pass_b", "roofline", "frac"),
                              "parity": ok(g(yfcc, kk, "parity"))} for kk in ("w2", "w64", "w64_between_clusters")} if isinstance(yfcc, dict) and "error" not in yfcc else None}
    if h is not None:
        chk(L.mmidx_destroy(h))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # side measurements -> bench_extra.json; text summary -> stderr; the ONE JSON line -> stdout, last
        importlib.import_module("bench_emit").emit(out, json_out=json_out, err=sys.stderr, extra_path=args.extra_out, summary=summ)


if __name__ == "__main__":
    main()
