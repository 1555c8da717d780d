#!/usr/bin/env python3
"""Per-rank cost of the W-GPU sharded search, measured on ONE GPU with W virtual shards (diagnostic, not a bench line).

Builds the cfg4 index once, hands every inverted list to shard cell % W (W handles on the same device), and runs the
steps of sharded.ShardedIVFPQ.search by hand for a batch of W x --batch queries: the coarse stage for one rank's slice,
then for EVERY shard pass A -> (MIN over shards) -> pass B -> compaction, then the ragged merge for one rank's slice.
Reported: mean / max time per shard of each phase = what one rank of a W-GPU run computes per step (communication and
host glue excluded).

  python tools/bench_virtual_shards.py --shards 8 [--n 100000000] [--batch 16384]
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--shards", type=int, default=8)
ap.add_argument("--n", type=int, default=100_000_000)
ap.add_argument("--batch", type=int, default=16384)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--chunk", type=int, default=2_000_000)
args = ap.parse_args()
nat = importlib.import_module("multimedia-indexing_amd._native")
sh = importlib.import_module("multimedia-indexing_amd.sharded")
torch.cuda.init()
L = nat.lib()
dev = torch.device("cuda", 0)
W, N, D, Cc, w, m, ks, k = args.shards, args.n, 128, 8192, 32, 16, 256, 100
B = args.batch * W
f64 = torch.float64
st = torch.cuda.current_stream().cuda_stream
g0 = torch.Generator(device=dev)
g0.manual_seed(1234)
mu = torch.randn(Cc, D, generator=g0, device=dev, dtype=f64)
coarse_h = mu.cpu().numpy()
pq_h = (0.15 * torch.randn(m, ks, D // m, generator=g0, device=dev, dtype=f64)).cpu().numpy()  # (codebook quality is irrelevant here)
hs = []
for r in range(W):
    h = C.c_void_p()
    nat.check(L.mmidx_create(nat.KIND_IVFPQ, D, m, ks, Cc, 0, None, None, 0, C.byref(h)))
    nat.check(L.mmidx_set_coarse(h, coarse_h.ctypes.data))
    nat.check(L.mmidx_set_pq(h, pq_h.ctypes.data))
    nat.check(L.mmidx_set_w(h, w))
    hs.append(h)
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
    cells = torch.empty(n, dtype=torch.int32, device=dev)
    codes = torch.empty(n, m, dtype=torch.int8, device=dev)
    nat.check(L.mmidx_encode_device(hs[0], n, X.data_ptr(), cells.data_ptr(), codes.data_ptr(), st))
    iids = torch.arange(n, device=dev, dtype=torch.int32) + c0
    for r in range(W):
        own = (cells % W) == r
        ii, oc, ok = iids[own].contiguous(), cells[own].contiguous(), codes[own].contiguous()
        torch.cuda.synchronize()
        nat.check(L.mmidx_add_codes_device(hs[r], ii.numel(), ii.data_ptr(), oc.data_ptr(), ok.data_ptr(), st))
    torch.cuda.synchronize()
    del X
for h in hs:
    nat.check(L.mmidx_sync_index(h))
Q += 0.01 * torch.randn(B, D, generator=gq, device=dev, dtype=f64)
engines = [sh.HipShardEngine(h, D, w, 0) for h in hs]
per = B // W


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    torch.cuda.synchronize()
    return out, e0.elapsed_time(e1)


acc = {}
for step in range(args.steps + 1):
    t = {}
    (cells, cdist), t["coarse_slice"] = timed(lambda: engines[0].coarse(Q[:per]))
    cells_all, cdist_all = engines[0].coarse(Q)  # (what the all-gather would deliver)
    Ts, ta = [], []
    for e in engines:
        T, ms = timed(lambda: e.pass_a(k, Q, cells_all))
        Ts.append(T)
        ta.append(ms)
    Tmin = torch.stack(Ts).min(0).values.contiguous()
    parts, tb, tc = [], [], []
    for e in engines:
        p, ms = timed(lambda: e.pass_b(k, Q, cells_all, cdist_all, Tmin))
        tb.append(ms)
        pd, pk, pc = p
        tot = int(pc[:per].sum())
        c, ms2 = timed(lambda: e.compact(k, pd, pk, pc, int(pc.sum())))
        tc.append(ms2)
        sd, sk = e.compact(k, pd[:per], pk[:per], pc[:per], tot)  # this shard's lists for rank 0's queries
        parts.append((sd, sk, pc[:per].contiguous()))
    rc = torch.stack([p[2] for p in parts])
    flat = rc.reshape(-1).to(torch.int64)
    poff = (torch.cumsum(flat, 0) - flat).reshape(W, per).contiguous()
    rd, rk = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
    _, t["merge_slice"] = timed(lambda: engines[0].merge(k, rd, rk, rc.contiguous(), poff))
    t["pass_a_mean"], t["pass_a_max"] = float(np.mean(ta)), float(np.max(ta))
    t["pass_b_mean"], t["pass_b_max"] = float(np.mean(tb)), float(np.max(tb))
    t["compact_mean"] = float(np.mean(tc))
    t["entries_sent_per_rank"] = float(np.mean([int(p[2].sum()) for p in parts])) * W
    if step:  # first step = warm-up
        for kk, v in t.items():
            acc.setdefault(kk, []).append(v)
out = {kk: round(float(np.mean(v)), 4) for kk, v in acc.items()}
out["per_rank_compute_ms"] = round(out["coarse_slice"] + out["pass_a_mean"] + out["pass_b_mean"] + out["compact_mean"] + out["merge_slice"], 4)
out["queries_per_step"] = B
out["shards"] = W
out["implied_qps_if_communication_were_free"] = round(B / out["per_rank_compute_ms"] * 1e3, 1)
print(json.dumps(out))
for h in hs:
    L.mmidx_destroy(h)
