"""Multi-GPU search: one process per GPU, inverted lists partitioned across ranks.

Partitioning: whole inverted lists, list c lives on rank `c % world` (`owner_of_cell`).  Every
(query, probe) work item therefore runs on exactly one rank (its residual LUT is built once), and
the reference's offer order (probe rank, position in list) stays well defined.  Codebooks are
replicated (8.25 MiB).  Per batch:

  1. coarse top-w for a 1/world slice of the queries            (mmidx_coarse_device)
  2. all-gather of the probe cells and their coarse distances    (RCCL, nq*w*12 bytes)
  3. pass A: scan of probe rank 0 where it is local -> thresholds  (mmidx_shard_pass_a_device)
  4. MIN all-reduce of the thresholds                             (RCCL, nq*8 bytes)
  5. pass B: remaining local probes under the global thresholds   (mmidx_shard_pass_b_device)
  6. all-to-all of the sorted partial lists to the query's owner  (RCCL: counts, then the valid entries only)
  7. merge of the `world` lists per owned query, all-gather results (mmidx_merge_partials_device)

The collectives go through torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the CPU
tests).  The per-rank engine is pluggable so that the orchestration (this file) is exercised on
CPU with world_size 2 by tests/test_sharded_gloo.py; on a GPU box the engine is `HipShardEngine`.
"""
import ctypes as C

import numpy as np

from . import _native as N


def owner_of_cell(cells, world):
    """rank that stores inverted list `cell`"""
    return cells % world


def merge_partials_host(k, pdist, pkey, pcount):
    """Host mirror of kernel K5 (k_merge_partials): numpy arrays [S][nq][k+1], [S][nq][k+1], [S][nq].
    Returns (iid [nq][k], dist [nq][k], count [nq]).  Order: ascending distance; equal distances
    later-offered first, offer order = key ascending (probe_rank << 32 | iid)."""
    S, nq, K1 = pdist.shape
    iid = np.full((nq, k), -1, np.int32)
    dist = np.full((nq, k), np.inf, np.float64)
    cnt = np.zeros(nq, np.int32)
    for q in range(nq):
        d = np.concatenate([pdist[s, q, :min(int(pcount[s, q]), K1)] for s in range(S)])
        ky = np.concatenate([pkey[s, q, :min(int(pcount[s, q]), K1)] for s in range(S)])
        if d.size == 0:
            continue
        order = np.lexsort((ky, d))[:K1]
        d, ky = d[order], ky[order]
        n = min(k, d.size)
        d, ky = d[:n], ky[:n]
        # reverse runs of equal distance
        out = np.arange(n)
        a = 0
        while a < n:
            b = a
            while b + 1 < n and d[b + 1] == d[a]:
                b += 1
            out[a:b + 1] = np.arange(b, a - 1, -1)
            a = b + 1
        iid[q, :n] = (ky[out] & 0xFFFFFFFF).astype(np.int64).astype(np.int32)
        dist[q, :n] = d[out]
        cnt[q] = n
    return iid, dist, cnt


class HipShardEngine:
    """Per-rank engine over the C ABI; all buffers are torch CUDA tensors (plumbing only)."""

    def __init__(self, handle, D, w, device_index):
        import torch

        self.torch = torch
        self.h, self.D, self.w, self.dev = handle, D, w, device_index
        self.L = N.lib()

    def _stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    def coarse(self, Qs):
        """probe cells [n][w] (nearest first) and the exact squared distance of each [n][w]"""
        t = self.torch
        cells = t.empty(Qs.shape[0], self.w, dtype=t.int32, device=Qs.device)
        cdist = t.empty(Qs.shape[0], self.w, dtype=t.float64, device=Qs.device)
        if Qs.shape[0]:
            N.check(self.L.mmidx_coarse_device(self.h, Qs.shape[0], Qs.data_ptr(), cells.data_ptr(), cdist.data_ptr(), self._stream()))
        return cells, cdist

    def pass_a(self, k, Q, cells):
        """scan probe rank 0 on this shard; returns the shard's thresholds [nq] float64 (+inf = none)"""
        t = self.torch
        T = t.empty(Q.shape[0], dtype=t.float64, device=Q.device)
        N.check(self.L.mmidx_shard_pass_a_device(self.h, k, Q.shape[0], Q.data_ptr(), cells.data_ptr(), T.data_ptr(),
                                                 self._stream()))
        return T

    def pass_b(self, k, Q, cells, cdist, T):
        """remaining probes with the cross-shard thresholds; returns sorted partial lists"""
        t = self.torch
        nq, K1 = Q.shape[0], k + 1
        pd = t.empty(nq, K1, dtype=t.float64, device=Q.device)
        pk = t.empty(nq, K1, dtype=t.int64, device=Q.device)
        pc = t.empty(nq, dtype=t.int32, device=Q.device)
        N.check(self.L.mmidx_shard_pass_b_device(self.h, k, nq, Q.data_ptr(), cells.data_ptr(),
                                                 cdist.data_ptr() if cdist is not None else None, T.data_ptr(), pd.data_ptr(),
                                                 pk.data_ptr(), pc.data_ptr(), self._stream()))
        return pd, pk, pc

    def search_partial(self, k, Q, cells):
        """single-call form (no threshold exchange)"""
        t = self.torch
        nq, K1 = Q.shape[0], k + 1
        pd = t.empty(nq, K1, dtype=t.float64, device=Q.device)
        pk = t.empty(nq, K1, dtype=t.int64, device=Q.device)
        pc = t.empty(nq, dtype=t.int32, device=Q.device)
        N.check(self.L.mmidx_search_partial_device(self.h, k, nq, Q.data_ptr(), cells.data_ptr(), pd.data_ptr(),
                                                   pk.data_ptr(), pc.data_ptr(), self._stream()))
        return pd, pk, pc

    def compact(self, k, pd, pk, pc, total):
        """the pc[q] valid entries of every list, concatenated in query order (total = their number)"""
        t = self.torch
        pc64 = pc.to(t.int64)
        poff = (t.cumsum(pc64, 0) - pc64).contiguous()
        od = t.empty(total, dtype=pd.dtype, device=pd.device)
        ok = t.empty(total, dtype=pk.dtype, device=pk.device)
        if pd.shape[0]:
            N.check(self.L.mmidx_compact_partials_device(self.dev, k, pd.shape[0], pd.data_ptr(), pk.data_ptr(), pc.data_ptr(),
                                                         poff.data_ptr(), od.data_ptr(), ok.data_ptr(), self._stream()))
        return od, ok

    def merge(self, k, pd_all, pk_all, pc_all, poff=None):
        """dense: pd_all / pk_all [S][nq][k+1]; ragged (poff [S][nq] int64 element offsets): flat arrays"""
        t = self.torch
        S, nq = pc_all.shape[0], pc_all.shape[1]
        iid = t.empty(nq, k, dtype=t.int32, device=pc_all.device)
        dist = t.empty(nq, k, dtype=t.float64, device=pc_all.device)
        cnt = t.empty(nq, dtype=t.int32, device=pc_all.device)
        if nq:
            N.check(self.L.mmidx_merge_partials_device(self.dev, k, nq, S, pd_all.data_ptr(), pk_all.data_ptr(),
                                                       pc_all.data_ptr(), poff.data_ptr() if poff is not None else None,
                                                       iid.data_ptr(), dist.data_ptr(), cnt.data_ptr(), self._stream()))
        return iid, dist, cnt


class ShardedIVFPQ:
    """computeNearestNeighbors over `world` shards (IVFPQ.computeKnnIVFADC, IVFPQ.java:408-450).

    Collectives per batch (B queries, K1 = k + 1):
      all-gather  probe cells        B*w*4 bytes  (+ their exact coarse distances, B*w*8: pass B's coarse bound)
      all-reduce  thresholds (MIN)   B*8 bytes            -- lets every shard prune with the global bound
      all-to-all  partial lists      counts (B*4 bytes/rank), then only the valid entries, 16 bytes each
                                     (about B*(k+few)*16 bytes over all ranks) -- query q is merged on rank q // per
      all-gather  results            B*k*12 bytes
    """

    def __init__(self, engine, rank, world, dist=None, group=None, force_collectives=False, max_batch=262144):
        self.engine, self.rank, self.world, self.dist, self.group = engine, rank, world, dist, group
        # queries per collective round: the shard phases take a bounded batch per call (pool memory); longer batches
        # are cut into rounds of max_batch queries (a multiple of world keeps the owner slices aligned)
        self.max_batch = max(world, max_batch - max_batch % world)
        # world == 1 normally short-circuits every collective; force_collectives issues them anyway
        # (a 1-rank process group) so that the RCCL calls can be exercised on a single-GPU box
        self.force = bool(force_collectives and dist is not None)

    def _all_gather(self, x):
        """stack of every rank's `x` along a new leading axis (same shape on all ranks)"""
        torch = __import__("torch")
        if self.world == 1 and not self.force:
            return x.unsqueeze(0)
        out = torch.empty((self.world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        if x.is_cuda:
            self.dist.all_gather_into_tensor(out, x.contiguous(), group=self.group)
        else:  # gloo: list form
            parts = [out[i] for i in range(self.world)]
            self.dist.all_gather(parts, x.contiguous(), group=self.group)
        return out

    def _all_to_all(self, x):
        """x [world][...]: slice r goes to rank r; returns [world][...] = what every rank sent to me"""
        torch = __import__("torch")
        if self.world == 1 and not self.force:
            return x
        out = torch.empty_like(x)
        self.dist.all_to_all_single(out, x.contiguous(), group=self.group)
        return out

    def search(self, k, Q, gather=True):
        """Q: [nq][D] float64 tensor, identical on every rank.  Returns (iid, dist, count) for all
        queries on every rank, or with gather=False only this rank's slice (queries
        [rank*per, (rank+1)*per), per = ceil(nq / world)) -- what a serving front-end needs when
        each rank answers the clients whose queries it owns.  Batches longer than max_batch run as several
        collective rounds (with gather=False a rank then holds its slice of every round, concatenated)."""
        torch = __import__("torch")
        if Q.shape[0] > self.max_batch:
            parts = [self.search(k, Q[i:i + self.max_batch], gather) for i in range(0, Q.shape[0], self.max_batch)]
            return tuple(torch.cat([p[j] for p in parts], 0) for j in range(3))
        nq, W = Q.shape[0], self.world
        per = (nq + W - 1) // W
        q0 = min(self.rank * per, nq)
        q1 = min(q0 + per, nq)
        cells_sl, cdist_sl = self.engine.coarse(Q[q0:q1])
        if q1 - q0 < per:  # pad the slice so that every rank contributes the same shape
            pad = torch.full((per - (q1 - q0), cells_sl.shape[1]), -1, dtype=cells_sl.dtype, device=cells_sl.device)
            cells_sl = torch.cat([cells_sl, pad], 0)
            cdist_sl = torch.cat([cdist_sl, torch.zeros(pad.shape, dtype=cdist_sl.dtype, device=cdist_sl.device)], 0)
        cells = self._all_gather(cells_sl).reshape(W * per, -1)[:nq].contiguous()
        cdist = self._all_gather(cdist_sl).reshape(W * per, -1)[:nq].contiguous()
        T = self.engine.pass_a(k, Q, cells)
        if W > 1 or self.force:
            self.dist.all_reduce(T, op=self.dist.ReduceOp.MIN, group=self.group)
        pd, pk, pc = self.engine.pass_b(k, Q, cells, cdist, T)
        if W == 1 and not self.force:  # (gather is moot: the one rank owns every query)
            return self.engine.merge(k, pd.unsqueeze(0), pk.unsqueeze(0), pc.unsqueeze(0))
        # owner merge: pad the query axis to W*per, view as [W][per][...], exchange, merge my slice
        K1 = k + 1
        if W * per > nq:
            padn = W * per - nq
            pd = torch.cat([pd, torch.full((padn, K1), float("inf"), dtype=pd.dtype, device=pd.device)], 0)
            pk = torch.cat([pk, torch.full((padn, K1), -1, dtype=pk.dtype, device=pk.device)], 0)
            pc = torch.cat([pc, torch.zeros(padn, dtype=pc.dtype, device=pc.device)], 0)
        # Most (rank, query) lists are empty or short (a query's candidates live on the few ranks that own its
        # nearest cells), so only the valid entries travel: counts first (fixed size), then a variable-size
        # all-to-all of the compacted lists; the merge kernel reads them ragged.
        pcw = pc.reshape(W, per)
        rc = self._all_to_all(pcw)                                   # [W][per]: what every rank holds for my queries
        sizes = torch.stack([pcw.sum(1), rc.sum(1)]).cpu().tolist()  # one host sync: the split sizes
        send_sz, recv_sz = [int(x) for x in sizes[0]], [int(x) for x in sizes[1]]
        sd, sk = self.engine.compact(k, pd, pk, pc, sum(send_sz))    # (query, position) order = destination-major
        rd = torch.empty(sum(recv_sz), dtype=pd.dtype, device=pd.device)
        rk = torch.empty(sum(recv_sz), dtype=pk.dtype, device=pk.device)
        if W == 1 and not self.force:
            rd, rk = sd, sk
        else:
            self.dist.all_to_all_single(rd, sd, recv_sz, send_sz, group=self.group)
            self.dist.all_to_all_single(rk, sk, recv_sz, send_sz, group=self.group)
        flat = rc.reshape(-1).to(torch.int64)
        poff = (torch.cumsum(flat, 0) - flat).reshape(W, per).contiguous()
        iid, dist_, cnt = self.engine.merge(k, rd, rk, rc.contiguous(), poff)  # [per][k]
        if not gather:
            return iid[:q1 - q0], dist_[:q1 - q0], cnt[:q1 - q0]
        iid = self._all_gather(iid).reshape(W * per, k)[:nq]
        dist_ = self._all_gather(dist_).reshape(W * per, k)[:nq]
        cnt = self._all_gather(cnt).reshape(W * per)[:nq]
        return iid, dist_, cnt
