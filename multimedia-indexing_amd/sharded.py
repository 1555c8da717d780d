"""Multi-GPU search: one process per GPU, inverted lists partitioned across ranks.

Partitioning: whole inverted lists, list c lives on rank `c % world` (`owner_of_cell`).  Every
(query, probe) work item therefore runs on exactly one rank (its residual LUT is built once), and
the reference's offer order (probe rank, position in list) stays well defined.  Codebooks are
replicated (8.25 MiB).  Every rank OWNS a slice of the queries (`search_owned`: the serving form -- a rank is handed the
queries of its clients and answers them); per batch:

  0. all-gather of the query vectors                              (RCCL, nq*D*8 bytes: every rank needs every query for
                                                                   the lookup tables of the lists it holds)
  1. coarse top-w for the rank's own slice                        (mmidx_coarse_device)
  2. all-gather of the probe cells and their coarse distances     (RCCL, nq*w*12 bytes)
  3. pass A: scan of probe rank 0 where it is local -> thresholds (mmidx_shard_pass_a_device)
  4. MIN all-reduce of the thresholds                             (RCCL, nq*8 bytes)
  5. pass B: remaining local probes under the global thresholds   (mmidx_shard_pass_b_device)
  6. all-to-all to the query's owner: list lengths (fixed size), then ONE variable-size exchange of the valid entries,
     16 bytes each (distance, probe_rank << 32 | iid)            (RCCL)
  7. merge of the `world` lists per owned query                   (mmidx_merge_partials_device) -> flags straddling ties
  8. tie replay for the flagged queries: three passes over the local lists with three fixed-size all-reduces
     (mmidx_shard_tie_phase_device) -- the answer is the single queue's, ties included

The one host synchronisation of a batch (the split sizes of step 6) is hidden by running the batch as two sub-batches: the
second sub-batch's steps 0-5 are enqueued before the host waits for the first one's sizes.

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
    Returns (iid [nq][k], dist [nq][k], count [nq], flag [nq]).  Order: ascending distance; equal distances
    later-offered first, offer order = key ascending (probe_rank << 32 | iid).  flag = 1: the k-th and (k+1)-th merged
    distances are equal (the tie replay decides which of the equal candidates stay)."""
    S, nq, K1 = pdist.shape
    iid = np.full((nq, k), -1, np.int32)
    dist = np.full((nq, k), np.inf, np.float64)
    cnt = np.zeros(nq, np.int32)
    flag = np.zeros(nq, np.int32)
    for q in range(nq):
        d = np.concatenate([pdist[s, q, :min(int(pcount[s, q]), K1)] for s in range(S)])
        ky = np.concatenate([pkey[s, q, :min(int(pcount[s, q]), K1)] for s in range(S)])
        if d.size == 0:
            continue
        order = np.lexsort((ky, d))[:K1]
        d, ky = d[order], ky[order]
        flag[q] = 1 if (d.size > k and d[k - 1] == d[k]) else 0
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
    return iid, dist, cnt, flag


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
        """dense: pd_all / pk_all [S][nq][k+1]; ragged (poff [S][nq] int64 element offsets): flat arrays.
        Returns (iid, dist, count, flag): flag[q] = 1 when a tie straddles position k"""
        t = self.torch
        S, nq = pc_all.shape[0], pc_all.shape[1]
        iid = t.empty(nq, k, dtype=t.int32, device=pc_all.device)
        dist = t.empty(nq, k, dtype=t.float64, device=pc_all.device)
        cnt = t.empty(nq, dtype=t.int32, device=pc_all.device)
        flag = t.zeros(nq, dtype=t.int32, device=pc_all.device)
        if nq:
            N.check(self.L.mmidx_merge_partials_device(self.dev, k, nq, S, pd_all.data_ptr(), pk_all.data_ptr(),
                                                       pc_all.data_ptr(), poff.data_ptr() if poff is not None else None,
                                                       iid.data_ptr(), dist.data_ptr(), cnt.data_ptr(), flag.data_ptr(), self._stream()))
        return iid, dist, cnt, flag

    def tie_phase(self, phase, k, Q, cells, fq, tau, counts, pB, tie_iids):
        """one pass of the cross-shard tie replay over this rank's lists (in place on counts / pB / tie_iids)"""
        N.check(self.L.mmidx_shard_tie_phase_device(self.h, phase, k, fq.shape[0], Q.data_ptr(), cells.data_ptr(), fq.data_ptr(),
                                                    tau.data_ptr(), counts.data_ptr(), pB.data_ptr(), tie_iids.data_ptr(), self._stream()))


class HostStagedDist:
    """torch.distributed look-alike for FUNCTIONAL runs of the N > 1 path where RCCL cannot form the group -- two ranks on one
    GPU (RCCL refuses duplicate devices): the collectives of a gloo group, device tensors staged through host memory.
    Same call signatures as the torch.distributed functions ShardedIVFPQ and bench.py use; everything else is passed through.
    Not a performance path: bench.py selects it only under MMIDX_BENCH_ONE_GPU=1 (tests/test_gpu_two_ranks.py)."""

    def __init__(self, dist):
        self._d = dist
        self.ReduceOp = dist.ReduceOp

    def __getattr__(self, name):
        return getattr(self._d, name)

    def all_reduce(self, x, op=None, group=None):
        c = x.cpu()
        self._d.all_reduce(c, op=self._d.ReduceOp.SUM if op is None else op, group=group)
        x.copy_(c)

    def all_gather(self, parts, x, group=None):
        self._d.all_gather(parts, x, group=group)

    def all_gather_into_tensor(self, out, x, group=None):
        torch = __import__("torch")
        c = x.cpu().contiguous()
        parts = [torch.empty_like(c) for _ in range(self._d.get_world_size(group))]
        self._d.all_gather(parts, c, group=group)
        out.copy_(torch.stack(parts).reshape(out.shape))

    def all_to_all_single(self, out, x, output_split_sizes=None, input_split_sizes=None, group=None):
        torch = __import__("torch")
        co = torch.empty(out.shape, dtype=out.dtype)
        self._d.all_to_all_single(co, x.cpu().contiguous(), output_split_sizes, input_split_sizes, group=group)
        out.copy_(co)

    def broadcast(self, x, src=0, group=None):
        c = x.cpu()
        self._d.broadcast(c, src=src, group=group)
        x.copy_(c)


class ShardedIVFPQ:
    """computeNearestNeighbors over `world` shards (IVFPQ.computeKnnIVFADC, IVFPQ.java:408-450).

    Collectives per batch of B queries (K1 = k + 1, F = world * tie_slots flagged-query slots):
      all-gather  query vectors      B*D*8 bytes          (search_owned: every rank contributes its own slice)
      all-gather  probe cells        B*w*4 bytes  (+ their exact coarse distances, B*w*8: pass B's coarse bound)
      all-reduce  thresholds (MIN)   B*8 bytes            -- lets every shard prune with the global bound
      all-to-all  list lengths       B*4 bytes per rank, then ONE variable-size all-to-all of the valid entries, 16 bytes each
                                     (about B*(k+few)*16 bytes over all ranks) -- query q is merged on rank q // per
      tie replay  all-gather F*12 bytes, all-reduce F*w*8, F*4 and F*k*4 bytes (fixed sizes; the kernels exit at once for
                                     unused slots)
      all-gather  results            B*k*12 bytes         (only with gather=True)
    """

    def __init__(self, engine, rank, world, dist=None, group=None, force_collectives=False, max_batch=262144, tie_slots=32,
                 pipeline=None):
        self.engine, self.rank, self.world, self.dist, self.group = engine, rank, world, dist, group
        # queries per collective round: the shard phases take a bounded batch per call (pool memory); longer batches
        # are cut into rounds of max_batch queries (a multiple of world keeps the owner slices aligned)
        self.max_batch = max(world, max_batch - max_batch % world)
        # world == 1 normally short-circuits every collective; force_collectives issues them anyway
        # (a 1-rank process group) so that the RCCL calls can be exercised on a single-GPU box
        self.force = bool(force_collectives and dist is not None)
        self.tie_slots = int(tie_slots)   # flagged queries replayed per owner and round; the replay runs in as many rounds as the
        self.tie_overflow = 0             # busiest owner needs (tie_overflow stays 0: kept for callers that read it)
        # two sub-batches in flight hide the host's wait for the exchange sizes behind the second one's kernels; with one rank
        # there is nothing to hide and two half batches only cost their fixed overheads twice
        self.pipeline = (world > 1) if pipeline is None else bool(pipeline)
        self._pin = None

    # -- collectives --------------------------------------------------------------------------------------------------
    def _collective(self):
        return self.world > 1 or self.force

    def _all_gather(self, x):
        """stack of every rank's `x` along a new leading axis (same shape on all ranks)"""
        torch = __import__("torch")
        if not self._collective():
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
        if not self._collective():
            return x
        out = torch.empty_like(x)
        self.dist.all_to_all_single(out, x.contiguous(), group=self.group)
        return out

    def _all_reduce(self, x, op):
        if self._collective():
            self.dist.all_reduce(x, op=getattr(self.dist.ReduceOp, op), group=self.group)
        return x

    # -- public entry points --------------------------------------------------------------------------------------------
    def search(self, k, Q, gather=True):
        """Q: [nq][D] float64 tensor, identical on every rank (the query exchange is skipped).  Returns (iid, dist, count)
        for all queries on every rank, or with gather=False only this rank's slice (queries [rank*per, (rank+1)*per),
        per = ceil(nq / world)).  Batches longer than max_batch run as several collective rounds (with gather=False a
        rank then holds its slice of every round, concatenated)."""
        torch = __import__("torch")
        if Q.shape[0] > self.max_batch:
            parts = [self.search(k, Q[i:i + self.max_batch], gather) for i in range(0, Q.shape[0], self.max_batch)]
            return tuple(torch.cat([p[j] for p in parts], 0) for j in range(3))
        st = self._stage1(k, Q, None)
        return self._stage2(k, st, gather)

    def search_owned(self, k, Q_own):
        """The serving form: this rank hands in ITS queries (the same number on every rank) and gets their answers; the
        query vectors are exchanged inside the call (all-gather).  With pipeline=True the batch runs as two sub-batches so
        that the host's wait for the first one's exchange sizes overlaps the second one's kernels."""
        torch = __import__("torch")
        n = Q_own.shape[0]
        cap = max(1, self.max_batch // self.world)
        if n > cap:
            parts = [self.search_owned(k, Q_own[i:i + cap]) for i in range(0, n, cap)]
            return tuple(torch.cat([p[j] for p in parts], 0) for j in range(3))
        if not self.pipeline or n < 2:
            return self._stage2(k, self._stage1(k, None, Q_own), False)
        h = (n + 1) // 2
        s1 = self._stage1(k, None, Q_own[:h])
        s2 = self._stage1(k, None, Q_own[h:])
        r1 = self._stage2(k, s1, False)
        r2 = self._stage2(k, s2, False)
        return tuple(torch.cat([r1[j], r2[j]], 0) for j in range(3))

    # -- stage 1: everything up to the list lengths; the split sizes start their way to the host ------------------------
    def _stage1(self, k, Q, Q_own):
        torch = __import__("torch")
        W = self.world
        if Q is None:  # query exchange
            per = Q_own.shape[0]
            Q = self._all_gather(Q_own.contiguous()).reshape(W * per, -1).contiguous()
            nq, q0, q1 = W * per, self.rank * per, (self.rank + 1) * per
        else:
            nq = Q.shape[0]
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
        self._all_reduce(T, "MIN")
        pd, pk, pc = self.engine.pass_b(k, Q, cells, cdist, T)
        st = {"Q": Q, "cells": cells, "nq": nq, "per": per, "q0": q0, "q1": q1, "pd": pd, "pk": pk, "pc": pc}
        if not self._collective():
            return st
        # owner merge: pad the query axis to W*per, view as [W][per][...]
        K1 = k + 1
        if W * per > nq:
            padn = W * per - nq
            st["pd"] = pd = torch.cat([pd, torch.full((padn, K1), float("inf"), dtype=pd.dtype, device=pd.device)], 0)
            st["pk"] = pk = torch.cat([pk, torch.full((padn, K1), -1, dtype=pk.dtype, device=pk.device)], 0)
            st["pc"] = pc = torch.cat([pc, torch.zeros(padn, dtype=pc.dtype, device=pc.device)], 0)
        # Most (rank, query) lists are empty or short (a query's candidates live on the few ranks that own its
        # nearest cells), so only the valid entries travel: lengths first (fixed size), then a variable-size
        # all-to-all of the compacted lists; the merge kernel reads them ragged.
        pcw = pc.reshape(W, per)
        rc = self._all_to_all(pcw)                          # [W][per]: what every rank holds for my queries
        sizes = torch.stack([pcw.sum(1), rc.sum(1)])        # [2][W] send / receive split sizes
        st["rc"] = rc
        if sizes.is_cuda:  # asynchronous copy into pinned memory + an event: stage 2 waits for it, not for the whole stream
            pin = torch.empty(sizes.shape, dtype=sizes.dtype, pin_memory=True)
            pin.copy_(sizes, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            st["sizes"], st["ev"] = pin, ev
        else:
            st["sizes"], st["ev"] = sizes, None
        return st

    # -- stage 2: the variable-size exchange, the owner merge, the tie replay ------------------------------------------------
    def _stage2(self, k, st, gather):
        torch = __import__("torch")
        W, per, nq, q0, q1 = self.world, st["per"], st["nq"], st["q0"], st["q1"]
        pd, pk, pc = st["pd"], st["pk"], st["pc"]
        if not self._collective():  # one rank, no process group: it owns every query
            iid, dist_, cnt, flag = self.engine.merge(k, pd.unsqueeze(0), pk.unsqueeze(0), pc.unsqueeze(0))
            iid = self._tie_replay(k, st, iid, dist_, cnt, flag)
            return iid, dist_, cnt
        if st["ev"] is not None:
            st["ev"].synchronize()
        sizes = st["sizes"].tolist()
        send_sz, recv_sz = [int(x) for x in sizes[0]], [int(x) for x in sizes[1]]
        sd, sk = self.engine.compact(k, pd, pk, pc, sum(send_sz))    # (query, position) order = destination-major
        # one exchange of 16-byte entries: (distance bits, key) as two int64 columns
        send = torch.stack([sd.view(torch.int64), sk], 1).contiguous()
        recv = torch.empty((sum(recv_sz), 2), dtype=torch.int64, device=send.device)
        self.dist.all_to_all_single(recv, send, recv_sz, send_sz, group=self.group)
        rd = recv[:, 0].contiguous().view(torch.float64)
        rk = recv[:, 1].contiguous()
        rc = st["rc"]
        flat = rc.reshape(-1).to(torch.int64)
        poff = (torch.cumsum(flat, 0) - flat).reshape(W, per).contiguous()
        iid, dist_, cnt, flag = self.engine.merge(k, rd, rk, rc.contiguous(), poff)  # [per][k]
        iid = self._tie_replay(k, st, iid, dist_, cnt, flag)
        if not gather:
            return iid[:q1 - q0], dist_[:q1 - q0], cnt[:q1 - q0]
        iid = self._all_gather(iid).reshape(W * per, k)[:nq]
        dist_ = self._all_gather(dist_).reshape(W * per, k)[:nq]
        cnt = self._all_gather(cnt).reshape(W * per)[:nq]
        return iid, dist_, cnt

    def _tie_replay(self, k, st, iid, dist_, cnt, flag):
        """Queries whose k-th and (k+1)-th distances tie: the bounded queue's replay over all ranks' lists (fixed shapes,
        no host synchronisation; every rank takes part whether it owns flagged queries or not)."""
        torch = __import__("torch")
        if self.tie_slots <= 0:
            return iid
        # Straddling ties are rare (equal PQ codes of near-duplicate images).  Every rank has to take part in the replay's
        # collectives, so the ranks first agree on whether ANY owner flagged a query: one 4-byte MAX all-reduce and one host
        # read per round -- cheaper than always running three passes and four collectives over empty slots (measured on one
        # GPU: 0.03 against 0.41 ms per 16384-query step)
        nfl = flag.sum().reshape(1).to(torch.int32)
        self._all_reduce(nfl, "MAX")
        if int(nfl.item()) == 0:
            return iid
        W, per, q0 = self.world, st["per"], st["q0"]
        Fo = self.tie_slots
        n_own = iid.shape[0]
        dev = iid.device
        fl = flag[:n_own].to(torch.int64)
        pos = torch.cumsum(fl, 0) - fl                      # ordinal of every flagged row
        # every owner replays Fo flagged queries per round; the MAX all-reduce above gave all ranks the same number of rounds
        rounds = (int(nfl.item()) + Fo - 1) // Fo
        self.tie_rounds = getattr(self, "tie_rounds", 0) + rounds
        w = st["cells"].shape[1]
        for rnd in range(rounds):
            take = (fl > 0) & (pos >= rnd * Fo) & (pos < (rnd + 1) * Fo)
            # (no nonzero(): it would synchronise with the host) every taken row scatters its index to its slot, the others to a
            # scratch slot past the end
            rows_x = torch.full((Fo + 1,), -1, dtype=torch.int64, device=dev)
            rows_x.scatter_(0, torch.where(take, pos - rnd * Fo, torch.full_like(pos, Fo)), torch.arange(n_own, dtype=torch.int64, device=dev))
            rows = rows_x[:Fo]
            valid = rows >= 0
            safe = torch.where(valid, rows, torch.zeros_like(rows))
            fq_own = torch.where(valid, (safe + q0).to(torch.int32), torch.full((Fo,), -1, dtype=torch.int32, device=dev))
            tau_own = torch.where(valid, dist_[safe, k - 1], torch.zeros(Fo, dtype=dist_.dtype, device=dev))
            fq = self._all_gather(fq_own).reshape(-1).contiguous()          # [W * Fo]
            tau = self._all_gather(tau_own).reshape(-1).contiguous()
            F = fq.shape[0]
            counts = torch.zeros((F, w, 2), dtype=torch.int32, device=dev)
            pB = torch.zeros(F, dtype=torch.int32, device=dev)
            ties = torch.full((F, k), -1, dtype=torch.int32, device=dev)
            self.engine.tie_phase(0, k, st["Q"], st["cells"], fq, tau, counts, pB, ties)
            self._all_reduce(counts, "SUM")
            self.engine.tie_phase(1, k, st["Q"], st["cells"], fq, tau, counts, pB, ties)
            self._all_reduce(pB, "SUM")
            self.engine.tie_phase(2, k, st["Q"], st["cells"], fq, tau, counts, pB, ties)
            self._all_reduce(ties, "MAX")
            mine = ties[self.rank * Fo:(self.rank + 1) * Fo] if self._collective() or W > 1 else ties[:Fo]
            # kept ties overwrite their slots of the flagged rows; unused slots (row -1) are routed to a scratch row
            iid_x = torch.cat([iid, torch.zeros((1, k), dtype=iid.dtype, device=dev)], 0)
            dst = torch.where(valid, rows, torch.full_like(rows, iid.shape[0]))
            cur = iid_x[dst]
            iid_x[dst] = torch.where(mine >= 0, mine, cur)
            iid = iid_x[:iid.shape[0]]
        return iid
