"""Debug aid: sweep 2's bitmap (MMIDX_K3MA_DUMP) against the set of (row, code) pairs whose exact distance is <= T."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth
mi = importlib.import_module("multimedia-indexing_amd")
from oracle import oracle as o
D, m, C, n, w, k, ks, nq = 128, 16, 6, 30000, 3, 100, 256, 90
rng = np.random.default_rng(7 * D + m + k)
mu = 0.5 * rng.standard_normal((C, D))
base = mu[rng.integers(0, C, n)] + rng.standard_normal((n, D))
ds = D // m
pq = np.stack([synth.kmeans((mu[rng.integers(0, C, 3000)] - base[:3000])[:, s * ds:(s + 1) * ds], 256, iters=2, seed=s) for s in range(m)])
ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512)
ix.loadCoarseQuantizer(mu); ix.loadProductQuantizer(pq); ix.setW(w)
ix.indexVectors([str(i) for i in range(n)], base)
Q = base[200:200 + nq] + 0.01 * rng.standard_normal((nq, D))
ix.set_option("passa_mfma", 1)
ix.set_option("mfma_sub", 8192)
os.environ["MMIDX_K3MA_DUMP"] = "/tmp/k3ma.bin"
got = ix.search_batch(k, Q)
raw = open("/tmp/k3ma.bin", "rb").read()
hdr = np.frombuffer(raw[:64], np.int64)
np1, ng, nsub, sub, stride, nq_, w_ = [int(x) for x in hdr[:7]]
off = 64
order = np.frombuffer(raw[off:off + 4 * np1], np.int32); off += 4 * np1
gd = np.frombuffer(raw[off:off + 16 * ng], np.int32).reshape(ng, 4); off += 16 * ng
T = np.frombuffer(raw[off:off + 8 * nq_], np.float64); off += 8 * nq_
bm = np.frombuffer(raw[off:], np.uint8)
print("pairs", np1, "groups", ng, "nsub", nsub, "sub", sub, "stride", stride)
# exact distances of every code of every cell to a query: oracle, w = 1, k = whole list
ref = o.OracleIndex(o.KIND_IVFPQ, D, m, ks, C)
ref.set_coarse(mu); ref.set_pq(pq); ref.set_w(1); ref.add_vectors(base)
d2c = ((base[:, None, :] - mu[None, :, :]) ** 2).sum(-1).argmin(1)
pos_in_list = np.zeros(n, np.int64)
for c in range(C):
    idx = np.where(d2c == c)[0]; pos_in_list[idx] = np.arange(len(idx))
rid, rd, rc = ref.search_batch(Q, 600)
tot_ok = tot_fp = tot_miss = 0
for gi in range(ng):
    cell, first, npg, _ = gd[gi]
    ntl = (npg + 15) >> 4
    for isub in range(nsub):
        item = bm[(gi * nsub + isub) * stride:(gi * nsub + isub + 1) * stride]
        words = item.view(np.uint16) if ntl <= 2 else item.view(np.uint32)
        got_set = set()
        for x in np.nonzero(words)[0]:
            wb = int(words[x]); ln = x & 63; pr = x >> 6; nn = ln & 15; g = ln >> 4
            for h in range(2):
                bits = (wb >> (8 * h)) & 0xFF if ntl <= 2 else (wb >> (16 * h)) & 0xFFFF
                for b in range(16):
                    if bits >> b & 1:
                        got_set.add(((b >> 2) * 16 + 4 * g + (b & 3), isub * sub + (2 * pr + h) * 16 + nn))
        exp_set = set()
        for row in range(npg):
            q = order[first + row] // w_
            for iid, dd in zip(rid[q, :rc[q]], rd[q, :rc[q]]):
                if dd <= T[q] and isub * sub <= pos_in_list[iid] < (isub + 1) * sub:
                    exp_set.add((row, int(pos_in_list[iid])))
        ok = len(exp_set & got_set); fp = len(got_set - exp_set); miss = len(exp_set - got_set)
        tot_ok += ok; tot_fp += fp; tot_miss += miss
        if gi == 0 and isub == 0:
            ms = sorted(exp_set - got_set)[:12]; fs = sorted(got_set - exp_set)[:12]
            print("group 0: np", npg, "ntl", ntl, "ok", ok, "false", fp, "missing", miss)
            print("  missing e.g.", ms)
            print("  false   e.g.", fs)
            rows_m = np.bincount([r for r, _ in exp_set - got_set], minlength=npg); rows_f = np.bincount([r for r, _ in got_set - exp_set], minlength=64)
            print("  missing by row", rows_m); print("  false by row  ", rows_f[:npg + 4])
print("total ok", tot_ok, "false positives", tot_fp, "missing", tot_miss)
# alternative decodings of group 0: which permutation of the four accumulator registers explains the bits?
import itertools
gi = 0; cell, first, npg, _ = gd[gi]; ntl = (npg + 15) >> 4
item = bm[:stride]; words = item.view(np.uint16) if ntl <= 2 else item.view(np.uint32)
exp_set = set()
for row in range(npg):
    q = order[first + row] // w_
    for iid, dd in zip(rid[q, :rc[q]], rd[q, :rc[q]]):
        if dd <= T[q]:
            exp_set.add((row, int(pos_in_list[iid])))
raw_bits = []
for x in np.nonzero(words)[0]:
    wb = int(words[x]); ln = x & 63; pr = x >> 6; nn = ln & 15; g = ln >> 4
    for h in range(2):
        bits = (wb >> (8 * h)) & 0xFF
        for b in range(8):
            if bits >> b & 1:
                raw_bits.append((b, g, (2 * pr + h) * 16 + nn))
print("bit index histogram:", np.bincount([b for b, _, _ in raw_bits], minlength=8))
for perm in itertools.permutations(range(4)):
    gs = set((4 * g + perm[b & 3], p) for b, g, p in raw_bits if b < 4)
    print(perm, "ok", len(gs & exp_set), "false", len(gs - exp_set), "missing", len(exp_set - gs))
