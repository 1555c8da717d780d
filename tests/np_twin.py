"""Independent numpy twin of the CPU oracle (test infrastructure).

Written in a deliberately different code shape from oracle/mmidx_oracle.c (vectorised over
candidates, closed-form bounded-queue result instead of a replayed queue) so that a slip in one
restatement shows up as a disagreement.  Arithmetic stays in the reference's order: numpy's
elementwise `a*a` then `+=` is one rounding per operation, no FMA, and the per-element
accumulation runs over the same index sequence (t ascending, s ascending).

Citations: J/ = /root/reference/src/main/java/gr/iti/mklab/visual/
"""
import numpy as np


# ---- LingPipe BoundedPriorityQueue, closed form (assumption A1) --------------------------------
def bpq_result(dists, k):
    """Positions (arrival indices) kept by a max-size-k queue after offering dists[0..n) in
    order, returned best -> worst (ties: later-inserted first).

    Closed form of the replay in mmidx_oracle.c: with tau the k-th smallest distance,
    'better' = d < tau, 'ties' = d == tau.  Until k non-junk (d <= tau) items have arrived every
    one of them is accepted; afterwards ties are rejected and every better item evicts the
    earliest-inserted tie still held.  So the kept ties are x_{e+1..p}: p = ties among the first
    k non-junk arrivals, e = better items arriving after that moment.
    """
    d = np.asarray(dists, dtype=np.float64)
    n = d.shape[0]
    if n <= k:
        keep = np.arange(n)
    else:
        tau = np.partition(d, k - 1)[k - 1]
        better = np.nonzero(d < tau)[0]
        ties = np.nonzero(d == tau)[0]
        nonjunk = np.nonzero(d <= tau)[0]
        s_m = nonjunk[k - 1]  # arrival index of the k-th non-junk item
        p = int(np.count_nonzero(ties <= s_m))
        e = int(np.count_nonzero(better > s_m))
        keep = np.concatenate([better, ties[e:p]])
    # best -> worst: distance ascending, later arrival first among equals
    order = np.lexsort((-keep, d[keep]))
    return keep[order]


def bpq_replay(dists, k):
    """Literal python replay (small inputs only)."""
    q = []  # (dist, -ins, pos)
    for pos, dist in enumerate(dists):
        if len(q) < k:
            q.append((dist, -pos, pos))
        else:
            q.sort()
            if not (dist < q[-1][0]):
                continue
            q.pop()  # worst; earliest inserted among equal-worst sorts last
            q.append((dist, -pos, pos))
        q.sort()
    q.sort()
    return np.array([e[2] for e in q], dtype=np.int64)


# ---- java.util.Random / Collections.shuffle  (RandomPermutation.java:29-40) -------------------
class JRandom:
    def __init__(self, seed):
        self.s = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1)

    def next(self, bits):
        self.s = (self.s * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        v = self.s >> (48 - bits)
        if v >= 1 << 31:
            v -= 1 << 32
        return v

    def next_int(self, bound=None):
        if bound is None:
            return self.next(32)
        if bound & (-bound) == bound:
            return (bound * self.next(31)) >> 31
        while True:
            bits = self.next(31)
            val = bits % bound
            t = (bits - val + (bound - 1)) & 0xFFFFFFFF
            if t < 0x80000000:
                return val


def random_permutation(seed, dim):
    r = JRandom(seed)
    lst = list(range(dim))
    for i in range(dim, 1, -1):
        j = r.next_int(i)
        lst[i - 1], lst[j] = lst[j], lst[i - 1]
    return np.array(lst, dtype=np.int32)


# ---- engines ------------------------------------------------------------------------------
def sq_dists_rows(A, v):
    """sum_j (A[i,j]-v[j])^2 accumulated j ascending (Linear.java:147-149 uses q-x, the coarse
    loops c-v; (a-b)^2 == (b-a)^2 exactly in IEEE arithmetic)."""
    acc = np.zeros(A.shape[0])
    for j in range(A.shape[1]):
        diff = A[:, j] - v[j]
        acc += diff * diff
    return acc


def linear_search(X, q, k):
    d = sq_dists_rows(X, q)
    pos = bpq_result(d, k)
    return pos.astype(np.int32), d[pos]


def transform(vec, tr, perm=None, rot=None):
    if tr == 2:
        return vec[perm]
    if tr == 1:
        out = np.zeros_like(vec)
        for i in range(vec.shape[0]):  # sequential over the inner index (A2)
            out += vec[i] * rot[i, :]
        return out
    return vec


def lookup_adc(pq, qv):
    m, ks, dsub = pq.shape
    lut = np.zeros((m, ks))
    sub = qv.reshape(m, dsub)
    for t in range(dsub):
        diff = sub[:, t][:, None] - pq[:, :, t]
        lut += diff * diff
    return lut


def adc_dists(lut, codes):
    """codes: [n, m] centroid indices"""
    d = np.zeros(codes.shape[0])
    for s in range(lut.shape[0]):
        d += lut[s, codes[:, s]]
    return d


def encode(v, pq, coarse=None, tr=0, perm=None, rot=None):
    cell = -1
    if coarse is not None:
        cell = int(np.argmin(sq_dists_rows(coarse, v)))  # first minimum wins
        vec = coarse[cell] - v  # IVFPQ.java:645
    else:
        vec = v
    vec = transform(vec, tr, perm, rot)
    lut = lookup_adc(pq, vec)  # same (a-b)^2 sums as computeNearestProductIndex
    return cell, np.argmin(lut, axis=1).astype(np.int32)


def pq_search(pq, codes, q, k, tr=0, perm=None, rot=None):
    lut = lookup_adc(pq, transform(q, tr, perm, rot))
    d = adc_dists(lut, codes)
    pos = bpq_result(d, k)
    return pos.astype(np.int32), d[pos]


def ivfpq_search(coarse, pq, lists, q, k, w, tr=0, perm=None, rot=None):
    """lists: list of (iids int array, codes [len, m]) per cell"""
    cd = sq_dists_rows(coarse, q)
    cells = bpq_result(cd, w)  # nearest first
    all_d, all_id = [], []
    for c in cells:
        iids, codes = lists[c]
        if len(iids) == 0:
            continue
        lut = lookup_adc(pq, transform(coarse[c] - q, tr, perm, rot))
        all_d.append(adc_dists(lut, codes))
        all_id.append(np.asarray(iids))
    if not all_d:
        return np.zeros(0, np.int32), np.zeros(0)
    d = np.concatenate(all_d)
    ids = np.concatenate(all_id)
    pos = bpq_result(d, k)
    return ids[pos].astype(np.int32), d[pos]


def normalize_l2(v):
    v = np.array(v, dtype=np.float64)
    acc = 0.0
    for x in v:
        acc += x * x
    n = np.sqrt(acc)
    return np.ones_like(v) if n == 0 else v / n


def pca_project(Vt, means, x, whitening):
    xc = x - means
    y = np.zeros(Vt.shape[0])
    for j in range(Vt.shape[1]):  # sequential over the inner index (A2)
        y += Vt[:, j] * xc[j]
    return normalize_l2(y) if whitening else y


def vlad(codebook, descs):
    nc, dl = codebook.shape
    out = np.zeros((nc, dl))
    for d in descs:
        c = int(np.argmin(sq_dists_rows(codebook, d)))
        out[c] += d - codebook[c]
    return out.reshape(-1)
