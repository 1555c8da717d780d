"""ctypes binding of the CPU ORACLE (test infrastructure, NOT product code).

PARITY STATUS: "parity unpinned" -- see oracle/mmidx_oracle.h.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmmidx_oracle.so")

KIND_PQ, KIND_IVFPQ = 1, 2
TR_NONE, TR_ROTATION, TR_PERMUTATION = 0, 1, 2


def build(force=False):
    src = os.path.join(_HERE, "mmidx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    vp = C.c_void_p
    sig = {
        "mmo_bpq_new": (vp, [C.c_int]),
        "mmo_bpq_free": (None, [vp]),
        "mmo_bpq_offer": (C.c_int, [vp, C.c_int, C.c_double]),
        "mmo_set_queue_rule": (None, [C.c_int]),
        "mmo_get_queue_rule": (C.c_int, []),
        "mmo_bpq_size": (C.c_int, [vp]),
        "mmo_bpq_last_dist": (C.c_double, [vp]),
        "mmo_bpq_poll": (C.c_int, [vp, ip, dp]),
        "mmo_bpq_to_arrays": (C.c_int, [vp, ip, dp]),
        "mmo_jdk_first_next_int": (C.c_int32, [C.c_int64]),
        "mmo_random_permutation": (None, [C.c_int64, C.c_int, vp]),
        "mmo_permute": (None, [vp, C.c_int, dp, dp]),
        "mmo_rotate": (None, [dp, C.c_int, dp, dp]),
        "mmo_normalize_l2": (None, [dp, C.c_int]),
        "mmo_normalize_l1": (None, [dp, C.c_int]),
        "mmo_normalize_power": (None, [dp, C.c_int, C.c_double]),
        "mmo_normalize_ssr": (None, [dp, C.c_int]),
        "mmo_linear_search": (C.c_int, [dp, C.c_int, C.c_int, dp, C.c_int, ip, dp]),
        "mmo_index_new": (vp, [C.c_int] * 6 + [vp, vp]),
        "mmo_index_free": (None, [vp]),
        "mmo_index_set_coarse": (None, [vp, dp]),
        "mmo_index_set_pq": (None, [vp, dp]),
        "mmo_index_set_w": (None, [vp, C.c_int]),
        "mmo_index_get_w": (C.c_int, [vp]),
        "mmo_index_size": (C.c_int, [vp]),
        "mmo_index_encode": (None, [vp, dp, ip, ip]),
        "mmo_index_add_vector": (C.c_int, [vp, dp]),
        "mmo_index_add_code": (C.c_int, [vp, C.c_int, C.c_int, ip]),
        "mmo_transform_to_byte": (C.c_int8, [C.c_int]),
        "mmo_index_load_lists": (C.c_int, [vp, C.c_int, vp, vp, vp]),
        "mmo_index_search": (C.c_int, [vp, C.c_int, dp, ip, dp]),
        "mmo_index_nearest_coarse": (None, [vp, dp, C.c_int, ip]),
        "mmo_index_lookup_adc": (None, [vp, dp, dp]),
        "mmo_pq_search_sdc": (C.c_int, [vp, C.c_int, C.c_int, ip, dp]),
        "mmo_index_get_record": (C.c_int, [vp, C.c_int, ip, ip]),
        "mmo_index_distance": (C.c_int, [vp, dp, C.c_int, dp]),
        "mmo_index_search_batch": (None, [vp, C.c_int, C.c_int, dp, ip, dp, ip, C.c_int]),
        "mmo_linear_search_batch": (None, [dp, C.c_int, C.c_int, C.c_int, C.c_int, dp, ip, dp, ip, C.c_int]),
        "mmo_index_probed_codes": (C.c_longlong, [vp, dp]),
        "mmo_index_list_sizes": (None, [vp, ip]),
        "mmo_pca_project": (None, [dp, dp, C.c_int, C.c_int, C.c_int, dp, dp]),
        "mmo_pca_whiten": (None, [dp, dp, C.c_int, C.c_int]),
        "mmo_nearest_centroid": (C.c_int, [dp, C.c_int, C.c_int, dp]),
        "mmo_vlad_aggregate": (None, [dp, C.c_int, C.c_int, dp, C.c_int, dp]),
        "mmo_vlad_aggregate_multi": (None, [dp, ip, C.c_int, C.c_int, dp, C.c_int, C.c_int, dp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int))


class BPQ:
    """LingPipe BoundedPriorityQueue<Result> emulation (assumption A1)."""

    def __init__(self, max_size):
        self._h = lib().mmo_bpq_new(max_size)
        if not self._h:
            raise ValueError("max size must be >= 1")

    def __del__(self):
        try:
            if getattr(self, "_h", None) and _lib is not None:
                _lib.mmo_bpq_free(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    def offer(self, id_, dist):
        return bool(lib().mmo_bpq_offer(self._h, int(id_), float(dist)))

    def __len__(self):
        return lib().mmo_bpq_size(self._h)

    def last(self):
        return lib().mmo_bpq_last_dist(self._h)

    def poll(self):
        i, d = C.c_int(), C.c_double()
        if not lib().mmo_bpq_poll(self._h, C.byref(i), C.byref(d)):
            return None
        return i.value, d.value

    def to_arrays(self):
        n = len(self)
        ids = np.zeros(n, np.int32)
        ds = np.zeros(n, np.float64)
        lib().mmo_bpq_to_arrays(self._h, ids.ctypes.data_as(C.POINTER(C.c_int)),
                                ds.ctypes.data_as(C.POINTER(C.c_double)))
        return ids, ds


QUEUE_RULE_A1, QUEUE_RULE_ACCEPT_EQUAL, QUEUE_RULE_EARLIER_FIRST = 0, 1, 2


def set_queue_rule(rule):
    """process-wide: which of the three plausible LingPipe BoundedPriorityQueue behaviours every bounded queue of the oracle follows
    (0 = assumption A1, the default; 1 = accept-equal-to-worst; 2 = earlier-inserted-first among equals) -- for the tests that show
    which answers do not depend on the assumption"""
    lib().mmo_set_queue_rule(int(rule))


def get_queue_rule():
    return int(lib().mmo_get_queue_rule())


def jdk_first_next_int(seed):
    return lib().mmo_jdk_first_next_int(seed)


def random_permutation(seed, dim):
    out = np.zeros(dim, np.int32)
    lib().mmo_random_permutation(seed, dim, out.ctypes.data)
    return out


def rotate(R, v):
    R, Rp = _d(R)
    v, vp = _d(v)
    out = np.zeros_like(v)
    lib().mmo_rotate(Rp, v.shape[0], vp, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def normalize(v, kind, a=0.5):
    v = np.array(v, dtype=np.float64, copy=True)
    p = v.ctypes.data_as(C.POINTER(C.c_double))
    L = lib()
    if kind == "l2":
        L.mmo_normalize_l2(p, v.size)
    elif kind == "l1":
        L.mmo_normalize_l1(p, v.size)
    elif kind == "power":
        L.mmo_normalize_power(p, v.size, a)
    elif kind == "ssr":
        L.mmo_normalize_ssr(p, v.size)
    else:
        raise ValueError(kind)
    return v


def linear_search(X, q, k):
    X, Xp = _d(X)
    q, qp = _d(q)
    ids = np.zeros(k, np.int32)
    ds = np.zeros(k, np.float64)
    n = lib().mmo_linear_search(Xp, X.shape[0], X.shape[1], qp, k,
                                ids.ctypes.data_as(C.POINTER(C.c_int)),
                                ds.ctypes.data_as(C.POINTER(C.c_double)))
    return ids[:n], ds[:n]


def linear_search_batch(X, Q, k, nthreads=1):
    X, Xp = _d(X)
    Q, Qp = _d(Q)
    nq = Q.shape[0]
    ids = np.full((nq, k), -1, np.int32)
    ds = np.full((nq, k), np.inf, np.float64)
    cn = np.zeros(nq, np.int32)
    lib().mmo_linear_search_batch(Xp, X.shape[0], X.shape[1], k, nq, Qp,
                                  ids.ctypes.data_as(C.POINTER(C.c_int)),
                                  ds.ctypes.data_as(C.POINTER(C.c_double)),
                                  cn.ctypes.data_as(C.POINTER(C.c_int)), nthreads)
    return ids, ds, cn


class OracleIndex:
    """PQ / IVFPQ restatement. kind: KIND_PQ | KIND_IVFPQ."""

    def __init__(self, kind, D, m, ks, C_=0, transform=TR_NONE, perm=None, rot=None):
        self.kind, self.D, self.m, self.ks, self.C = kind, D, m, ks, C_
        pp = rp = None
        self._keep = []
        if perm is not None:
            perm = np.ascontiguousarray(perm, np.int32)
            self._keep.append(perm)
            pp = perm.ctypes.data
        if rot is not None:
            rot = np.ascontiguousarray(rot, np.float64)
            self._keep.append(rot)
            rp = rot.ctypes.data
        self._h = lib().mmo_index_new(kind, D, m, ks, C_, transform, pp, rp)
        if not self._h:
            raise ValueError("The given number of subvectors is not valid!")

    def __del__(self):
        try:
            if getattr(self, "_h", None) and _lib is not None:
                _lib.mmo_index_free(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    def set_coarse(self, coarse):
        a, p = _d(coarse)
        assert a.shape == (self.C, self.D)
        lib().mmo_index_set_coarse(self._h, p)

    def set_pq(self, pq):
        a, p = _d(pq)
        assert a.size == self.D * self.ks
        lib().mmo_index_set_pq(self._h, p)

    def set_w(self, w):
        lib().mmo_index_set_w(self._h, w)

    @property
    def w(self):
        return lib().mmo_index_get_w(self._h)

    def __len__(self):
        return lib().mmo_index_size(self._h)

    def encode(self, v):
        v, vp = _d(v)
        cell = C.c_int()
        code = np.zeros(self.m, np.int32)
        lib().mmo_index_encode(self._h, vp, C.byref(cell), code.ctypes.data_as(C.POINTER(C.c_int)))
        return cell.value, code

    def encode_batch(self, X):
        X = np.ascontiguousarray(X, np.float64)
        cells = np.zeros(X.shape[0], np.int32)
        codes = np.zeros((X.shape[0], self.m), np.int32)
        for i in range(X.shape[0]):
            cells[i], codes[i] = self.encode(X[i])
        return cells, codes

    def add_vector(self, v):
        v, vp = _d(v)
        return lib().mmo_index_add_vector(self._h, vp)

    def add_vectors(self, X):
        X = np.ascontiguousarray(X, np.float64)
        for i in range(X.shape[0]):
            self.add_vector(X[i])

    def add_code(self, iid, cell, code):
        code, cp = _i(code)
        return lib().mmo_index_add_code(self._h, int(iid), int(cell), cp)

    def add_codes(self, iids, cells, codes):
        codes = np.ascontiguousarray(codes, np.int32)
        for i in range(len(iids)):
            self.add_code(iids[i], cells[i] if cells is not None else -1, codes[i])

    def load_lists(self, off, iids, codes_stored):
        """Bulk loadIndexInMemory: list-major arrays, codes in the stored form."""
        off = np.ascontiguousarray(off, np.int64)
        iids = np.ascontiguousarray(iids, np.int32)
        dt = np.int8 if self.ks <= 256 else np.int16
        codes = np.ascontiguousarray(codes_stored, dt)
        rc = lib().mmo_index_load_lists(self._h, len(off) - 1, off.ctypes.data, iids.ctypes.data,
                                        codes.ctypes.data)
        if rc:
            raise ValueError("list count mismatch")

    def search(self, q, k):
        q, qp = _d(q)
        ids = np.zeros(k, np.int32)
        ds = np.zeros(k, np.float64)
        n = lib().mmo_index_search(self._h, k, qp, ids.ctypes.data_as(C.POINTER(C.c_int)),
                                   ds.ctypes.data_as(C.POINTER(C.c_double)))
        if n < 0:
            raise ValueError("invalid k / w")
        return ids[:n], ds[:n]

    def search_batch(self, Q, k, nthreads=1):
        Q, Qp = _d(Q)
        nq = Q.shape[0]
        ids = np.full((nq, k), -1, np.int32)
        ds = np.full((nq, k), np.inf, np.float64)
        cn = np.zeros(nq, np.int32)
        lib().mmo_index_search_batch(self._h, k, nq, Qp, ids.ctypes.data_as(C.POINTER(C.c_int)),
                                     ds.ctypes.data_as(C.POINTER(C.c_double)),
                                     cn.ctypes.data_as(C.POINTER(C.c_int)), nthreads)
        return ids, ds, cn

    def nearest_coarse(self, q, w):
        q, qp = _d(q)
        out = np.zeros(w, np.int32)
        lib().mmo_index_nearest_coarse(self._h, qp, w, out.ctypes.data_as(C.POINTER(C.c_int)))
        return out

    def lookup_adc(self, qv):
        qv, qp = _d(qv)
        lut = np.zeros((self.m, self.ks), np.float64)
        lib().mmo_index_lookup_adc(self._h, qp, lut.ctypes.data_as(C.POINTER(C.c_double)))
        return lut

    def get_record(self, iid):
        """(cell, stored code) of an internal id, or None when it does not exist (IVFPQ.java:801-880)"""
        cell = C.c_int(0)
        code = np.zeros(self.m, np.int32)
        ok = lib().mmo_index_get_record(self._h, int(iid), C.byref(cell), code.ctypes.data_as(C.POINTER(C.c_int)))
        return (cell.value, code) if ok else None

    def distance(self, q, iid):
        """computeDistanceIVFADC (IVFPQ.java:464-497); None when the id does not exist"""
        q, qp = _d(q)
        out = C.c_double(0.0)
        ok = lib().mmo_index_distance(self._h, qp, int(iid), C.byref(out))
        return out.value if ok else None

    def search_sdc(self, iid, k):
        ids = np.zeros(k, np.int32)
        ds = np.zeros(k, np.float64)
        n = lib().mmo_pq_search_sdc(self._h, k, iid, ids.ctypes.data_as(C.POINTER(C.c_int)),
                                    ds.ctypes.data_as(C.POINTER(C.c_double)))
        if n < 0:
            raise ValueError("sdc not available")
        return ids[:n], ds[:n]

    def probed_codes(self, q):
        q, qp = _d(q)
        return lib().mmo_index_probed_codes(self._h, qp)

    def list_sizes(self):
        n = self.C if self.kind == KIND_IVFPQ else 1
        out = np.zeros(n, np.int32)
        lib().mmo_index_list_sizes(self._h, out.ctypes.data_as(C.POINTER(C.c_int)))
        return out


def pca_whiten(Vt, eig):
    Vt = np.array(Vt, dtype=np.float64, copy=True)
    eig, ep = _d(eig)
    lib().mmo_pca_whiten(Vt.ctypes.data_as(C.POINTER(C.c_double)), ep, Vt.shape[0], Vt.shape[1])
    return Vt


def pca_project(Vt, means, x, whitening):
    Vt, Vp = _d(Vt)
    means, mp = _d(means)
    x, xp = _d(x)
    y = np.zeros(Vt.shape[0], np.float64)
    lib().mmo_pca_project(Vp, mp, Vt.shape[0], Vt.shape[1], int(bool(whitening)), xp,
                          y.ctypes.data_as(C.POINTER(C.c_double)))
    return y


def nearest_centroid(codebook, desc):
    cb, cp = _d(codebook)
    d, dp = _d(desc)
    return lib().mmo_nearest_centroid(cp, cb.shape[0], cb.shape[1], dp)


def vlad_aggregate(codebook, descs):
    cb, cp = _d(codebook)
    descs = np.ascontiguousarray(descs, np.float64).reshape(-1, cb.shape[1])
    out = np.zeros(cb.size, np.float64)
    lib().mmo_vlad_aggregate(cp, cb.shape[0], cb.shape[1],
                             descs.ctypes.data_as(C.POINTER(C.c_double)), descs.shape[0],
                             out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def vlad_aggregate_multi(codebooks, descs, normalizations_on=True):
    dl = codebooks[0].shape[1]
    nc = np.array([cb.shape[0] for cb in codebooks], np.int32)
    cat = np.ascontiguousarray(np.concatenate([np.asarray(cb, np.float64) for cb in codebooks]))
    descs = np.ascontiguousarray(descs, np.float64).reshape(-1, dl)
    out = np.zeros(int(nc.sum()) * dl, np.float64)
    lib().mmo_vlad_aggregate_multi(cat.ctypes.data_as(C.POINTER(C.c_double)),
                                   nc.ctypes.data_as(C.POINTER(C.c_int)), len(codebooks), dl,
                                   descs.ctypes.data_as(C.POINTER(C.c_double)), descs.shape[0],
                                   int(bool(normalizations_on)),
                                   out.ctypes.data_as(C.POINTER(C.c_double)))
    return out
