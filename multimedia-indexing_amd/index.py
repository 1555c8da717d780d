"""Host-side mirror of the reference's index classes over the C ABI.

Same names, argument meaning and error behaviour as
  gr.iti.mklab.visual.datastructures.{AbstractSearchStructure, PQ, IVFPQ}
(J/ = src/main/java/gr/iti/mklab/visual/ in the reference tree), so that parity tests read like
tests of the reference.  The reference keeps id <-> iid maps and the persistent records in
BDB-JE; that half stays on the Java side of the JNI boundary and is represented
here by plain dicts.  All vector arithmetic happens in libmmidx_hip.so on the GPU.
"""
import ctypes as C
import time

import os

import numpy as np

from . import _native as N
from ._native import MmidxError


class TransformationType:
    """PQ.TransformationType, J/datastructures/PQ.java:78-80"""
    None_ = 0
    RandomRotation = 1
    RandomPermutation = 2


class Answer:
    """J/utilities/Answer.java:8-60 (timers are nanoseconds, ASS:285-287, :352-358)."""

    def __init__(self, ids, distances, nameLookupTime, indexSearchTime):
        self._ids, self._distances = ids, distances
        self._nameLookupTime, self._indexSearchTime = nameLookupTime, indexSearchTime

    def getIds(self):
        return self._ids

    def getDistances(self):
        return self._distances

    def getIndexSearchTime(self):
        return self._indexSearchTime

    def getNameLookupTime(self):
        return self._nameLookupTime


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def read_quantizer(filename, num_centroids, centroid_length):
    """AbstractFeatureAggregator.readQuantizer, J/aggregation/AbstractFeatureAggregator.java:234-254:
    one centroid per line, comma separated; lines without a comma are skipped as headers."""
    q = np.zeros((num_centroids, centroid_length), np.float64)
    counter = 0
    with open(filename) as f:
        for line in f:
            line = line.rstrip("\n")
            if "," not in line:
                continue
            vals = line.split(",")
            for i, v in enumerate(vals):
                q[counter, i] = float(v)
            counter += 1
    return q


class AbstractSearchStructure:
    """Template-method base, J/datastructures/AbstractSearchStructure.java."""

    def __init__(self, vectorLength, maxNumVectors, readOnly=False, countSizeOnLoad=True, loadCounter=0,
                 loadIndexInMemory=True, cacheSize=512):
        self.vectorLength = vectorLength
        self.maxNumVectors = maxNumVectors
        self.readOnly = readOnly
        self.loadCounter = loadCounter
        self.loadIndexInMemory = loadIndexInMemory
        self._id_to_iid = {}
        self._iid_to_id = {}
        self._h = None

    # -- id maps (BDB in the reference: ASS:383-419, :537-562) --
    def getInternalId(self, id_):
        return self._id_to_iid.get(id_, -1)

    def getId(self, iid):
        return self._iid_to_id.get(int(iid))

    def isIndexed(self, id_):
        return id_ in self._id_to_iid

    def getLoadCounter(self):
        return self.loadCounter

    def _create_mapping(self, id_):
        self._iid_to_id[self.loadCounter] = id_
        self._id_to_iid[id_] = self.loadCounter

    # -- ASS:229-257 --
    def indexVector(self, id_, vector):
        if self.loadCounter >= self.maxNumVectors:
            print("Maximum index capacity reached, no more vectors can be indexed!")
            return False
        if self.isIndexed(id_):
            print(f"Vector '{id_}' already indexed!")
            return False
        vector = np.asarray(vector, dtype=np.float64)
        # dimension check first so that a failed call leaves no mapping behind
        if vector.ndim != 1 or vector.shape[0] != self.vectorLength:
            raise MmidxError(N.ERR_WRONG_DIM, "The dimensionality of the vector is wrong!")
        self._create_mapping(id_)
        try:
            self.indexVectorInternal(vector)
        except Exception:
            # a failed native call must not leave the id marked as indexed (the reference's createMapping precedes
            # indexVectorInternal too, ASS:244-248, but its only failure there is the dimension check done above)
            self._iid_to_id.pop(self.loadCounter, None)
            self._id_to_iid.pop(id_, None)
            raise
        self.loadCounter += 1
        return True

    def indexVectors(self, ids, vectors):
        """Batch overload (one launch for the whole batch); same checks per id."""
        vectors = _f64(vectors)
        if vectors.ndim != 2 or vectors.shape[1] != self.vectorLength:
            raise MmidxError(N.ERR_WRONG_DIM, "The dimensionality of the vector is wrong!")
        keep, seen = [], set()
        for i, id_ in enumerate(ids):
            if self.loadCounter + len(keep) >= self.maxNumVectors:
                print("Maximum index capacity reached, no more vectors can be indexed!")
                break
            if self.isIndexed(id_) or id_ in seen:
                print(f"Vector '{id_}' already indexed!")
                continue
            keep.append(i)
            seen.add(id_)
        if not keep:
            return 0
        iids = np.arange(self.loadCounter, self.loadCounter + len(keep), dtype=np.int32)
        self._add_vectors(vectors[keep], iids)
        for i in keep:
            self._create_mapping(ids[i])
            self.loadCounter += 1
        return len(keep)

    # -- ASS:281-291, :320-328, :345-373 --
    def computeNearestNeighbors(self, k, query):
        if isinstance(query, str):
            iid = self.getInternalId(query)
            start = time.perf_counter_ns()
            res = self.computeNearestNeighborsInternalById(k, iid)
            return self._look_up(res, time.perf_counter_ns() - start)
        if not self.loadIndexInMemory:
            raise MmidxError(N.ERR_NOT_IN_MEMORY, "Cannot execute query because the index is not loaded in memory!")
        start = time.perf_counter_ns()
        res = self.computeNearestNeighborsInternal(k, query)
        return self._look_up(res, time.perf_counter_ns() - start)

    def _look_up(self, res, search_ns):
        iids, dists = res
        start = time.perf_counter_ns()
        ids = [self.getId(i) for i in iids]
        return Answer(ids, np.array(dists, dtype=np.float64), time.perf_counter_ns() - start, search_ns)

    def close(self):
        if self._h:
            N.lib().mmidx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _PQBase(AbstractSearchStructure):
    _kind = None

    def _create(self, numSubVectors, numProductCentroids, transformation, numCoarseCentroids, device, perm, rot):
        self.numSubVectors = numSubVectors
        self.numProductCentroids = numProductCentroids
        self.transformation = transformation
        self.numCoarseCentroids = numCoarseCentroids
        L = N.lib()
        h = C.c_void_p()
        pp = rp = None
        if perm is not None:
            self._perm = np.ascontiguousarray(perm, np.int32)
            pp = self._perm.ctypes.data
        if rot is not None:
            self._rot = _f64(rot)
            rp = self._rot.ctypes.data
        devices = getattr(self, "_devices", None)
        if devices is not None:
            # one index over several GPUs from this one process (mmidx_create_sharded; the Java form is -Dmmidx.devices=0,1,...)
            N.preload_rccl()
            self._devs = (C.c_int * len(devices))(*[int(d) for d in devices])
            N.check(L.mmidx_create_sharded(self._kind, self.vectorLength, numSubVectors, numProductCentroids, numCoarseCentroids,
                                           transformation, pp, rp, len(devices), self._devs, C.byref(h)))
        else:
            N.check(L.mmidx_create(self._kind, self.vectorLength, numSubVectors, numProductCentroids, numCoarseCentroids,
                                   transformation, pp, rp, device, C.byref(h)))
        self._h = h
        self.subVectorLength = self.vectorLength // numSubVectors
        self._code_dtype = np.int8 if numProductCentroids <= 256 else np.int16

    # ---- quantizers ----
    def loadProductQuantizer(self, filename_or_array):
        """IVFPQ.java:275-288 / PQ.java:210-223: m*ks lines of dsub comma-separated doubles."""
        m, ks, ds = self.numSubVectors, self.numProductCentroids, self.subVectorLength
        if isinstance(filename_or_array, str):
            pq = np.zeros((m, ks, ds), np.float64)
            with open(filename_or_array) as f:
                for i in range(m):
                    for j in range(ks):
                        parts = f.readline().split(",")
                        for t in range(ds):
                            pq[i, j, t] = float(parts[t])
        else:
            pq = _f64(filename_or_array, (m, ks, ds))
        N.check(N.lib().mmidx_set_pq(self._h, pq.ctypes.data))

    # ---- native calls ----
    def _add_vectors(self, X, iids):
        X = _f64(X)
        iids = np.ascontiguousarray(iids, np.int32)
        N.check(N.lib().mmidx_add_vectors(self._h, X.shape[0], X.ctypes.data, iids.ctypes.data, None, None))

    def indexVectorInternal(self, vector):
        if len(vector) != self.vectorLength:
            raise MmidxError(N.ERR_WRONG_DIM, "The dimensionality of the vector is wrong!")
        self._add_vectors(_f64(vector).reshape(1, -1), np.array([self.loadCounter], np.int32))

    def encode(self, X):
        """cells [n] (-1 for PQ), codes [n][m] in the stored form (int8 = idx-128 / int16)."""
        X = _f64(X)
        if X.ndim != 2 or X.shape[1] != self.vectorLength:
            raise MmidxError(N.ERR_WRONG_DIM, "The dimensionality of the vector is wrong!")
        cells = np.zeros(X.shape[0], np.int32)
        codes = np.zeros((X.shape[0], self.numSubVectors), self._code_dtype)
        N.check(N.lib().mmidx_encode(self._h, X.shape[0], X.ctypes.data, cells.ctypes.data, codes.ctypes.data))
        return cells, codes

    def search_batch(self, k, Q):
        """Batch overload of computeNearestNeighborsInternal: (iids [nq][k], dists [nq][k], counts [nq])."""
        Q = _f64(Q)
        if Q.ndim != 2 or Q.shape[1] != self.vectorLength:
            raise MmidxError(N.ERR_WRONG_DIM, "The dimensionality of the vector is wrong!")
        nq = Q.shape[0]
        iids = np.full((nq, max(k, 1)), -1, np.int32)
        dists = np.full((nq, max(k, 1)), np.inf, np.float64)
        counts = np.zeros(nq, np.int32)
        N.check(N.lib().mmidx_search(self._h, k, nq, Q.ctypes.data, iids.ctypes.data, dists.ctypes.data, counts.ctypes.data))
        return iids, dists, counts

    def computeNearestNeighborsInternal(self, k, query):
        iids, dists, counts = self.search_batch(k, _f64(query).reshape(1, -1))
        n = int(counts[0])
        return iids[0, :n].copy(), dists[0, :n].copy()

    def search_sdc_batch(self, k, iids):
        """Batch form of computeNearestNeighborsInternal(k, int iid): PQ.computeKnnSDC, PQ.java:334-374."""
        iids = np.ascontiguousarray(iids, np.int32)
        nq = iids.shape[0]
        out_i = np.full((nq, max(k, 1)), -1, np.int32)
        out_d = np.full((nq, max(k, 1)), np.inf, np.float64)
        cnt = np.zeros(nq, np.int32)
        N.check(N.lib().mmidx_search_sdc(self._h, k, nq, iids.ctypes.data, out_i.ctypes.data, out_d.ctypes.data, cnt.ctypes.data))
        return out_i, out_d, cnt

    def computeNearestNeighborsInternalById(self, k, iid):
        # PQ: computeKnnSDC (PQ.java:334-374).  IVFPQ: computeKnnIVFSDC returns null in the reference
        # (IVFPQ.java:509-511) -> the native call reports UNSUPPORTED.
        if iid < 0:
            raise MmidxError(N.ERR_INVALID_ARG, "Id does not exist!")
        i, d, c = self.search_sdc_batch(k, np.array([iid], np.int32))
        n = int(c[0])
        return i[0, :n].copy(), d[0, :n].copy()

    def export(self):
        """List-major snapshot (list_off [nlists+1], iids [n], codes [n][m] stored form)."""
        nl = self.numCoarseCentroids if self._kind == N.KIND_IVFPQ else 1
        off = np.zeros(nl + 1, np.int64)
        N.check(N.lib().mmidx_export(self._h, off.ctypes.data, None, None))
        n = int(off[-1])
        iids = np.zeros(n, np.int32)
        codes = np.zeros((n, self.numSubVectors), self._code_dtype)
        N.check(N.lib().mmidx_export(self._h, off.ctypes.data, iids.ctypes.data, codes.ctypes.data))
        return off, iids, codes

    def saveSnapshot(self, filename):
        """Flat snapshot of the in-memory index for fast restart: the library's own file (mmidx_save: the
        list-major arrays loadIndexInMemory builds, IVFPQ.java:680-728) plus, next to it, the id map this mirror keeps where the
        Java side keeps its BDB environment (`filename`.ids.npz)."""
        N.check(N.lib().mmidx_save(self._h, os.fsencode(filename)))
        iids = np.array(sorted(self._iid_to_id), np.int32)
        # ids as a fixed-width unicode array: loadable without pickle (an untrusted snapshot must not be able to run code)
        ids = np.array([str(self._iid_to_id[int(i)]) for i in iids], dtype=str)
        np.savez(str(filename) + ".ids.npz", iids=iids, ids=ids, load_counter=np.int64(self.loadCounter))

    def loadSnapshot(self, filename):
        """Inverse of saveSnapshot on an empty index of the same shape (quantizers loaded separately, as after the reference's
        constructor): mmidx_load for the records, the sidecar for the id map."""
        N.check(N.lib().mmidx_load(self._h, os.fsencode(filename)))
        z = np.load(str(filename) + ".ids.npz", allow_pickle=False)
        for i, name in zip(z["iids"], z["ids"]):
            self._iid_to_id[int(i)] = str(name)
            self._id_to_iid[str(name)] = int(i)
        self.loadCounter = int(z["load_counter"])

    # -- per-id utilities (IVFPQ.java:464-497, :801-880; the reference reads the BDB record of the id) --
    def _record(self, id_):
        iid = self.getInternalId(id_)
        if iid == -1:
            raise MmidxError(N.ERR_INVALID_ARG, "Id does not exist!")
        iids = np.array([iid], np.int32)
        cell = np.zeros(1, np.int32)
        code = np.zeros((1, self.numSubVectors), self._code_dtype)
        N.check(N.lib().mmidx_get_codes(self._h, 1, iids.ctypes.data, cell.ctypes.data, code.ctypes.data))
        return int(cell[0]), code[0]

    def getPQCodeByte(self, id_):
        """IVFPQ.java:801-824: the stored byte code (index - 128) of the vector with the given id"""
        if self.getInternalId(id_) == -1:
            raise MmidxError(N.ERR_INVALID_ARG, "Id does not exist!")
        if self.numProductCentroids > 256:
            raise MmidxError(N.ERR_INVALID_ARG, "Call the short variant of the method!")
        return self._record(id_)[1]

    def getPQCodeShort(self, id_):
        """IVFPQ.java:833-856"""
        if self.getInternalId(id_) == -1:
            raise MmidxError(N.ERR_INVALID_ARG, "Id does not exist!")
        if self.numProductCentroids <= 256:
            raise MmidxError(N.ERR_INVALID_ARG, "Call the short variant of the method!")  # (sic: the reference's message)
        return self._record(id_)[1]

    def computeDistance(self, qVector, existingVecId):
        """distance between a query vector and the code of an indexed vector: IVFPQ.computeDistanceIVFADC, IVFPQ.java:464-497
        (for PQ: the same sum with the query in place of the residual)"""
        iid = self.getInternalId(existingVecId)
        if iid == -1:
            raise MmidxError(N.ERR_INVALID_ARG, "Id does not exist!")
        q = _f64(qVector, (1, self.vectorLength))
        iids = np.array([iid], np.int32)
        out = np.zeros(1, np.float64)
        N.check(N.lib().mmidx_distance(self._h, 1, q.ctypes.data, iids.ctypes.data, out.ctypes.data))
        return float(out[0])

    def size(self):
        n = C.c_int64()
        N.check(N.lib().mmidx_size(self._h, C.byref(n)))
        return n.value

    def set_profiling(self, on=True):
        N.check(N.lib().mmidx_set_profiling(self._h, int(on)))

    def set_option(self, name, value):
        """measurement / A-B switches of the native library (mmidx_set_option)"""
        N.check(N.lib().mmidx_set_option(self._h, name.encode(), int(value)))

    def get_stats(self):
        s = N.Stats()
        N.check(N.lib().mmidx_get_stats(self._h, C.byref(s)))
        return {f: getattr(s, f) for f, _ in N.Stats._fields_}

    def get_dispatch(self):
        """which kernel family served each stage of the most recent search sub-batch: {"coarse": .., "pass_a": .., "pre": .., "pass_b": ..}"""
        buf = C.create_string_buffer(512)
        N.check(N.lib().mmidx_get_dispatch(self._h, buf, 512))
        return dict(kv.split("=", 1) for kv in buf.value.decode().split(";"))

    @staticmethod
    def transformToByte(code):
        """PQ.transformToByte, PQ.java:552-558"""
        return (np.asarray(code, np.int32) - 128).astype(np.int8)

    @staticmethod
    def transformToShort(code):
        """PQ.transformToShort, PQ.java:544-550"""
        return np.asarray(code, np.int32).astype(np.int16)


class PQ(_PQBase):
    """J/datastructures/PQ.java (exhaustive ADC search)."""
    _kind = N.KIND_PQ

    def __init__(self, vectorLength, maxNumVectors, readOnly, BDBEnvHome, numSubVectors, numProductCentroids,
                 transformation, *rest, device=0, perm=None, rot=None):
        # PQ.java:142-144 (11 args) and :197-198 (8 args: ..., cacheSize)
        if len(rest) == 1:
            countSizeOnLoad, loadCounter, loadIndexInMemory, cacheSize = True, 0, True, rest[0]
        elif len(rest) == 4:
            countSizeOnLoad, loadCounter, loadIndexInMemory, cacheSize = rest
        else:
            raise TypeError("PQ(vectorLength, maxNumVectors, readOnly, BDBEnvHome, numSubVectors, "
                            "numProductCentroids, transformation, [countSizeOnLoad, loadCounter, "
                            "loadIndexInMemory,] cacheSize)")
        super().__init__(vectorLength, maxNumVectors, readOnly, countSizeOnLoad, loadCounter, loadIndexInMemory, cacheSize)
        self.BDBEnvHome = BDBEnvHome
        self._create(numSubVectors, numProductCentroids, transformation, 0, device, perm, rot)

    def loadIndex(self, codes):
        """loadIndexInMemory PQ.java:436-483: codes [n][m] in the stored form, iid = position."""
        codes = np.ascontiguousarray(codes, self._code_dtype)
        n = codes.shape[0]
        iids = np.arange(self.loadCounter, self.loadCounter + n, dtype=np.int32)
        N.check(N.lib().mmidx_add_codes(self._h, n, iids.ctypes.data, None, codes.ctypes.data))
        for i in iids:
            self._create_mapping(str(int(i)))
            self.loadCounter += 1


class IVFPQ(_PQBase):
    """J/datastructures/IVFPQ.java (IVFADC)."""
    _kind = N.KIND_IVFPQ

    def __init__(self, vectorLength, maxNumVectors, readOnly, BDBEnvHome, numSubVectors, numProductCentroids,
                 transformation, numCoarseCentroids, *rest, device=0, perm=None, rot=None, devices=None):
        # IVFPQ.java:174-177 (12 args) and :261-263 (9 args: ..., cacheSize); devices = [0, 1, ...]: the inverted lists are
        # partitioned over these GPUs (a device listed twice = virtual shards: tests on a one-GPU box)
        self._devices = list(devices) if devices is not None else None
        if len(rest) == 1:
            countSizeOnLoad, loadCounter, loadIndexInMemory, cacheSize = True, 0, True, rest[0]
        elif len(rest) == 4:
            countSizeOnLoad, loadCounter, loadIndexInMemory, cacheSize = rest
        else:
            raise TypeError("IVFPQ(vectorLength, maxNumVectors, readOnly, BDBEnvHome, numSubVectors, "
                            "numProductCentroids, transformation, numCoarseCentroids, [countSizeOnLoad, "
                            "loadCounter, loadIndexInMemory,] cacheSize)")
        super().__init__(vectorLength, maxNumVectors, readOnly, countSizeOnLoad, loadCounter, loadIndexInMemory, cacheSize)
        self.BDBEnvHome = BDBEnvHome
        self._create(numSubVectors, numProductCentroids, transformation, numCoarseCentroids, device, perm, rot)

    def setW(self, w):
        """IVFPQ.java:95-97"""
        N.check(N.lib().mmidx_set_w(self._h, int(w)))

    def getW(self):
        w = C.c_int()
        N.check(N.lib().mmidx_get_w(self._h, C.byref(w)))
        return w.value

    def loadCoarseQuantizer(self, filename_or_array):
        """IVFPQ.java:297-300"""
        if isinstance(filename_or_array, str):
            cq = read_quantizer(filename_or_array, self.numCoarseCentroids, self.vectorLength)
        else:
            cq = _f64(filename_or_array, (self.numCoarseCentroids, self.vectorLength))
        N.check(N.lib().mmidx_set_coarse(self._h, cq.ctypes.data))

    def indexPQCode(self, id_, listId, code):
        """IVFPQ.java:357-386: append a pre-computed (id, listId, byte[] code)."""
        if self.numProductCentroids > 256:
            raise MmidxError(N.ERR_BYTE_OVERFLOW,
                             "Byte is not sufficient to enumerate the centroids of the product quantizer!")
        if self.loadCounter >= self.maxNumVectors:
            print("Maximum index capacity reached, no more vectors can be indexed!")
            return False
        if self.isIndexed(id_):
            print(f"Vector '{id_}' already indexed!")
            return False
        code = np.ascontiguousarray(code, np.int8).reshape(1, -1)
        iid = np.array([self.loadCounter], np.int32)
        cell = np.array([listId], np.int32)
        N.check(N.lib().mmidx_add_codes(self._h, 1, iid.ctypes.data, cell.ctypes.data, code.ctypes.data))
        self._create_mapping(id_)
        self.loadCounter += 1
        return True

    def loadIndex(self, iids, listIds, codes):
        """loadIndexInMemory IVFPQ.java:680-728: stream (iid, listId, code) records."""
        iids = np.ascontiguousarray(iids, np.int32)
        listIds = np.ascontiguousarray(listIds, np.int32)
        codes = np.ascontiguousarray(codes, self._code_dtype)
        N.check(N.lib().mmidx_add_codes(self._h, len(iids), iids.ctypes.data, listIds.ctypes.data, codes.ctypes.data))
        for i in iids:
            self._iid_to_id[int(i)] = str(int(i))
            self._id_to_iid[str(int(i))] = int(i)
        self.loadCounter += len(iids)

    def getInvertedListId(self, id_):
        """IVFPQ.java:865-880: the inverted list (coarse cell) of the vector with the given id"""
        return self._record(id_)[0]

    def computeDistanceIVFADC(self, qVector, existingVecId):
        """IVFPQ.java:464-497"""
        return self.computeDistance(qVector, existingVecId)

    def listSizes(self):
        out = np.zeros(self.numCoarseCentroids, np.int32)
        N.check(N.lib().mmidx_list_sizes(self._h, out.ctypes.data))
        return out

    def outputItemsPerList(self):
        """IVFPQ.java:654-673"""
        s = self.listSizes()
        print(f"Maximum number of vectors: {int(s.max())}")
        print(f"Minimum number of vectors: {int(s.min())}")
        print(f"Average number of vectors: {float(s.sum()) / self.numCoarseCentroids}")


class Linear(AbstractSearchStructure):
    """Linear(vectorLength, maxNumVectors, readOnly, BDBEnvHome, ...), J/datastructures/Linear.java:67-100: exhaustive
    exact search (BASELINE config 1).  indexVector / computeNearestNeighbors as everywhere; the arithmetic is the
    reference's (sequential fp64 squared distance per vector, bounded queue in index order, Linear.java:138-163)."""

    def __init__(self, vectorLength, maxNumVectors, readOnly=False, BDBEnvHome="", loadIndexInMemory=True, countSizeOnLoad=True,
                 loadCounter=0, device=0):
        super().__init__(vectorLength, maxNumVectors, readOnly, countSizeOnLoad, loadCounter, loadIndexInMemory)
        h = C.c_void_p()
        N.check(N.lib().mmidx_linear_create(vectorLength, maxNumVectors, device, C.byref(h)))
        self._h = h

    def _add_vectors(self, X, iids):
        X = _f64(X)
        N.check(N.lib().mmidx_linear_add(self._h, X.shape[0], X.ctypes.data))

    def indexVectorInternal(self, vector):
        if len(vector) != self.vectorLength:
            raise MmidxError(N.ERR_WRONG_DIM, "The dimensionality of the vector is wrong!")
        self._add_vectors(_f64(vector).reshape(1, -1), None)

    def search_batch(self, k, Q):
        Q = _f64(Q)
        if Q.ndim != 2 or Q.shape[1] != self.vectorLength:
            raise MmidxError(N.ERR_WRONG_DIM, "The dimensionality of the vector is wrong!")
        nq = Q.shape[0]
        iids = np.full((nq, max(k, 1)), -1, np.int32)
        dists = np.full((nq, max(k, 1)), np.inf, np.float64)
        counts = np.zeros(nq, np.int32)
        N.check(N.lib().mmidx_linear_search(self._h, k, nq, Q.ctypes.data, iids.ctypes.data, dists.ctypes.data, counts.ctypes.data))
        return iids, dists, counts

    def computeNearestNeighborsInternal(self, k, query):
        iids, dists, counts = self.search_batch(k, _f64(query).reshape(1, -1))
        n = int(counts[0])
        return iids[0, :n].copy(), dists[0, :n].copy()

    def computeNearestNeighborsInternalById(self, k, iid):
        # Linear.java:181-186: the stored vector of iid is the query
        return self.computeNearestNeighborsInternal(k, self.getVector(iid))

    def getVector(self, iid):
        """Linear.getVector, Linear.java:253-263"""
        out = np.zeros(self.vectorLength, np.float64)
        N.check(N.lib().mmidx_linear_get_vector(self._h, int(iid), out.ctypes.data))
        return out

    def size(self):
        n = C.c_int64(0)
        N.check(N.lib().mmidx_linear_size(self._h, C.byref(n)))
        return n.value

    def close(self):
        if self._h:
            N.lib().mmidx_linear_destroy(self._h)
            self._h = None
