"""Host-side mirror of the two steps that feed the index (BASELINE config 5), over the C ABI:

  gr.iti.mklab.visual.dimreduction.PCA                        (J/dimreduction/PCA.java)
  gr.iti.mklab.visual.aggregation.VladAggregator              (J/aggregation/VladAggregator.java)
  gr.iti.mklab.visual.aggregation.VladAggregatorMultipleVocabularies

Only the apply side is native (projection = one batched f64-MFMA GEMM, aggregation = one block per
image); learning the PCA basis / the codebooks stays offline, as in the reference (EJML SVD, Weka).
"""
import ctypes as C

import numpy as np

from . import _native as N
from ._native import MmidxError


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class PCA:
    """PCA(numComponents, numTrainingSamples, sampleSize, doWhitening), PCA.java:75-93."""

    def __init__(self, numComponents, numTrainingSamples, sampleSize, doWhitening, device=0):
        self.numComponents, self.sampleSize, self.doWhitening = numComponents, sampleSize, bool(doWhitening)
        self.numTrainingSamples = numTrainingSamples
        self.device = device
        self._h = None
        self.isPcaInitialized = False

    def loadPCAFromFile(self, filename):
        """PCA.java:257-318: line 1 means, line 2 eigenvalues, then one component per line (space separated)."""
        with open(filename) as f:
            means = np.array(f.readline().strip().split(" "), dtype=np.float64)
            if means.shape[0] != self.sampleSize:
                raise MmidxError(N.ERR_INVALID_ARG, "Means line is wrong!")
            eig_line = f.readline()
            eig = None
            if self.doWhitening:
                eig = np.array(eig_line.strip().split(" "), dtype=np.float64)
                if eig.shape[0] < self.numComponents:
                    raise MmidxError(N.ERR_INVALID_ARG, "Eigenvalues line is wrong!")
            Vt = np.zeros((self.numComponents, self.sampleSize))
            for i in range(self.numComponents):
                Vt[i] = np.array(f.readline().strip().split(" ")[: self.sampleSize], dtype=np.float64)
        self.load(means, eig, Vt)

    def load(self, means, eig, Vt):
        """Same as loadPCAFromFile with the parsed arrays (Vt = raw components, whitening folded natively)."""
        means, Vt = _f64(means), _f64(Vt).reshape(self.numComponents, self.sampleSize)
        eig = _f64(eig)[: self.numComponents].copy() if eig is not None else None
        h = C.c_void_p()
        N.check(N.lib().mmidx_pca_create(self.numComponents, self.sampleSize, int(self.doWhitening), means.ctypes.data,
                                         eig.ctypes.data if eig is not None else None, Vt.ctypes.data, self.device, C.byref(h)))
        self.close()
        self._h = h
        self.isPcaInitialized = True

    def project(self, X):
        """Batch form of sampleToEigenSpace: X [n][sampleSize] -> [n][numComponents]."""
        if not self.isPcaInitialized:
            raise MmidxError(N.ERR_NOT_READY, "PCA is not correctly initiallized!")  # sic, PCA.java:191
        X = _f64(X)
        if X.ndim != 2 or X.shape[1] != self.sampleSize:
            raise MmidxError(N.ERR_WRONG_DIM, "Unexpected vector length!")  # IllegalArgumentException, PCA.java:194
        Y = np.zeros((X.shape[0], self.numComponents))
        N.check(N.lib().mmidx_pca_project(self._h, X.shape[0], X.ctypes.data, Y.ctypes.data))
        return Y

    def sampleToEigenSpace(self, sampleData):
        """PCA.java:188-208 (does not modify its argument)."""
        x = _f64(sampleData)
        if x.ndim != 1 or x.shape[0] != self.sampleSize:
            raise MmidxError(N.ERR_WRONG_DIM, "Unexpected vector length!")
        return self.project(x.reshape(1, -1))[0]

    def close(self):
        if self._h:
            N.lib().mmidx_pca_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VladAggregatorMultipleVocabularies:
    """VladAggregatorMultipleVocabularies(double[][][] codebooks), normalizationsOn default true."""

    def __init__(self, codebooks, normalizationsOn=True, device=0):
        cbs = [_f64(cb) for cb in codebooks]
        self.descriptorLength = cbs[0].shape[1]
        self.numCentroids = [cb.shape[0] for cb in cbs]
        self.normalizationsOn = bool(normalizationsOn)
        ncent = np.array(self.numCentroids, np.int32)
        cat = np.ascontiguousarray(np.concatenate([cb.reshape(-1) for cb in cbs]))
        h = C.c_void_p()
        N.check(N.lib().mmidx_vlad_create(len(cbs), ncent.ctypes.data, self.descriptorLength, cat.ctypes.data,
                                          int(self.normalizationsOn), device, C.byref(h)))
        self._h = h
        self.vectorLength = int(sum(self.numCentroids)) * self.descriptorLength

    def getVectorLength(self):
        return self.vectorLength

    def aggregate_batch(self, descriptor_sets):
        """descriptor_sets: list of [n_i][descriptorLength] arrays (n_i may be 0) -> [nimg][vectorLength]."""
        nimg = len(descriptor_sets)
        off = np.zeros(nimg + 1, np.int64)
        for i, d in enumerate(descriptor_sets):
            off[i + 1] = off[i] + (0 if d is None else len(d))
        total = int(off[-1])
        descs = np.zeros((max(total, 1), self.descriptorLength))
        for i, d in enumerate(descriptor_sets):
            if off[i + 1] > off[i]:
                descs[off[i]:off[i + 1]] = _f64(d).reshape(-1, self.descriptorLength)
        out = np.zeros((nimg, self.vectorLength))
        N.check(N.lib().mmidx_vlad_aggregate(self._h, nimg, off.ctypes.data, descs.ctypes.data, out.ctypes.data))
        return out

    def set_option(self, name, value):
        """measurement / test switch: "exact" = 1 -> the one-kernel form with the fp64 brute-force assignment"""
        N.check(N.lib().mmidx_vlad_set_option(self._h, name.encode(), int(value)))

    def aggregate(self, descriptors):
        """VladAggregatorMultipleVocabularies.aggregate(double[][]), :84-101"""
        return self.aggregate_batch([descriptors])[0]

    def close(self):
        if getattr(self, "_h", None):
            N.lib().mmidx_vlad_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VladAggregator(VladAggregatorMultipleVocabularies):
    """VladAggregator(double[][] codebook): raw VLAD, no normalisation (VladAggregator.java:56-70)."""

    def __init__(self, codebook, device=0):
        super().__init__([codebook], normalizationsOn=False, device=device)


class ImageVectorizer:
    """Batch form of ImageVectorization.transformToVector (J/vectorization/ImageVectorization.java:169-208):
    descriptors -> VLAD -> PCA projection in one native call (`mmidx_vectorize`), replacing the reference's
    per-image thread pool (ImageVectorizer.java:123-126); the 8192-d VLAD vectors never leave the GPU."""

    def __init__(self, aggregator, pca):
        if aggregator.getVectorLength() != pca.sampleSize:
            raise MmidxError(N.ERR_WRONG_DIM, "aggregator vector length does not match the PCA sample size")
        self.aggregator, self.pca = aggregator, pca

    def transform_batch(self, descriptor_sets):
        """descriptor_sets: list of [n_i][descriptorLength] arrays -> [nimg][numComponents]"""
        ag = self.aggregator
        nimg = len(descriptor_sets)
        off = np.zeros(nimg + 1, np.int64)
        for i, d in enumerate(descriptor_sets):
            off[i + 1] = off[i] + (0 if d is None else len(d))
        total = int(off[-1])
        descs = np.zeros((max(total, 1), ag.descriptorLength))
        for i, d in enumerate(descriptor_sets):
            if off[i + 1] > off[i]:
                descs[off[i]:off[i + 1]] = _f64(d).reshape(-1, ag.descriptorLength)
        out = np.zeros((nimg, self.pca.numComponents))
        N.check(N.lib().mmidx_vectorize(ag._h, self.pca._h, nimg, off.ctypes.data, descs.ctypes.data, out.ctypes.data))
        return out

    def transformToVector(self, descriptors):
        return self.transform_batch([descriptors])[0]
