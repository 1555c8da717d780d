"""Codebook learning on the GPU -- host-side mirror of gr.iti.mklab.visual.quantization.

    AbstractQuantizerLearning.learnAndWriteQuantizer   AbstractQuantizerLearning.java:39-81
    CoarseQuantizerLearning (main)                     CoarseQuantizerLearning.java:39-72
    ProductQuantizationLearning (main)                 ProductQuantizationLearning.java:247-305
    ResidualVectorComputation.ComputeResidualVector    ResidualVectorComputation.java:27-37

The reference clusters with Weka's SimpleKMeans (absent third-party dependency); here the clustering is
`mmidx_kmeans` (csrc/mmidx_learn.hip): the same algorithm restated -- seeding by java.util.Random(seed), optional
k-means++, min-max attribute normalisation (Weka's default distance), empty clusters dropped -- not Weka's exact
stream or summation order (parity unpinned for this row).  Files are written in the format the loaders read
(AbstractFeatureAggregator.readQuantizer, AFA:234-254: one centroid per line, comma separated; the product
quantizer file is the m sub-quantizers one after the other, IVFPQ.java:275-288; clusters that came out empty
are replaced by all-1000 centroids so that nothing quantizes to them, ProductQuantizationLearning.java:285-302).
"""
import numpy as np

from . import _native as N

KMEANS_PLUS_PLUS, KMEANS_NORMALIZE = 1, 2


def kmeans(data, numClusters, maxIterations=100, seed=1, kMeansPlusPlus=False, normalize=True, init=None, device=0):
    """Returns (centroids [k' <= numClusters][d], assignment [n], squared error, iterations)."""
    X = np.ascontiguousarray(data, np.float64)
    if X.ndim != 2:
        raise N.MmidxError(N.ERR_INVALID_ARG, "data must be [n][d]")
    n, d = X.shape
    out = np.zeros((numClusters, d), np.float64)
    assign = np.zeros(n, np.int32)
    import ctypes as C

    sse, iters, kout = C.c_double(0.0), C.c_int32(0), C.c_int32(0)
    flags = (KMEANS_PLUS_PLUS if kMeansPlusPlus else 0) | (KMEANS_NORMALIZE if normalize else 0)
    ini = None if init is None else np.ascontiguousarray(init, np.float64)
    if ini is not None and ini.shape != (numClusters, d):
        raise N.MmidxError(N.ERR_INVALID_ARG, "init must be [numClusters][d]")
    st = N.lib().mmidx_kmeans(device, n, d, numClusters, maxIterations, seed, flags, X.ctypes.data,
                             ini.ctypes.data if ini is not None else None, out.ctypes.data, assign.ctypes.data,
                             C.addressof(sse), C.addressof(iters), C.addressof(kout))
    if st != N.OK:
        raise N.MmidxError(st, f"k-means failed ({N.STATUS_NAMES.get(st, st)})")
    return out[:kout.value].copy(), assign, sse.value, iters.value


def _write_centroids(fh, centroids):
    for row in centroids:
        fh.write(",".join(repr(float(v)) for v in row) + "\n")


class AbstractQuantizerLearning:
    @staticmethod
    def learnAndWriteQuantizer(outFilePath, data, numClusters, maxIterations, seed, numSlots=1, kMeansPlusPlus=False):
        """AbstractQuantizerLearning.java:39-81 (numSlots is Weka's thread count: meaningless here)"""
        cent, _, _, _ = kmeans(data, numClusters, maxIterations, seed, kMeansPlusPlus)
        with open(outFilePath, "w") as fh:
            _write_centroids(fh, cent)
        return cent


class CoarseQuantizerLearning:
    @staticmethod
    def learn(vectors, numClusters, maxIterations=100, seed=1, kMeansPlusPlus=False, outFilePath=None):
        """CoarseQuantizerLearning.java:39-72 with the learning vectors given as an array"""
        cent, _, _, _ = kmeans(vectors, numClusters, maxIterations, seed, kMeansPlusPlus)
        if outFilePath:
            with open(outFilePath, "w") as fh:
                _write_centroids(fh, cent)
        return cent


class ResidualVectorComputation:
    """residual = nearest coarse centroid - vector (ResidualVectorComputation.java:27-37: the sign of IVFPQ.java:645)"""

    def __init__(self, coarseQuantizer):
        self.coarse = np.ascontiguousarray(coarseQuantizer, np.float64)

    def computeResidualVectors(self, vectors, device=0):
        import ctypes as C

        X = np.ascontiguousarray(vectors, np.float64)
        Cn, D = self.coarse.shape
        h = C.c_void_p()
        N.check(N.lib().mmidx_create(N.KIND_IVFPQ, D, 1, 2, Cn, N.TR_NONE, None, None, device, C.byref(h)))
        try:
            N.check(N.lib().mmidx_set_coarse(h, self.coarse.ctypes.data))
            import torch  # device buffers only

            dX = torch.from_numpy(X).cuda(device)
            cells = torch.empty(X.shape[0], dtype=torch.int32, device=dX.device)
            N.check(N.lib().mmidx_assign_device(h, X.shape[0], dX.data_ptr(), cells.data_ptr(), None))
            torch.cuda.synchronize()
            c = cells.cpu().numpy()
        finally:
            N.lib().mmidx_destroy(h)
        return self.coarse[c] - X, c


class RandomPermutation:
    """J/utilities/RandomPermutation.java:29-56 on the host -- what the learner applies to its training vectors
    (ProductQuantizationLearning.java:176-178, seed 1) so that the codebooks live in the space the index quantises in
    (IVFPQ.java:136, :193: the same seed; libmmidx_hip derives the same indices natively).  java.util.Random's 48-bit LCG and
    Collections.shuffle restated from the JDK's specification."""

    def __init__(self, seed, dim):
        state = (int(seed) ^ 0x5DEECE66D) & ((1 << 48) - 1)

        def nxt(bits):
            nonlocal state
            state = (state * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
            v = state >> (48 - bits)
            return v - (1 << 32) if v >= (1 << 31) else v  # (int) cast: only next(32) can come out negative

        def next_int(bound):
            if bound & (-bound) == bound:
                return (bound * nxt(31)) >> 31
            while True:
                bits = nxt(31)
                val = bits % bound
                if bits - val + (bound - 1) < (1 << 31):  # (no int overflow)
                    return val

        idx = list(range(dim))
        for i in range(dim, 1, -1):  # Collections.shuffle: swap(i - 1, nextInt(i))
            j = next_int(i)
            idx[i - 1], idx[j] = idx[j], idx[i - 1]
        self.randomlyPermutatedIndices = np.asarray(idx, np.int32)

    def permute(self, v):
        """out[i] = v[perm[i]] (RandomPermutation.java:50-56); rows of a matrix alike"""
        return np.asarray(v)[..., self.randomlyPermutatedIndices]


class ProductQuantizationLearning:
    @staticmethod
    def learn(vectors, m, numProductCentroids, maxIterations=100, numKmeansRepeats=1, coarseQuantizer=None, transform=None,
              outFilePath=None):
        """ProductQuantizationLearning.java:247-305: per sub-space, k-means++ with seeds 1..numKmeansRepeats, keep the
        lowest squared error; residuals w.r.t. the coarse quantizer first when one is given (IVF);
        transform = a callable applied to every (residual) vector (RandomRotation.rotate / RandomPermutation.permute)."""
        X = np.ascontiguousarray(vectors, np.float64)
        n, D = X.shape
        if D % m:
            raise N.MmidxError(N.ERR_INVALID_SUBVECTORS, "d is not a multiple of m")
        if coarseQuantizer is not None:
            X, _ = ResidualVectorComputation(coarseQuantizer).computeResidualVectors(X)
        if transform is not None:
            X = np.stack([transform(v) for v in X])
        dsub = D // m
        pq = np.full((m, numProductCentroids, dsub), 1000.0)  # missing clusters: far-away fake centroids
        for s in range(m):
            sub = np.ascontiguousarray(X[:, s * dsub:(s + 1) * dsub])
            best = None
            for j in range(numKmeansRepeats):
                cent, _, sse, _ = kmeans(sub, numProductCentroids, maxIterations, j + 1, kMeansPlusPlus=True)
                if best is None or sse < best[1]:
                    best = (cent, sse)
            pq[s, :best[0].shape[0]] = best[0]
        if outFilePath:
            with open(outFilePath, "w") as fh:
                for s in range(m):
                    _write_centroids(fh, pq[s])
        return pq
