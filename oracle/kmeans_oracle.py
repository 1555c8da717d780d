"""CPU restatement of the clustering behind the reference's codebook learners -- TEST INFRASTRUCTURE (only tests/ import it).

PARITY UNPINNED.  The reference learns its coarse and product quantizers with Weka's SimpleKMeans:
    J/quantization/AbstractQuantizerLearning.java:39-81   learnAndWriteQuantizer: new SimpleKMeans(), optional
        setInitializationMethod(KMEANS_PLUS_PLUS) (:53-56), setSeed(seed) (:58), setNumClusters (:59),
        setMaxIterations (:60), setFastDistanceCalc(true) (:62), buildClusterer(data) (:64), centroids written one per
        line, comma separated (:73-80)
    J/quantization/CoarseQuantizerLearning.java:39-72, ProductQuantizationLearning.java:247-305   the callers
Weka (weka-dev 3.7.x, pom.xml) is a third-party dependency that is absent from /root/reference and from this image, and the
reference ships no test or fixture for the learners, so nothing pins this file to Weka's output.

What IS Weka's, restated from its published algorithm (SimpleKMeans.buildClusterer):
    * default seeding: walk j = n-1 .. 0, pick instIndex = Random(seed).nextInt(j + 1), take the instance as a centre if
      no equal centre exists yet, swap it to position j, stop at k centres                          -> random_seeding()
    * k-means++ seeding: first centre = instance Random(seed).nextInt(n), every next centre drawn with probability
      proportional to the squared distance to the nearest centre so far (cumulative sum, nextDouble)   -> plus_plus_seeding()
    * distance = EuclideanDistance with attribute normalisation to [0, 1] (min-max over the data) by default
    * Lloyd iterations until no instance changes cluster or maxIterations; empty clusters are dropped -> lloyd()
    * java.util.Random: the JDK's documented 48-bit LCG                                              -> JavaRandom
What is NOT Weka's and is this build's own choice (libmmidx_hip's kernels make the same one, which is what the GPU tests
compare bit for bit): the nearest centre is the exact sequential fp64 squared distance with the FIRST index winning ties, a
centroid is the sum of its members in ascending index order divided by their number, and after an iteration that dropped a
cluster every instance counts as "changed".  Weka's own summation order and tie rule are not known here.
"""
import numpy as np


class JavaRandom:
    """java.util.Random (JDK javadoc): seed scrambling, next(bits), nextInt(bound), nextDouble()"""

    def __init__(self, seed):
        self.s = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1)

    def next(self, bits):
        self.s = (self.s * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        v = self.s >> (48 - bits)
        return v - (1 << bits) if v >= (1 << (bits - 1)) and bits == 32 else v

    def nextInt(self, bound):
        r = self.next(31)
        m = bound - 1
        if bound & m == 0:
            return (bound * r) >> 31
        u = r
        while True:
            r = u % bound
            if u - r + m < (1 << 31):
                return r
            u = self.next(31)

    def nextDouble(self):
        return ((self.next(26) << 27) + self.next(27)) * 2.0 ** -53


def seq_sqdist(x, c):
    """sequential fp64 squared distance, dimension ascending"""
    acc = 0.0
    for a, b in zip(x, c):
        df = a - b
        acc += df * df
    return acc


def random_seeding(X, k, seed):
    """SimpleKMeans' default initialisation (indices of the picked instances)"""
    r = JavaRandom(seed)
    perm, picks = list(range(len(X))), []
    for j in range(len(X) - 1, -1, -1):
        i = r.nextInt(j + 1)
        picks.append(perm[i])
        perm[j], perm[i] = perm[i], perm[j]
        if len(picks) == k:
            break
    return picks


def plus_plus_seeding(X, k, seed, margin=1e-9):
    """k-means++ initialisation; raises when a draw lands within `margin` (relative) of a bucket edge, where the order of a
    parallel prefix sum could pick the neighbouring instance"""
    r = JavaRandom(seed)
    picks = [r.nextInt(len(X))]
    d2 = None
    for _ in range(1, k):
        nd = np.array([seq_sqdist(x, X[picks[-1]]) for x in X])
        d2 = nd if d2 is None else np.minimum(d2, nd)
        cum = np.cumsum(d2)
        target = r.nextDouble() * cum[-1]
        idx = int(np.searchsorted(cum, target, side="right"))
        if abs(cum[min(idx, len(X) - 1)] - target) <= margin * cum[-1]:
            raise ValueError("fixture too close to a bucket edge")
        picks.append(min(idx, len(X) - 1))
    return picks


def lloyd(X, C0, max_iter):
    """Lloyd iterations: sequential fp64 distances, first index wins, index-ordered sums, empty clusters dropped.
    Returns (centroids, assignment into them, iterations)"""
    C = np.array(C0, dtype=np.float64).copy()
    a_old = np.full(len(X), -1)
    iters = 0
    while True:
        iters += 1
        a = np.array([int(np.argmin([seq_sqdist(x, c) for c in C])) for x in X])
        changed = int((a != a_old).sum())
        newC, keep = [], []
        for c in range(len(C)):
            mem = np.nonzero(a == c)[0]
            if len(mem):
                acc = np.zeros(X.shape[1])
                for i in mem:
                    acc = acc + X[i]
                newC.append(acc / float(len(mem)))
                keep.append(c)
        done = changed == 0 or iters >= max_iter
        dropped = len(keep) != len(C)
        remap = {c: t for t, c in enumerate(keep)}
        C = np.array(newC)
        if done:
            return C, np.array([remap[c] for c in a]), iters
        a_old = np.full(len(X), -1) if dropped else a


def minmax_normalise(X):
    """EuclideanDistance's default attribute normalisation (the clustering then runs in this space)"""
    X = np.asarray(X, np.float64)
    return (X - X.min(0)) / (X.max(0) - X.min(0))
