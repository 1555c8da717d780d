"""GPU codebook learning (csrc/mmidx_learn.hip) against the CPU restatement of the same algorithm (oracle/kmeans_oracle.py).

Weka's SimpleKMeans is an absent third-party dependency (parity with the reference unpinned for this row);
what IS pinned here: from identical seeds the GPU path and the numpy restatement below produce bit-identical
centroids (exact fp64 argmin with first-index ties, means as index-ordered sums), the JDK random stream of the
seeding, the dropping of empty clusters, and the file formats the reference's loaders read."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mi():
    try:
        import torch

        torch.cuda.init()
    except Exception:
        pass
    m = importlib.import_module("multimedia-indexing_amd")
    if m.lib().mmidx_device_count() < 1:
        pytest.fail("libmmidx_hip.so found no HIP device")
    return m


from oracle import kmeans_oracle as ko  # noqa: E402  (the CPU restatement: oracle/kmeans_oracle.py, parity unpinned -- Weka is absent)

JavaRandom, seq_sqdist, lloyd_twin = ko.JavaRandom, ko.seq_sqdist, ko.lloyd


def test_lloyd_matches_twin_bit_for_bit(mi):
    q = mi.quantization
    rng = np.random.default_rng(3)
    X = np.concatenate([rng.standard_normal((150, 6)) + 4 * rng.standard_normal((1, 6)) for _ in range(5)])
    init = X[rng.choice(len(X), 12, replace=False)]
    cent, assign, sse, iters = q.kmeans(X, 12, maxIterations=50, init=init, normalize=False)
    tC, tA, tI = lloyd_twin(X, init, 50)
    assert iters == tI and cent.shape == tC.shape
    assert np.array_equal(assign, tA)
    assert np.array_equal(cent, tC)
    assert sse == pytest.approx(sum(seq_sqdist(x, cent[a]) for x, a in zip(X, assign)), rel=1e-12)


def test_random_seeding_follows_the_jdk_stream(mi):
    q = mi.quantization
    rng = np.random.default_rng(5)
    X = rng.standard_normal((400, 4))
    k, seed = 7, 1
    picks = ko.random_seeding(X, k, seed)
    cent, assign, _, iters = q.kmeans(X, k, maxIterations=1, seed=seed, normalize=False)
    tC, tA, _ = lloyd_twin(X, X[picks], 1)
    assert iters == 1 and np.array_equal(cent, tC) and np.array_equal(assign, tA)


def test_kmeans_plus_plus_seeding(mi):
    q = mi.quantization
    rng = np.random.default_rng(6)
    X = rng.standard_normal((300, 3))
    k, seed = 5, 2
    picks = ko.plus_plus_seeding(X, k, seed)  # (raises if a draw sits on a bucket edge, where a parallel scan could differ)
    cent, assign, _, _ = q.kmeans(X, k, maxIterations=1, seed=seed, kMeansPlusPlus=True, normalize=False)
    tC, tA, _ = lloyd_twin(X, X[picks], 1)
    assert np.array_equal(cent, tC) and np.array_equal(assign, tA)


def test_empty_clusters_are_dropped_and_normalisation(mi):
    q = mi.quantization
    rng = np.random.default_rng(8)
    X = np.concatenate([rng.standard_normal((100, 2)) * 0.1, rng.standard_normal((100, 2)) * 0.1 + [50.0, 0.0]])
    init = np.array([[0.0, 0.0], [50.0, 0.0], [1000.0, 1000.0], [25.0, 500.0]])  # two centres nobody is close to
    cent, assign, _, _ = q.kmeans(X, 4, maxIterations=20, init=init, normalize=False)
    assert cent.shape == (2, 2) and set(assign.tolist()) == {0, 1}
    assert np.allclose(cent[0], X[:100].mean(0)) and np.allclose(cent[1], X[100:].mean(0))
    # Weka's default distance normalises every attribute to [0, 1]: a badly scaled attribute no longer dominates
    Y = np.stack([np.r_[np.zeros(100), np.ones(100)] + 0.01 * rng.standard_normal(200), 1e4 * rng.standard_normal(200)], 1)
    cN, aN, _, _ = q.kmeans(Y, 2, maxIterations=50, init=Y[[0, 150]], normalize=True)
    twin_norm = ko.minmax_normalise(Y)
    tC, tA, _ = lloyd_twin(twin_norm, twin_norm[[0, 150]], 50)
    assert np.array_equal(aN, tA)
    for c in range(2):  # centroids are reported in the original space: means of the un-normalised members
        assert np.allclose(cN[c], Y[aN == c].mean(0), rtol=1e-12)


def test_learned_quantizers_feed_the_index(mi, tmp_path):
    """coarse + residual product quantizer learned on the GPU, written in the reference's file formats, read back by
    the loaders, index + search -> sane recall on the learning distribution."""
    q = mi.quantization
    rng = np.random.default_rng(11)
    D, C, m, ks, n = 16, 8, 4, 16, 4000
    mu = 3.0 * rng.standard_normal((C, D))
    X = mu[rng.integers(0, C, n)] + 0.3 * rng.standard_normal((n, D))
    cq_file, pq_file = str(tmp_path / "qcoarse.csv"), str(tmp_path / "pq.csv")
    coarse = q.CoarseQuantizerLearning.learn(X, C, maxIterations=30, seed=1, kMeansPlusPlus=True, outFilePath=cq_file)
    assert coarse.shape == (C, D)
    pq = q.ProductQuantizationLearning.learn(X, m, ks, maxIterations=30, numKmeansRepeats=2, coarseQuantizer=coarse, outFilePath=pq_file)
    assert pq.shape == (m, ks, D // m)
    assert len(open(pq_file).read().strip().split("\n")) == m * ks
    ix = mi.IVFPQ(D, n, False, "", m, ks, mi.TransformationType.None_, C, 512)
    ix.loadCoarseQuantizer(cq_file)
    ix.loadProductQuantizer(pq_file)
    ix.setW(3)
    ix.indexVectors([str(i) for i in range(n)], X)
    hits = 0
    for i in range(0, 200):
        ans = ix.computeNearestNeighbors(5, X[i] + 0.001 * rng.standard_normal(D))
        hits += str(i) in ans.getIds()
    assert hits >= 150
    ix.close()
