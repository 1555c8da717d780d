#!/usr/bin/env python3
"""BASELINE config 5, end to end on one MI355X, all stages through the C ABI:

  synthetic SURF-64 descriptors -> batched VLAD (128 centroids -> 8192-d, power + L2)
     -> PCA 8192 -> 128 with whitening (f64 MFMA)  -> IVFPQ index / search

mirrors the reference's ImageVectorization.transformToVector (J/vectorization/ImageVectorization.java:
169-208: aggregate, then PCA.sampleToEigenSpace) followed by indexVector / computeNearestNeighbors.
Codebooks and the PCA basis are synthetic (learning is offline in the reference).

  python examples/config5_pipeline.py --images 20000
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import synth  # noqa: E402


def run(n_images=20000, n_queries=256, k=10, seed=0, cells=256, w=8, verbose=True):
    mi = importlib.import_module("multimedia-indexing_amd")
    rng = np.random.default_rng(seed)
    dl, ncent, nc_out = 64, 128, 128
    # "visual words" + images = mixtures of a few topics, so that near-duplicates exist
    codebook = rng.standard_normal((ncent, dl)) / 8.0
    topics = rng.standard_normal((64, 24, dl))
    def make_image(t, noise):
        nd = int(rng.integers(200, 801))
        base = topics[t][rng.integers(0, 24, size=nd)]
        d = base + noise * rng.standard_normal((nd, dl))
        return d / np.linalg.norm(d, axis=1, keepdims=True)
    t0 = time.time()
    topic_of = rng.integers(0, 64, size=n_images)
    images = [make_image(t, 0.35) for t in topic_of]
    t_gen = time.time() - t0
    vlad = mi.VladAggregatorMultipleVocabularies([codebook], normalizationsOn=True)
    t0 = time.time()
    V = np.concatenate([vlad.aggregate_batch(images[i:i + 4096]) for i in range(0, n_images, 4096)])
    t_vlad = time.time() - t0
    # PCA basis: random orthonormal rows, synthetic singular values, means = sample mean (whitening on)
    ss = ncent * dl
    Vt = np.linalg.qr(rng.standard_normal((ss, nc_out)))[0].T.copy()
    eig = np.linspace(4.0, 0.5, nc_out)
    pca = mi.PCA(nc_out, 0, ss, True)
    pca.load(V.mean(0), eig, Vt)
    t0 = time.time()
    X = pca.project(V)
    t_pca = time.time() - t0
    # IVFPQ over the projected vectors; quantizers learned on the GPU (quantization.py = the reference's
    # CoarseQuantizerLearning / ProductQuantizationLearning with Weka's k-means restated in HIP)
    D, m, ks = nc_out, 16, 256
    t0 = time.time()
    learn = X[: min(n_images, 20000)]
    coarse = mi.quantization.CoarseQuantizerLearning.learn(learn, cells, maxIterations=10, seed=1, kMeansPlusPlus=True)
    if coarse.shape[0] < cells:  # dropped clusters: pad like the product-quantizer learner does
        coarse = np.concatenate([coarse, np.full((cells - coarse.shape[0], D), 1000.0)])
    pq = mi.quantization.ProductQuantizationLearning.learn(learn, m, ks, maxIterations=8, numKmeansRepeats=1, coarseQuantizer=coarse)
    t_learn = time.time() - t0
    ix = mi.IVFPQ(D, n_images, False, "", m, ks, mi.TransformationType.None_, cells, 512)
    ix.loadCoarseQuantizer(coarse)
    ix.loadProductQuantizer(pq)
    ix.setW(w)
    t0 = time.time()
    ix.indexVectors([f"img{i}" for i in range(n_images)], X)
    t_index = time.time() - t0
    # queries: re-rendered copies of indexed images (same topic words, fresh noise) through the same front end
    qi = rng.choice(n_images, n_queries, replace=False)
    qimgs = [images[i] + 0.02 * rng.standard_normal(images[i].shape) for i in qi]
    Q = mi.frontend.ImageVectorizer(vlad, pca).transform_batch(qimgs)  # descriptors -> 128-d in one native call
    t0 = time.time()
    iids, dists, counts = ix.search_batch(k, Q)
    t_search = time.time() - t0
    exact = ((Q[:, None, :] - X[None]) ** 2).sum(-1).argmin(1) if n_queries * n_images * D < 4e8 else \
        ((Q * Q).sum(1)[:, None] - 2 * Q @ X.T + (X * X).sum(1)[None]).argmin(1)
    out = {"images": n_images, "queries": n_queries, "k": k,
           "recall_at_1_vs_exact": float(np.mean(iids[:, 0] == exact)),
           "self_hit_rate": float(np.mean(iids[:, 0] == qi)),
           "seconds": {"descriptor_synthesis_cpu": round(t_gen, 2), "vlad": round(t_vlad, 3), "pca": round(t_pca, 3), "learn_quantizers": round(t_learn, 3),
                       "index": round(t_index, 3), "search": round(t_search, 4)},
           "answer0": ix.computeNearestNeighbors(k, Q[0]).getIds()[:3]}
    for o in (vlad, pca, ix):
        o.close()
    if verbose:
        print(json.dumps(out))
    return out, (X, Q, iids, dists, counts, coarse, pq, cells, w, k)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=20000)
    ap.add_argument("--queries", type=int, default=256)
    a = ap.parse_args()
    run(a.images, a.queries)
