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


def run_device(n_images=100_000, n_queries=256, k=10, seed=0, cells=1024, w=8, chunk=8192, n_sample=8, verbose=False):
    """The same pipeline with the front end resident on the device (BASELINE config 5 at scale): descriptors are synthesised
    in HBM chunk by chunk (torch), `mmidx_vectorize_device` turns a chunk of images into 128-d vectors without the 8192-d VLAD
    vectors ever leaving the device, the vectors go to the index.  Returns (summary, everything a checker needs: the projected
    vectors, the queries, the results, the quantizers, the PCA / codebook parameters and the raw descriptors of a few sample
    images)."""
    import ctypes as C

    import torch

    mi = importlib.import_module("multimedia-indexing_amd")
    nat = importlib.import_module("multimedia-indexing_amd._native")
    L = mi.lib()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(seed)
    dl, ncent, nc_out = 64, 128, 128
    codebook = rng.standard_normal((ncent, dl)) / 8.0
    topics = torch.from_numpy(rng.standard_normal((64, 24, dl))).to(dev)
    ss = ncent * dl
    Vt = np.linalg.qr(rng.standard_normal((ss, nc_out)))[0].T.copy()
    eig = np.linspace(4.0, 0.5, nc_out)
    vlad = mi.VladAggregatorMultipleVocabularies([codebook], normalizationsOn=True)

    def synth_chunk(c0, n, noise=0.35):
        g = torch.Generator(device=dev)
        g.manual_seed(777 + c0)
        topic = torch.randint(0, 64, (n,), generator=g, device=dev)
        nd = torch.randint(200, 801, (n,), generator=g, device=dev)
        off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        off[1:] = torch.cumsum(nd, 0)
        tot = int(off[-1].item())
        img = torch.repeat_interleave(torch.arange(n, device=dev), nd, output_size=tot)
        word = torch.randint(0, 24, (tot,), generator=g, device=dev)
        d = topics[topic[img], word] + noise * torch.randn(tot, dl, generator=g, device=dev, dtype=torch.float64)
        d = d / d.norm(dim=1, keepdim=True)
        return off, d.contiguous(), int(nd.max().item())

    t0 = time.time()
    # PCA means = mean VLAD vector of the first chunk (whitening on; basis and singular values synthetic)
    off0, d0, mx0 = synth_chunk(0, min(chunk, n_images))
    n0 = off0.numel() - 1
    V0 = torch.empty(n0, ss, dtype=torch.float64, device=dev)
    nat.check(L.mmidx_vlad_aggregate_device(vlad._h, n0, off0.data_ptr(), d0.data_ptr(), mx0, V0.data_ptr(), stream))
    torch.cuda.synchronize()
    means = V0.mean(0).cpu().numpy()
    del V0
    pca = mi.PCA(nc_out, 0, ss, True)
    pca.load(means, eig, Vt)
    X = torch.empty(n_images, nc_out, dtype=torch.float64, device=dev)
    sample_descs, sample_ids = [], []
    qdesc = None
    for c0 in range(0, n_images, chunk):
        n = min(chunk, n_images - c0)
        off, d, mx = (off0, d0, mx0) if c0 == 0 else synth_chunk(c0, n)
        nat.check(L.mmidx_vectorize_device(vlad._h, pca._h, n, off.data_ptr(), d.data_ptr(), mx, X[c0:c0 + n].data_ptr(), stream))
        torch.cuda.synchronize()
        if c0 == 0:
            oh = off.cpu().numpy()
            for j in range(min(n_sample, n)):
                sample_ids.append(j)
                sample_descs.append(d[oh[j]:oh[j + 1]].cpu().numpy())
            # queries: re-rendered copies of the first n_queries images (fresh noise on every descriptor)
            nqi = min(n_queries, n)
            qoff = off[:nqi + 1].clone()
            gq = torch.Generator(device=dev)
            gq.manual_seed(4242)
            qdesc = (qoff, (d[:int(oh[nqi])] + 0.02 * torch.randn(int(oh[nqi]), dl, generator=gq, device=dev, dtype=torch.float64)).contiguous(), mx)
    t_front = time.time() - t0
    qoff, qd, qmx = qdesc
    nq = qoff.numel() - 1
    Qd = torch.empty(nq, nc_out, dtype=torch.float64, device=dev)
    nat.check(L.mmidx_vectorize_device(vlad._h, pca._h, nq, qoff.data_ptr(), qd.data_ptr(), qmx, Qd.data_ptr(), stream))
    torch.cuda.synchronize()
    # what the DATA allows: the exact fp64 nearest neighbour (Linear semantics) of every query among the 128-d vectors -- is it the
    # image the query was re-rendered from?  (the engine's self-hit rate below cannot exceed what 16-byte codes make of this)
    xn = (X * X).sum(1)
    exact_hits = 0
    for q0 in range(0, nq, 256):
        dd = xn[None, :] - 2.0 * (Qd[q0:q0 + 256] @ X.T)
        exact_hits += int((dd.argmin(1) == torch.arange(q0, min(q0 + 256, nq), device=dev)).sum().item())
    exact_self_hit = exact_hits / max(1, nq)
    del xn
    Xh, Q = X.cpu().numpy(), Qd.cpu().numpy()
    D, m, ks = nc_out, 16, 256
    t0 = time.time()
    learn = Xh[: min(n_images, 40000)]
    coarse = mi.quantization.CoarseQuantizerLearning.learn(learn, cells, maxIterations=10, seed=1, kMeansPlusPlus=True)
    if coarse.shape[0] < cells:
        coarse = np.concatenate([coarse, np.full((cells - coarse.shape[0], D), 1000.0)])
    pq = mi.quantization.ProductQuantizationLearning.learn(learn[:20000], m, ks, maxIterations=8, numKmeansRepeats=1, coarseQuantizer=coarse)
    t_learn = time.time() - t0
    ix = mi.IVFPQ(D, n_images, False, "", m, ks, mi.TransformationType.None_, cells, 512)
    ix.loadCoarseQuantizer(coarse)
    ix.loadProductQuantizer(pq)
    ix.setW(w)
    t0 = time.time()
    ix.indexVectors([f"img{i}" for i in range(n_images)], Xh)
    t_index = time.time() - t0
    t0 = time.time()
    iids, dists, counts = ix.search_batch(k, Q)
    t_search = time.time() - t0
    out = {"images": n_images, "queries": int(nq), "k": k, "self_hit_rate": float(np.mean(iids[:, 0] == np.arange(nq))),
           "exact_self_hit_rate": exact_self_hit,
           "self_hit_note": "self_hit_rate: IVFPQ top-1 = the re-rendered image; exact_self_hit_rate: the same with exact fp64 brute force over the 128-d vectors "
                            "(what the synthetic data allows before any quantization); engine and oracle agree bit for bit on this pipeline "
                            "(tests/test_gpu_frontend.py::test_config5_pipeline_end_to_end)",
           "images_per_s_front_end": round(n_images / t_front, 1),
           "seconds": {"synthesis_and_front_end": round(t_front, 2), "learn_quantizers": round(t_learn, 2), "index": round(t_index, 3),
                       "search": round(t_search, 4)}}
    for o_ in (vlad, pca, ix):
        o_.close()
    if verbose:
        print(json.dumps(out))
    return out, {"X": Xh, "Q": Q, "iids": iids, "dists": dists, "counts": counts, "coarse": coarse, "pq": pq, "cells": cells, "w": w, "k": k,
                 "sample_ids": sample_ids, "sample_descs": sample_descs, "codebook": codebook, "Vt": Vt, "eig": eig, "means": means}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=20000)
    ap.add_argument("--queries", type=int, default=256)
    ap.add_argument("--device-front-end", action="store_true", help="synthesise and vectorise on the device (config 5 at scale)")
    a = ap.parse_args()
    if a.device_front_end:
        run_device(a.images, a.queries, verbose=True)
    else:
        run(a.images, a.queries)
