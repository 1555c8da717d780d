#!/usr/bin/env python3
"""Randomised differential test: random small IVFPQ / PQ configurations through the HIP path and through the CPU oracle
(checker), ids and distance bits compared, ties included (the oracle replays the bounded queue).

    python tests/fuzz_parity.py [cases] [seed]
"""
import importlib
import os
import sys

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):  # (256 BLAS threads under a 16-CPU quota crawl)
    os.environ.setdefault(_v, "8")

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synth  # noqa: E402
from oracle import oracle as o  # noqa: E402  (checker only)

try:
    import torch

    torch.cuda.init()
except Exception:
    pass
mi = importlib.import_module("multimedia-indexing_amd")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(cases):
    kind = rng.choice(["ivfpq", "ivfpq", "pq"])
    m = int(rng.choice([1, 2, 4, 8, 16, 32, 6, 64, 128]))  # (128 x 256 byte codes: the table of twice the LDS, k_scan_split)
    dsub = int(rng.choice([1, 2, 4, 8, 16, 3]))
    if m == 128 and dsub > 4:
        dsub = int(rng.choice([1, 2, 4]))  # (keeps the oracle's encoder and the k-means short)
    if m == 64 and dsub > 8 and rng.random() < 0.5:
        dsub = 4  # (keeps most of the 64 x 16 cases' k-means short; YFCC's own 64 x 16 stays in the mix)
    D = m * dsub
    # (ks > 256 with m >= 32: short codes whose table -- m x ks doubles -- exceeds the LDS and lives in global scratch)
    ks = int(rng.choice([2, 16, 64, 256, 256, 300, 700]))
    if ks == 700 and m < 32:
        ks = 300
    n = int(rng.integers(1, 40000))
    k = int(rng.choice([1, 2, 10, 100, 101, 255, 256, 600, 384, 4095]))
    C = int(rng.choice([1, 2, 7, 40, 130, 300, 1100]))
    w = int(rng.integers(1, C + 1))
    tr = int(rng.choice([0, 0, 2, 1])) if D > 1 else 0
    dup = rng.random() < 0.3
    hist = int(rng.choice([-1, 1, 0]))
    v1 = int(rng.random() < 0.25)
    wide = int(rng.random() < 0.3)
    nogrp = int(rng.random() < 0.2)
    fused = int(rng.random() < 0.3)
    union = int(rng.choice([0, 0, -1, 1]))  # K3g: hint-driven / always / never the instances that rank the union of verified candidates
    spre = int(rng.choice([-1, 1, 1, 0]))  # K3s (certified Smin of every far pair in front of pass B's sort): hint-driven / always / never
    nomf = int(rng.random() < 0.3)  # K3m (the matrix-core bound of pass B) off: K3g / K3f
    mfsub = int(rng.choice([0, 0, 64, 1024]))  # K3m: codes per item (0 = sized from the call)
    mfq = int(rng.choice([0, 0, 0, 8]))  # K3m: survivor records per launch (8: nearly every query goes through the redo path)
    kcv1 = int(rng.random() < 0.3)  # K3mk (D = 256 ... 2048): the form without LDS-DMA also where k_scan_mfma_kc2 applies
    wsel = int(rng.random() < 0.7)  # the coarse stage's exact selection by one wave per query (k_coarse_front_sel) where it applies
    pam = int(rng.choice([-1, -1, 1, 1]))  # K3ma (pass A on the matrix cores, two sweeps): by the batch (never, at seven queries) / forced where the shape allows
    pamw = int(rng.random() < 0.7)  # ... its second sweep by the eight-wave instance
    nosplit = int(rng.random() < 0.25)  # m = 128: the table-in-global kernels instead of k_scan_split
    paq = int(rng.choice([-1, 1, 1, 0]))  # K3q (round 6: pass A decided on integer table sums): by the batch / forced where the shape allows / off
    q_shape = kind == "ivfpq" and rng.random() < 0.3  # ... and three cases in ten of the IVFPQ draws get a shape it takes: 16 x 256 byte codes, k <= 151
    if q_shape:
        m, dsub, ks = 16, int(rng.choice([4, 8, 16])), 256
        D = m * dsub
        k = int(rng.choice([1, 2, 10, 100, 101, 151]))
        C = int(rng.choice([1, 2, 7, 40]))
        w = int(rng.integers(1, C + 1))
        tr = int(rng.choice([0, 0, 2, 1]))
        paq = 1
    # the oracle's encoder is one thread ((C + ks) x D fp64 triples per vector): keep a case near a second of it
    n = max(1, min(n, int(1.5e9 / ((C + ks) * D))))
    desc = dict(kind=kind, D=D, m=m, ks=ks, n=n, k=k, C=C, w=w, tr=tr, dup=dup, hist=hist, v1=v1, wide=wide, nogrp=nogrp, fused=fused, union=union, spre=spre, nomf=nomf, nosplit=nosplit, mfsub=mfsub, mfq=mfq, kcv1=kcv1, wsel=wsel, pam=pam, pamw=pamw, paq=paq)
    try:
        nb = min(n, 3000)
        if kind == "ivfpq":
            p = synth.make_ivfpq_problem(n=max(nb, ks + C + 8), D=D, C=C, m=m, ks=ks, nq=6, seed=int(rng.integers(1 << 30)), iters=2)  # (codebook quality is irrelevant here)
        else:
            p = synth.make_pq_problem(n=max(nb, ks + 8), D=D, m=m, ks=ks, nq=6, seed=int(rng.integers(1 << 30)), iters=2)
        base = rng.standard_normal((n, D)) * 0.6 + (p["coarse"][rng.integers(0, C, n)] if kind == "ivfpq" else 0.0)
        if dup and n > 10:
            base[n // 2:] = base[:n - n // 2]
        perm = o.random_permutation(1, D) if tr == 2 else None
        rot = np.linalg.qr(rng.standard_normal((D, D)))[0] if tr == 1 else None
        if kind == "ivfpq":
            ix = mi.IVFPQ(D, n, False, "", m, ks, tr, C, 512, rot=rot)
            ix.loadCoarseQuantizer(p["coarse"])
            ref = o.OracleIndex(o.KIND_IVFPQ, D, m, ks, C, transform=tr, perm=perm, rot=rot)
            ref.set_coarse(p["coarse"])
            ix.setW(w)
            ref.set_w(w)
        else:
            ix = mi.PQ(D, n, False, "", m, ks, tr, 512, rot=rot)
            ref = o.OracleIndex(o.KIND_PQ, D, m, ks, transform=tr, perm=perm, rot=rot)
        ix.loadProductQuantizer(p["pq"])
        ref.set_pq(p["pq"])
        ix.set_option("passa_hist", hist)
        ix.set_option("coarse_v1", v1)
        ix.set_option("passa_wide", wide)
        ix.set_option("no_grp", nogrp)
        ix.set_option("coarse_fused", fused)
        ix.set_option("no_union", union)
        ix.set_option("smin_pre", spre)
        ix.set_option("no_mfma", nomf)
        ix.set_option("mfma_sub", mfsub)
        ix.set_option("mfma_qcap", mfq)
        ix.set_option("mfma_kc_v1", kcv1)
        ix.set_option("mfma_kc_tpw", 16 if case % 7 == 3 else 8)
        ix.set_option("coarse_wave_sel", wsel)
        ix.set_option("passa_mfma", pam if not q_shape else -1)
        ix.set_option("passa_q", paq)
        ix.set_option("passa_mfma_wide", pamw)
        ix.set_option("no_split_table", nosplit)
        ix.set_option("smin_valu", int(case % 3 == 0))
        ix.set_option("smin_bf16", int(case % 4 != 1))
        ix.set_option("coarse_dma_kc", int(case % 5 != 2))  # (K1e' with LDS-DMA for vectors of several 128-dimension chunks)
        half = n // 2
        ix.indexVectors([str(i) for i in range(half)], base[:half])
        if half:
            ix.search_batch(min(k, 3), base[:2])  # a search between the two adds: the CSR is rebuilt incrementally
        ix.indexVectors([str(i) for i in range(half, n)], base[half:])
        ref.add_vectors(base)
        nself = 24 if q_shape else 4  # (K3q: several queries per list -- groups of one to four pairs, several groups per list)
        Q = np.concatenate([base[rng.integers(0, n, nself)] + 0.01 * rng.standard_normal((nself, D)), p["queries"][:3]])
        got = ix.search_batch(k, Q)
        want = ref.search_batch(Q, k)
        # ids, counts and distance BITS, rotation included: the kernels rotate in the oracle's order (sequential over the
        # row index, RandomRotation.java:44-49 through a plain row-vector x matrix product) -- the same statement
        # test_ivfpq_transforms makes.  What stays an assumption (A2) is that EJML's CommonOps.mult uses that order too;
        # against the Java classes the promise for a rotation is 1e-12, against this oracle it is equality.
        ok = np.array_equal(got[2], want[2]) and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        ix.close()
        if not ok:
            bad += 1
            print("MISMATCH", desc, flush=True)
    except mi.MmidxError as e:
        print("native error (acceptable if UNSUPPORTED)", e.status, desc, str(e)[:80], flush=True)
        if e.status != 10:
            bad += 1
print(f"{cases} cases, {bad} bad")
sys.exit(1 if bad else 0)
