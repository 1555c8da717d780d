#!/usr/bin/env python3
"""PCA / VLAD front-end timings of the library MMIDX_LIB points at (A/B of kernel variants): python tools/dbg/pca_time.py"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

torch.cuda.init()
mi = importlib.import_module("multimedia-indexing_amd")
nat = importlib.import_module("multimedia-indexing_amd._native")
ex = importlib.import_module("bench_extras")
r = ex.cfg5(mi.lib(), nat, mi, images_e2e=0)
print(os.environ.get("MMIDX_LIB", "default"), json.dumps({k: r[k] for k in ("pca_8192_to_128",)}))
