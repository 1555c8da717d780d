#!/usr/bin/env python3
"""cfg5 front end alone (tests/bench_extras.cfg5): PCA, VLAD, fused call, end to end"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "examples")):
    sys.path.insert(0, p)
import torch
torch.cuda.init()
mi = importlib.import_module("multimedia-indexing_amd")
nat = importlib.import_module("multimedia-indexing_amd._native")
import bench_extras as be
r = be.cfg5(mi.lib(), nat, mi, images_e2e=int(sys.argv[1]) if len(sys.argv) > 1 else 1000000)
print(json.dumps({k: v for k, v in r.items() if k != "pca_8192_to_128"}))
