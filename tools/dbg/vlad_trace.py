#!/usr/bin/env python3
"""cfg5's front end alone (tests/bench_extras.cfg5 without the end-to-end leg): for kernel traces of the VLAD / PCA kernels"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import bench_extras as be
mi = importlib.import_module("multimedia-indexing_amd")
nat = importlib.import_module("multimedia-indexing_amd._native")
out = be.cfg5(mi.lib(), nat, mi, images_e2e=0)
print(json.dumps({k: v for k, v in out.items() if "vlad" in k or "fused" in k}))
