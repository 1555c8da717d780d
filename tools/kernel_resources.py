#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    hipcc --offload-arch=gfx950 ... -c mmidx_api.hip -o /tmp/x.o -Rpass-analysis=kernel-resource-usage 2> res.txt
    python tools/kernel_resources.py res.txt
"""
import re
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
KEYS = [("vgpr", r"VGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"SGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"),
        ("occ", r"Occupancy \[waves/SIMD\]"), ("lds", r"LDS Size \[bytes/block\]")]
for b in blocks:
    name = b.split("\n")[0].strip()
    vals = []
    for label, pat in KEYS:
        mm = re.search(pat + r": (\d+)", b)
        vals.append(f"{label} {mm.group(1) if mm else '?':>5}")
    print(f"{name[:70]:70s} " + " ".join(vals))
