#!/usr/bin/env python3
"""pass A's gather: LDS alone against kg rows on the vector-memory path (mmidx_probe_split_gather)"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
nat = importlib.import_module("multimedia-indexing_amd._native")
L = nat.lib()
v = (C.c_double * 4)()
for rep in range(2):
    nat.check(L.mmidx_probe_lds_gather(0, 16, 3, v)); print("lds only   ", f"{v[0]:.3e} gathers/s  {v[1]:.0f} GB/s  {v[3]*1e3:.3f} ms")
    for kg in (1, 2, 4):
        nat.check(L.mmidx_probe_split_gather(0, kg, v)); print(f"kg = {kg}     ", f"{v[0]:.3e} gathers/s  {v[1]:.0f} GB/s  {v[3]*1e3:.3f} ms")
