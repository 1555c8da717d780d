import sys; sys.path[:0]=[".","tests"]
import importlib, bench_extras
mi=importlib.import_module("multimedia-indexing_amd"); nat=importlib.import_module("multimedia-indexing_amd._native")
import torch; torch.cuda.init()
r=bench_extras.cfg5(mi.lib(), nat, mi, images_e2e=0, device=0)
print({k:(round(v.get("images_per_s")/1e6,3), v.get("ms")) for k,v in r.items() if isinstance(v,dict)})
