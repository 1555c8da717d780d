import importlib, sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import synth
from oracle import oracle as o
torch.cuda.init()
mi = importlib.import_module("multimedia-indexing_amd")
sh = importlib.import_module("multimedia-indexing_amd.sharded")
S=2
D, C, m, ks, n, w = 32, 24, 8, 256, 9000, 6
p = synth.make_ivfpq_problem(n=n, D=D, C=C, m=m, ks=ks, nq=40, seed=5 + S)
ref = o.OracleIndex(o.KIND_IVFPQ, D, m, ks, C); ref.set_coarse(p["coarse"]); ref.set_pq(p["pq"]); ref.set_w(w); ref.add_vectors(p["base"])
full = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512); full.loadCoarseQuantizer(p["coarse"]); full.loadProductQuantizer(p["pq"]); full.setW(w)
cells, codes = full.encode(p["base"])
shards=[]
for r in range(S):
    ix = mi.IVFPQ(D, n, False, "", m, ks, 0, C, 512); ix.loadCoarseQuantizer(p["coarse"]); ix.loadProductQuantizer(p["pq"]); ix.setW(w)
    own = np.nonzero(sh.owner_of_cell(cells, S) == r)[0]
    ix.loadIndex(own.astype(np.int32), cells[own], codes[own]); shards.append(ix)
Q = torch.tensor(p["queries"], dtype=torch.float64, device="cuda")
engines = [sh.HipShardEngine(ix._h, D, w, 0) for ix in shards]
probe = engines[0].coarse(Q)
torch.cuda.synchronize()
print("probe0", probe[0].cpu().numpy(), "oracle", ref.nearest_coarse(p["queries"][0], w))
k=10
parts = [e.search_partial(k, Q, probe) for e in engines]
torch.cuda.synchronize()
for s_,x in enumerate(parts):
    print("shard", s_, "cnt", x[2][0].item(), "d", x[0][0][:5].cpu().numpy(), "key", [(int(v)>>32, int(v)&0xffffffff) for v in x[1][0][:5].cpu().numpy()])
rid, rd = ref.search(p["queries"][0], k)
print("oracle", rid, rd[:5])
iid, dd, cnt = engines[0].merge(k, torch.stack([x[0] for x in parts]), torch.stack([x[1] for x in parts]), torch.stack([x[2] for x in parts]))
torch.cuda.synchronize()
print("devmerge", iid[0].cpu().numpy(), dd[0][:5].cpu().numpy())
hi, hd, hc = sh.merge_partials_host(k, torch.stack([x[0] for x in parts]).cpu().numpy(), torch.stack([x[1] for x in parts]).cpu().numpy(), torch.stack([x[2] for x in parts]).cpu().numpy())
print("hostmerge", hi[0], hd[0][:5])
