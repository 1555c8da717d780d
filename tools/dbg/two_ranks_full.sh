#!/bin/bash
# full-size (100 M) functional run of the N = 2 path on one GPU (host-staged gloo) against the unsharded index: answers of batch 0
R=${1:-2}
out=gpurun_out/two_full; mkdir -p $out
A="--no-cpu --gt 0 --hard-steps 0 --other-configs 0 --exhaustive-steps 0 --steps 5 --warmup 1 --settle 2 --nbatches 2"
timeout 600 python bench.py --gpus 1 --batch 32768 --dump /tmp/one $A > $out/one.json 2> $out/one.log
MMIDX_BENCH_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $R --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus $R --batch $((32768 / R)) --dump /tmp/two $A > $out/two.json 2> $out/two.log
python - <<PY
import numpy as np
ref = np.load("/tmp/one.rank0.npz")
parts = [np.load(f"/tmp/two.rank{r}.npz") for r in range($R)]
for key in ("cnt", "iid", "dist"):
    got = np.concatenate([p[key] for p in parts])
    print(key, got.shape, "equal" if np.array_equal(got, ref[key]) else "DIFFERENT")
PY
tail -c 600 $out/two.json
