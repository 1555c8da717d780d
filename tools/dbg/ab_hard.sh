#!/bin/bash
# A/B of kernel variants on the hard workload (sigma 1.0: every far probe survives the coarse bound):
#   tools/dbg/ab_hard.sh outdir name1 name2 ...   ("base" = the in-tree library)
out=gpurun_out/$1; shift
mkdir -p $out
for v in "$@"; do
  if [ "$v" = base ]; then unset MMIDX_LIB; else export MMIDX_LIB=$PWD/multimedia-indexing_amd/csrc/ab/libmmidx_$v.so; fi
  timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu --gt 0 --exhaustive-steps 0 --other-configs 0 --hard-steps 6 $ABARGS > $out/$v.json 2> $out/$v.log
  python - <<PY
import json
try:
    j = json.loads(open("$out/$v.json").read().strip().splitlines()[-1])
    h = j["hard"]
    print("$v", "headline", j["value"], "hard", h["value"], h["ms_per_step"], "passB ms", h["roofline"]["avg_launch_ms"], "parity", h["parity"])
except Exception as e:
    print("$v failed", e)
PY
done
