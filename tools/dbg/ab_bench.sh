#!/bin/bash
# A/B of kernel variants built with `make -C multimedia-indexing_amd/csrc variant NAME=x EXTRA=-D...`:
#   tools/dbg/ab_bench.sh outdir name1 name2 ...   ("base" = the in-tree library)
out=gpurun_out/$1; shift
mkdir -p $out
for v in "$@"; do
  if [ "$v" = base ]; then unset MMIDX_LIB; else export MMIDX_LIB=$PWD/multimedia-indexing_amd/csrc/ab/libmmidx_$v.so; fi
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --gt 0 --exhaustive-steps 0 --hard-steps 0 --other-configs 0 $ABARGS > $out/$v.json 2> $out/$v.log
  python - <<PY
import json
try:
    j = json.loads(open("$out/$v.json").read().strip().splitlines()[-1])
    w = j.get("roofline_whole_search") or {}
    print("$v", j["value"], j["ms_per_step"], "passA ms", j["roofline"].get("avg_launch_ms"), "frac", j["roofline"].get("frac"), "coarse ms", w.get("coarse_ms_per_step"), "merge ms", w.get("merge_ms_per_step"))
except Exception as e:
    print("$v failed", e)
PY
done
