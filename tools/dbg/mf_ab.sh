#!/bin/bash
# same-box A/B of library variants (tools/dbg/build_variant.sh) on the hard workload (sigma 1.0 as the main object) and, with SPREAD=1, spread:
#   tools/dbg/mf_ab.sh outdir name1 name2 ...   ("base" = the in-tree library)
out=gpurun_out/$1; shift
mkdir -p $out
for v in "$@"; do
  if [ "$v" = base ]; then unset MMIDX_LIB; else export MMIDX_LIB=$PWD/tools/dbg/lib_$v.so; fi
  timeout 400 python bench.py --sigma 1.0 --steps 8 --warmup 2 --no-cpu --gt 0 --exhaustive-steps 0 --other-configs 0 --extras 0 --hard-steps 0 --spread-steps ${SPREAD_STEPS:-0} $ABARGS > $out/$v.json 2> $out/$v.log
  python - <<PY
import json
try:
    j = json.loads(open("$out/$v.json").read().strip().splitlines()[-1])
    print("$v", "hard-as-main", round(j["value"]), j["ms_per_step"], {k: v for k, v in j.items() if k.endswith("_ms_per_step")})
    s = j.get("spread")
    if s: print("$v", "spread", round(s["value"]), s["ms_per_step"], s["stage_ms_per_step"], "surv/q", s.get("mfma_survivors_per_query"), "redo", s.get("mfma_redo_queries_per_step"))
except Exception as e:
    print("$v failed", e)
PY
done
