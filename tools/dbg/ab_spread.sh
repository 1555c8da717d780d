#!/bin/bash
# same-box A/B on the spread object and cfg2 (flat PQ): tools/dbg/ab_spread.sh outdir name1 name2 ...  ("base" = the in-tree library)
out=gpurun_out/$1; shift
mkdir -p $out
for v in "$@"; do
  if [ "$v" = base ]; then unset MMIDX_LIB; else export MMIDX_LIB=$PWD/multimedia-indexing_amd/csrc/ab/libmmidx_$v.so; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --gt 0 --exhaustive-steps 0 --other-configs 1 --extras 0 --spread-steps 6 --hard-steps 0 > $out/$v.json 2> $out/$v.log
  python - <<PY
import json
try:
    j = json.loads(open("$out/$v.json").read().strip().splitlines()[-1])
    sp = j["spread"]; oc = j.get("other_configs") or {}
    print("$v", "spread", sp["value"], sp["ms_per_step"], "verified/q", sp.get("verified_codes_per_query"), "parity", sp.get("parity"),
          "| cfg2", {k: v for k, v in oc.get("cfg2_pq_adc_1M", {}).items() if k in ("gpu_qps", "qps", "queries_per_s", "parity")})
except Exception as e:
    print("$v failed", e)
PY
done
