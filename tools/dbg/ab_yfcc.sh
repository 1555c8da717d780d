#!/bin/bash
# same-box A/B on the yfcc object (95 M x 1024-d, 64 x 256): tools/dbg/ab_yfcc.sh outdir name1 name2 ...  ("base" = the in-tree library)
# YFCC_ARGS: arguments of tests/bench_yfcc.py (default: w = 64 only, no oracle gate)
out=gpurun_out/$1; shift
mkdir -p $out
for v in "$@"; do
  if [ "$v" = base ]; then unset MMIDX_LIB; else export MMIDX_LIB=$PWD/multimedia-indexing_amd/csrc/ab/libmmidx_$v.so; fi
  timeout 600 python tests/bench_yfcc.py ${YFCC_ARGS:---w 64 --parity 0} > $out/$v.json 2> $out/$v.log
  python - <<PY
import json
try:
    j = json.loads(open("$out/$v.json").read().strip().splitlines()[-1])
    for w, o in j.items():
        if isinstance(o, dict) and "queries_per_s" in o:
            print("$v", w, "q/s", o["queries_per_s"], "ms", o.get("ms_per_step"), o.get("stage_ms_per_step"), "verified/q", o.get("verified_codes_per_query"), "parity", o.get("parity"))
except Exception as e:
    print("$v failed", e)
PY
done
