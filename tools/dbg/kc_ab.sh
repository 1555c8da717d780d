#!/bin/bash
# same-box A/B of K3mk variants on the yfcc object: tools/dbg/kc_ab.sh outdir name1 name2 ... ("base" = the in-tree library; else tools/dbg/lib_<name>.so)
out=gpurun_out/$1; shift
mkdir -p $out
for v in "$@"; do
  if [ "$v" = base ]; then unset MMIDX_LIB; else export MMIDX_LIB=$PWD/tools/dbg/lib_$v.so; fi
  timeout 600 python tests/bench_yfcc.py ${YFCC_ARGS:---w 64 --parity 0 --steps 3} > $out/$v.json 2> $out/$v.log
  python - <<PY
import json
try:
    j = json.loads(open("$out/$v.json").read().strip().splitlines()[-1])
    for w, o in j.items():
        if isinstance(o, dict) and "queries_per_s" in o and "between" in w:
            print("$v", w, "q/s", o["queries_per_s"], "ms", o.get("ms_per_step"), o.get("pass_b"))
except Exception as e:
    print("$v failed", e)
PY
done
