#!/bin/bash
# K3m iteration loop on the GPU box: tools/dbg/mf_bench.sh outdir [bench args]: the K3m tests, then headline / hard / spread / cfg1-3
out=gpurun_out/$1; shift
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mfma or union or smin or golden" 2>&1 | tail -5 > $out/tests.log
tail -3 $out/tests.log
timeout 900 python bench.py --no-cpu --gt 0 --exhaustive-steps 0 --other-configs 1 --extras 0 --spread-steps 6 --hard-steps 10 "$@" > $out/bench.json 2> $out/bench.err
python - <<PY
import json
try:
    j = json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
    print("headline", round(j["value"]), j["ms_per_step"])
    for k in ("hard", "spread"):
        h = j.get(k)
        if h: print(k, round(h["value"]), h["ms_per_step"], h["stage_ms_per_step"], "verified/q", h.get("verified_codes_per_query"), "surv/q", h.get("mfma_survivors_per_query"), "redo/step", h.get("mfma_redo_queries_per_step"), h.get("parity"))
    oc = j.get("other_configs") or {}
    for k, v in oc.items():
        if isinstance(v, dict) and "qps_gpu" in v: print(k, round(v["qps_gpu"]), v.get("ids_match"), v.get("max_abs_ddist"))
except Exception as e:
    print("failed", e); print(open("$out/bench.err").read()[-2000:])
PY
