#!/bin/bash
# A/B of K3g's union histogram on the hard workload: tools/dbg/ab_union.sh
F="--extras 0 --other-configs 0 --exhaustive-steps 0 --steps 5 --spread-steps 0 --no-cpu --gt 0"
for o in 0 1 0 1; do
  python bench.py $F --opt no_union=$o 2>/dev/null > /tmp/ab_$o.json
  python - "$o" <<'PY'
import json, sys
o = sys.argv[1]
j = json.load(open(f"/tmp/ab_{o}.json"))
h = j["hard"]
print("no_union", o, "headline", j["value"], "hard", h["value"], h["stage_ms_per_step"])
PY
done
