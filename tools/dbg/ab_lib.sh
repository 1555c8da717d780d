#!/bin/bash
# same-box A/B of two builds on headline + hard: tools/dbg/ab_lib.sh <libA.so|default> <libB.so|default> [bench args]
A=$1; B=$2; shift; shift
F="--extras 0 --other-configs 0 --exhaustive-steps 0 --steps 5 --spread-steps 0 --no-cpu --gt 0 $*"
for L in $A $B $A $B; do
  if [ "$L" = default ]; then unset MMIDX_LIB; else export MMIDX_LIB=$PWD/multimedia-indexing_amd/csrc/ab/$L; fi
  python bench.py $F 2>/dev/null > /tmp/ab.json
  python - "$L" <<'PY'
import json, sys
j = json.load(open("/tmp/ab.json"))
h = j.get("hard") or {}
print(sys.argv[1], "headline", j["value"], j["ms_per_step"], "hard", h.get("value"), h.get("stage_ms_per_step"))
PY
done
