#!/bin/bash
# same-box A/B of several builds on the headline: tools/dbg/ab_many.sh <lib|default> ... (each twice, interleaved)
F="--extras 0 --other-configs 0 --exhaustive-steps 0 --steps 8 --spread-steps 0 --hard-steps 0 --no-cpu --gt 0"
for rep in 1 2; do
for L in "$@"; do
  if [ "$L" = default ]; then unset MMIDX_LIB; else export MMIDX_LIB=$PWD/multimedia-indexing_amd/csrc/ab/$L; fi
  python bench.py $F 2>/dev/null > /tmp/ab.json
  python - "$L" <<'PY'
import json, sys
j = json.load(open("/tmp/ab.json"))
print(sys.argv[1], "headline", j["value"], j["ms_per_step"], "pass A ms", j["roofline"]["avg_launch_ms"], "parity", (j.get("parity") or {}).get("ids_match"))
PY
done
done
