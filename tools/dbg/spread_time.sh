#!/bin/bash
# spread workload timing for A/B builds (results may be wrong for timing builds: no parity here): tools/dbg/spread_time.sh lib...
for L in "$@"; do
  if [ "$L" = default ]; then unset MMIDX_LIB; else export MMIDX_LIB=$PWD/multimedia-indexing_amd/csrc/ab/$L; fi
  python bench.py --extras 0 --other-configs 0 --exhaustive-steps 0 --steps 3 --hard-steps 0 --no-cpu --gt 0 $BOPT 2>/dev/null > /tmp/s.json
  python - "$L" <<'PY'
import json, sys
j = json.load(open("/tmp/s.json")); s = j["spread"]
print(sys.argv[1], "spread", s["value"], s["ms_per_step"], s["verified_codes_per_query"], s["stage_ms_per_step"])
PY
done
