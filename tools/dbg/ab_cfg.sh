#!/bin/bash
# same-box A/B on BASELINE configs 1-3 (tests/bench_configs.py, parity-gated): tools/dbg/ab_cfg.sh name1 name2 ...
for v in "$@"; do
  if [ "$v" = base ]; then unset MMIDX_LIB; else export MMIDX_LIB=$PWD/multimedia-indexing_amd/csrc/ab/libmmidx_$v.so; fi
  timeout 600 python tests/bench_configs.py 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', {k:{kk:vv for kk,vv in v.items() if kk in ('qps_gpu','ids_match','qps_gpu_host_buffers')} if isinstance(v,dict) else v for k,v in j.items()})
"
done
