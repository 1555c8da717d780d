#!/bin/bash
# A/B of coarse-stage variants inside one box: tools/dbg/coarse_ab.sh
cd "$GRAFT_REPO_ROOT"
for v in base notouch base notouch; do
  if [ $v = base ]; then unset MMIDX_LIB; else export MMIDX_LIB=$PWD/tools/dbg/lib_$v.so; fi
  echo "== $v"; bash tools/dbg/coarse_time.sh $v 2>&1 | grep -E "gmin16"
done
unset MMIDX_LIB
for o in 0 1 0 1; do python bench.py --no-cpu --exhaustive-steps 0 --gt 0 --opt coarse_nodma=$o 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nodma=$o', d['value'], d['ms_per_step'], d['roofline']['coarse_ms_per_step'])"; done
