#!/bin/bash
# kernel-trace statistics of a few kernels for one or more library variants:  tools/dbg/kt.sh '<kernel regex>' <bench args...> -- lib1 lib2 ...
KRE=$1; shift
ARGS=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do ARGS+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for L in "$@"; do
  d=/tmp/kt_$(basename $L .so); rm -rf $d
  MMIDX_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats -d $d -o kt -- python bench.py "${ARGS[@]}" > $d.log 2>&1
  echo "== $L: $(grep -o '"value": [0-9.]*' $d.log | head -1)"
  python tools/rocprof_summary.py $d/kt_results.db 60 2>/dev/null | grep -E "$KRE" | cut -c1-150
done
