#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
rm -rf /tmp/ct_$tag; mkdir -p /tmp/ct_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ct_$tag -o kt -- python tools/bench_coarse.py "$@" > /tmp/ct_$tag/log 2>&1
echo "== $tag $@"; tail -1 /tmp/ct_$tag/log
python tools/rocprof_summary.py /tmp/ct_$tag/kt_results.db 12 2>&1 | grep -E "k_" | cut -c1-120
