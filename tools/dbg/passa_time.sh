#!/bin/bash
# prints the average duration of every k_scan* kernel of a short bench run under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
rm -rf /tmp/pt_$tag; mkdir -p /tmp/pt_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pt_$tag -o kt -- python bench.py --steps 5 --warmup 1 --no-cpu --gt 0 --exhaustive-steps 0 "$@" > /tmp/pt_$tag/log 2>&1
echo "== $tag $@"
python tools/rocprof_summary.py /tmp/pt_$tag/kt_results.db 60 2>&1 | grep -E "k_scan|k_merge|k_coarse|k_pair" | cut -c1-110
