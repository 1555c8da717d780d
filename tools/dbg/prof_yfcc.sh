#!/bin/bash
# kernel trace of the yfcc object alone (w = 64): tools/dbg/prof_yfcc.sh <tag>
TAG=${1:-yf}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/kt -o kt -- python tests/bench_yfcc.py --w 64 --parity 0 > gpurun_out/$TAG/bench.log 2>&1
python tools/rocprof_summary.py /tmp/prof_$TAG/kt/kt_results.db 60 | grep -v "at::native\|rocprim\|Cijk\|rocclr" > gpurun_out/$TAG/kernel_stats.txt 2>&1
cat gpurun_out/$TAG/kernel_stats.txt | cut -c1-150
