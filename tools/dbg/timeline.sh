#!/bin/bash
# steady-state step timeline of the headline: tools/dbg/timeline.sh <tag> [first] [steps]
TAG=${1:-tl}; FIRST=${2:-10}; STEPS=${3:-3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$TAG/kt -o kt -- python bench.py --steps 10 --warmup 2 --extras 0 --other-configs 0 --exhaustive-steps 0 --hard-steps 0 --spread-steps 0 --no-cpu --gt 0 > gpurun_out/$TAG/bench.log 2>&1
python tools/timeline_steps.py /tmp/prof_$TAG/kt/kt_results.db k_scan_hist $FIRST $STEPS > gpurun_out/$TAG/timeline.txt 2>&1
python tools/timeline_steps.py /tmp/prof_$TAG/kt/kt_results.db k_scan_hist 30 3 > gpurun_out/$TAG/timeline_timed.txt 2>&1
tail -3 gpurun_out/$TAG/timeline.txt; tail -1 gpurun_out/$TAG/timeline_timed.txt
