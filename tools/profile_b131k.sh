#!/bin/bash
# Kernel trace of the batch-131072 single-GPU step (K3ma pass A): per-kernel statistics and one steady-state step's timeline.
#   usage (GPU box, through gpurun): tools/profile_b131k.sh <tag> [extra bench.py arguments]
set -u
TAG=${1:-b131k}
shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT" /tmp/prof_$TAG
BARGS="--batch 131072 --nbatches 1 --steps 6 --warmup 2 --settle 3 --hard-steps 0 --spread-steps 0 --other-configs 0 --extras 0 --yfcc-n 0 --cfg5-images 0 --exhaustive-steps 0 --no-cpu --gt 0 $*"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/kt -o kt -- python bench.py $BARGS > "$OUT/bench_kernel_trace.log" 2>&1
python tools/rocprof_summary.py /tmp/prof_$TAG/kt/kt_results.db 40 > "$OUT/kernel_stats.txt" 2>&1
python tools/timeline_steps.py /tmp/prof_$TAG/kt/kt_results.db k_a1_pair_count 8 1 > "$OUT/timeline.txt" 2>&1
grep '"metric"' "$OUT/bench_kernel_trace.log" > "$OUT/bench.json"
head -45 "$OUT/kernel_stats.txt"
