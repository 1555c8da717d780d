#!/bin/bash
# Round-3 profiles, second half of the round (K3s in front of pass B, early-out in K3g's scan): kernel traces of the default
# workloads and of the yfcc object, and counter passes of k_pair_smin_mfma on the yfcc object.  Text summaries only.
#   usage (on the GPU box, through gpurun): tools/profile_r03b.sh <tag>
set -u
TAG=${1:-r03b}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT" /tmp/prof_$TAG
LIGHT="--extras 0 --spread-steps 0 --other-configs 0 --exhaustive-steps 0"
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/kt -o kt -- python bench.py --steps 10 --warmup 2 $LIGHT > "$OUT/bench_kernel_trace.log" 2>&1
python tools/rocprof_summary.py /tmp/prof_$TAG/kt/kt_results.db 45 > "$OUT/kernel_stats.txt" 2>&1
grep '"metric"' "$OUT/bench_kernel_trace.log" > "$OUT/bench.json"
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/ky -o ky -- python tests/bench_yfcc.py --parity 0 > "$OUT/yfcc.json" 2> "$OUT/yfcc_trace.log"
python tools/rocprof_summary.py /tmp/prof_$TAG/ky/ky_results.db 30 > "$OUT/yfcc_kernel_stats.txt" 2>&1
KRE='k_pair_smin|k_scan_grp|k_scan_hist'
i=0
for P in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" \
         "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  d=/tmp/prof_$TAG/pmc_y_$i
  mkdir -p $d
  timeout 700 rocprofv3 --pmc $P --kernel-include-regex "$KRE" --output-format csv -d $d -o pmc -- python tests/bench_yfcc.py --w 64 --parity 0 > $d/bench.log 2>&1
done
python tools/pmc_summary.py "/tmp/prof_$TAG/pmc_y_*" "$KRE" > "$OUT/yfcc_pmc_kernels.txt" 2>&1
ls -la "$OUT"
