#!/bin/bash
# Round-4 profiles (K3m: pass B as a certified fp16-MFMA lower bound): kernel trace of the default workloads with the launches of
# every kernel clustered per workload, and counter passes of k_scan_mfma / k_mfma_verify / k_scan_hist on the hard and on the spread
# workload, each in its own run.  Text summaries only.     usage (on the GPU box, through gpurun): tools/profile_r04.sh <tag>
set -u
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT" /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/kt -o kt -- python bench.py --steps 10 --warmup 2 --extras 0 --exhaustive-steps 0 > "$OUT/bench_kernel_trace.log" 2>&1
python tools/rocprof_summary.py /tmp/prof_$TAG/kt/kt_results.db 45 > "$OUT/kernel_stats.txt" 2>&1
python tools/kernel_calls.py /tmp/prof_$TAG/kt/kt_results.db 'k_scan_mfma|k_mfma_verify|k_mfma_redo|k_scan_hist|k_group_build|k_scan_filt|k_merge|k_pair|k_coarse|k_split' > "$OUT/kernel_calls.txt" 2>&1
python tools/timeline_steps.py /tmp/prof_$TAG/kt/kt_results.db k_scan_hist 12 3 > "$OUT/timeline.txt" 2>&1
grep '"metric"' "$OUT/bench_kernel_trace.log" > "$OUT/bench.json"
KRE='k_scan_mfma|k_mfma_verify|k_scan_hist'
for W in hard spread; do
  if [ $W = hard ]; then ARGS="--sigma 1.0 --hard-steps 0 --spread-steps 0"; else ARGS="--hard-steps 0 --spread-steps 3"; fi
  i=0
  for P in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
    i=$((i+1))
    d=/tmp/prof_$TAG/pmc_${W}_$i
    mkdir -p $d
    timeout 700 rocprofv3 --pmc $P --kernel-include-regex "$KRE" --output-format csv -d $d -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu --gt 0 --extras 0 --other-configs 0 --exhaustive-steps 0 $ARGS > $d/bench.log 2>&1
  done
  python tools/pmc_summary.py "/tmp/prof_$TAG/pmc_${W}_*" "$KRE" > "$OUT/${W}_pmc_kernels.txt" 2>&1
done
ls -la "$OUT"
