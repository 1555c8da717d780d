#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of bench.py, keep only text
# summaries under gpurun_out/<tag>/ (the raw .db / .csv files are too large to be copied back).
#   usage: tools/profile_on_box.sh <tag> [kernel-regex]
set -u
TAG=${1:-r01x}
KRE=${2:-k_scan|k_coarse|k_merge}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT" /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/kt -o kt -- python bench.py --steps 10 --warmup 2 > "$OUT/bench_kernel_trace.log" 2>&1
python tools/rocprof_summary.py /tmp/prof_$TAG/kt/kt_results.db 40 > "$OUT/kernel_stats.txt" 2>&1
grep '"metric"' "$OUT/bench_kernel_trace.log" > "$OUT/bench.json"
i=0
for P in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
         "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  d=/tmp/prof_$TAG/pmc_$i
  mkdir -p $d
  timeout 700 rocprofv3 --pmc $P --kernel-include-regex "$KRE" --output-format csv -d $d -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu --gt 0 > $d/bench.log 2>&1
done
python tools/pmc_summary.py "/tmp/prof_$TAG/pmc_*" "$KRE" > "$OUT/pmc_kernels.txt" 2>&1
ls -la "$OUT"
