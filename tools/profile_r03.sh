#!/bin/bash
# Round-3 profiles, run on the GPU box through gpurun: kernel traces (headline + hard, yfcc) and PMC passes of the headline and of
# the hard workload alone (the hard workload = the headline path on mixture noise 1.0).  Text summaries only, under gpurun_out/<tag>/.
#   usage: tools/profile_r03.sh <tag>
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT" /tmp/prof_$TAG
LIGHT="--extras 0 --spread-steps 0 --other-configs 0 --exhaustive-steps 0"
# 1. kernel trace: headline + hard
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/kt -o kt -- python bench.py --steps 10 --warmup 2 $LIGHT > "$OUT/bench_kernel_trace.log" 2>&1
python tools/rocprof_summary.py /tmp/prof_$TAG/kt/kt_results.db 45 > "$OUT/kernel_stats.txt" 2>&1
grep '"metric"' "$OUT/bench_kernel_trace.log" > "$OUT/bench.json"
# 2. kernel trace: the reference's flagship shape
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/ky -o ky -- python tests/bench_yfcc.py --parity 0 > "$OUT/yfcc.json" 2> "$OUT/yfcc_trace.log"
python tools/rocprof_summary.py /tmp/prof_$TAG/ky/ky_results.db 30 > "$OUT/yfcc_kernel_stats.txt" 2>&1
# 3. PMC passes (counters only, separate runs): headline, then the hard workload alone
KRE='k_scan|k_coarse|k_merge'
for W in headline hard; do
  SIG=0.15; [ $W = hard ] && SIG=1.0
  i=0
  for P in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    d=/tmp/prof_$TAG/pmc_${W}_$i
    mkdir -p $d
    timeout 700 rocprofv3 --pmc $P --kernel-include-regex "$KRE" --output-format csv -d $d -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu --gt 0 --hard-steps 0 --sigma $SIG $LIGHT > $d/bench.log 2>&1
  done
  python tools/pmc_summary.py "/tmp/prof_$TAG/pmc_${W}_*" "$KRE" > "$OUT/${W}_pmc_kernels.txt" 2>&1
done
ls -la "$OUT"
