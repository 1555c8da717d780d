#!/bin/bash
# Counter passes (each in its own run) + kernel trace of the VLAD kernels (tools/dbg/vlad_time.py = bench_extras.cfg5 without the end-to-end run).
#   usage (GPU box, through gpurun): tools/pmc_vlad.sh <tag>
set -u
TAG=${1:-vlad_pmc}
KRE='k_vlad|k_assign_gmin16'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT" /tmp/prof_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/kt -o kt -- python tools/dbg/vlad_time.py > "$OUT/trace.log" 2>&1
python tools/rocprof_summary.py /tmp/prof_$TAG/kt/kt_results.db 12 > "$OUT/kernel_stats.txt" 2>&1
i=0
for P in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
         "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  d=/tmp/prof_$TAG/pmc_$i
  mkdir -p $d
  timeout 300 rocprofv3 --pmc $P --kernel-include-regex "$KRE" --output-format csv -d $d -o pmc -- python tools/dbg/vlad_time.py > $d/run.log 2>&1
done
python tools/pmc_summary.py "/tmp/prof_$TAG/pmc_*" "$KRE" > "$OUT/pmc_kernels.txt" 2>&1
head -14 "$OUT/kernel_stats.txt"; grep -v "^#" "$OUT/pmc_kernels.txt"
