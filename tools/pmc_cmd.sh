#!/bin/bash
# PMC passes over an arbitrary command: tools/pmc_cmd.sh <tag> <kernel-regex> <command ...>   (summary -> gpurun_out/<tag>/pmc_kernels.txt)
set -u
TAG=$1; KRE=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT" /tmp/prof_$TAG
i=0
for P in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
         "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  d=/tmp/prof_$TAG/pmc_$i
  mkdir -p $d
  timeout 900 rocprofv3 --pmc $P --kernel-include-regex "$KRE" --output-format csv -d $d -o pmc -- "$@" > $d/cmd.log 2>&1
  tail -2 $d/cmd.log
done
python tools/pmc_summary.py "/tmp/prof_$TAG/pmc_*" "$KRE" > "$OUT/pmc_kernels.txt" 2>&1
cat "$OUT/pmc_kernels.txt"
