#!/bin/bash
# counter passes of a few kernels for the default library:  tools/dbg/pmc1.sh <name> '<kernel regex>' <bench args...>
NAME=$1; KRE=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PMCS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" \
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" \
      "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA")
i=0; rm -rf /tmp/pmc1_${NAME}_*
for P in "${PMCS[@]}"; do
  i=$((i+1)); d=/tmp/pmc1_${NAME}_$i; mkdir -p $d
  timeout 600 rocprofv3 --pmc $P --kernel-include-regex "$KRE" --output-format csv -d $d -o pmc -- python bench.py "$@" > $d/run.log 2>&1
done
mkdir -p gpurun_out/dbg
python tools/pmc_summary.py "/tmp/pmc1_${NAME}_*" "$KRE" > gpurun_out/dbg/pmc_$NAME.txt 2>&1
grep -v "^#" gpurun_out/dbg/pmc_$NAME.txt | cut -c1-25,60-125
