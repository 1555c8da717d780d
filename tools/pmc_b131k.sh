#!/bin/bash
# Counter passes (each in its own run, no tracing domains) of K3ma's kernels on the batch-131072 single-GPU step.
#   usage (GPU box, through gpurun): tools/pmc_b131k.sh <tag> [kernel regex]
set -u
TAG=${1:-b131k_pmc}
KRE=${2:-'k_scan_mfma|k_a1_verify|k_a1_select|k_coarse_front_sel|k_coarse_gmin16'}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT" /tmp/prof_$TAG
BARGS="--batch 131072 --nbatches 1 --steps 3 --warmup 1 --settle 2 --hard-steps 0 --spread-steps 0 --other-configs 0 --extras 0 --yfcc-n 0 --cfg5-images 0 --exhaustive-steps 0 --no-cpu --gt 0"
i=0
for P in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
         "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  d=/tmp/prof_$TAG/pmc_$i
  mkdir -p $d
  timeout 400 rocprofv3 --pmc $P --kernel-include-regex "$KRE" --output-format csv -d $d -o pmc -- python bench.py $BARGS > $d/bench.log 2>&1
done
python tools/pmc_summary.py "/tmp/prof_$TAG/pmc_*" "$KRE" > "$OUT/pmc_kernels.txt" 2>&1
cat "$OUT/pmc_kernels.txt"
