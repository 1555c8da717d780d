#!/bin/bash
# Round-6 profiles.  usage (GPU box, through gpurun): tools/profile_r06.sh <part> <tag>
#   part 1: the default bench untraced (JSON + stderr summary) and under the kernel trace (kernel stats, calls per workload, headline timeline)
#   part 2: counter passes of the headline step (k_scan_q: K3q, round 6) and of the batch-131072 step (K3q as well) + its kernel trace
#   part 3: counter passes of the VLAD kernel and of the yfcc legs (k_scan_hist<64,..>, k_scan_mfma_kc2)
set -u
PART=${1:-1}
TAG=${2:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT" /tmp/prof_$TAG
PMCS=("SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
      "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY")
pmc() {  # pmc <name> <kernel regex> <command...>
  local name=$1 kre=$2; shift 2
  local i=0
  for P in "${PMCS[@]}"; do
    i=$((i+1)); local d=/tmp/prof_$TAG/pmc_${name}_$i; mkdir -p $d
    timeout 600 rocprofv3 --pmc $P --kernel-include-regex "$kre" --output-format csv -d $d -o pmc -- "$@" > $d/run.log 2>&1
  done
  python tools/pmc_summary.py "/tmp/prof_$TAG/pmc_${name}_*" "$kre" > "$OUT/${name}_pmc_kernels.txt" 2>&1
}
SIDE0="--hard-steps 0 --spread-steps 0 --other-configs 0 --extras 0 --yfcc-n 0 --cfg5-images 0 --exhaustive-steps 0 --dry-run-shards 0 --no-cpu --gt 0"
if [ "$PART" = 1 ]; then
  timeout 900 python bench.py > "$OUT/bench_untraced.json" 2> "$OUT/bench_untraced.log"
  grep "summary:" "$OUT/bench_untraced.log" > "$OUT/bench_summary.txt"; cp bench_extra.json "$OUT/bench_extra.json" 2>/dev/null
  timeout 1200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/kt -o kt -- python bench.py --steps 10 --warmup 2 --extras 0 --exhaustive-steps 0 --dry-run-shards 0 --yfcc-n 0 > "$OUT/bench_kernel_trace.log" 2>&1
  python tools/rocprof_summary.py /tmp/prof_$TAG/kt/kt_results.db 50 > "$OUT/kernel_stats.txt" 2>&1
  python tools/kernel_calls.py /tmp/prof_$TAG/kt/kt_results.db 'k_scan_q|k_scan_mfma|k_mfma_verify|k_mfma_redo|k_scan_hist|k_group_build|k_scan_filt|k_merge|k_pair|k_coarse|k_split|k_a1_' > "$OUT/kernel_calls.txt" 2>&1
  python tools/timeline_steps.py /tmp/prof_$TAG/kt/kt_results.db k_scan_q 12 2 > "$OUT/timeline.txt" 2>&1
  grep '"metric"' "$OUT/bench_kernel_trace.log" > "$OUT/bench.json"
  tail -3 "$OUT/bench_summary.txt" | cut -c1-1500
elif [ "$PART" = 2 ]; then
  pmc headline 'k_scan_q|k_scan_hist|k_coarse_front_sel|k_coarse_gmin16' python bench.py --steps 3 --warmup 1 --settle 2 --big-batch 0 $SIDE0
  pmc b131k 'k_scan_q|k_a1_pair|k_coarse_front_sel|k_coarse_gmin16|k_merge' python bench.py --batch 131072 --nbatches 1 --steps 3 --warmup 1 --settle 2 $SIDE0
  tools/profile_b131k.sh ${TAG}_b131k --dry-run-shards 0 > /dev/null 2>&1
  cp gpurun_out/${TAG}_b131k/kernel_stats.txt "$OUT/b131k_kernel_stats.txt"; cp gpurun_out/${TAG}_b131k/timeline.txt "$OUT/b131k_timeline.txt"; cp gpurun_out/${TAG}_b131k/bench.json "$OUT/b131k_bench.json"
  grep -E "FETCH_SIZE|WRITE_SIZE" "$OUT"/*_pmc_kernels.txt | cut -c1-200
else
  pmc vlad 'k_vlad|k_assign_gmin16' python tools/dbg/vlad_time.py
  pmc yfcc 'k_scan_hist|k_scan_mfma_kc|k_mfma_verify|k_pair_smin' python tests/bench_yfcc.py --parity 0
  grep -E "FETCH_SIZE|WRITE_SIZE" "$OUT"/*_pmc_kernels.txt | cut -c1-200
fi
ls -la "$OUT"
