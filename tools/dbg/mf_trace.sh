#!/bin/bash
# kernel trace of the hard + spread objects (K3m iteration): tools/dbg/mf_trace.sh outdir [bench args]
out=gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out /tmp/prof_mf
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_mf/kt -o kt -- python bench.py --no-cpu --gt 0 --exhaustive-steps 0 --other-configs 0 --extras 0 --spread-steps 6 --hard-steps 10 --steps 5 --warmup 1 "$@" > $out/trace_bench.log 2>&1
python tools/rocprof_summary.py /tmp/prof_mf/kt/kt_results.db 30 > $out/kernel_stats.txt 2>&1
python tools/kernel_calls.py /tmp/prof_mf/kt/kt_results.db 'k_scan_mfma|k_mfma_verify|k_mfma_redo|k_scan_hist|k_group_build|k_scan_filt|k_merge|k_pair' > $out/kernel_calls.txt 2>&1
cat $out/kernel_calls.txt
