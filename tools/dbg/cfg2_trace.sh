#!/bin/bash
# kernel trace of cfg2 (flat PQ) alone: tools/dbg/cfg2_trace.sh outdir [option=value ...]
out=gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out /tmp/prof_c2
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2/kt -o kt -- python tools/dbg/cfg2_time.py "$@" > $out/cfg2.log 2>&1
tail -3 $out/cfg2.log
python tools/kernel_calls.py /tmp/prof_c2/kt/kt_results.db 'k_scan_mfma|k_mfma|k_scan_hist|k_scan<|k_group_build|k_scan_filt|k_merge|k_flat|k_scan_grp' > $out/cfg2_calls.txt 2>&1
cat $out/cfg2_calls.txt
