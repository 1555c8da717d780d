#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/vs; mkdir -p /tmp/vs
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/vs -o kt -- python tools/bench_virtual_shards.py --shards 8 --steps 3 > /tmp/vs/log 2>&1
tail -1 /tmp/vs/log
python tools/rocprof_summary.py /tmp/vs/kt_results.db 60 2>&1 | grep -E "k_|copyBuffer|fillBuffer" | grep -v "k_assign\|k_encode\|k_place" | cut -c1-120
