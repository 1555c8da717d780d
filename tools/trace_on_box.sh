#!/bin/bash
# kernel trace only (no PMC): tools/trace_on_box.sh <tag> [bench args...]
set -u
TAG=${1:-r01x}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT" /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/kt -o kt -- python bench.py --steps 10 --warmup 2 "$@" > "$OUT/bench_kernel_trace.log" 2>&1
python tools/rocprof_summary.py /tmp/prof_$TAG/kt/kt_results.db 40 > "$OUT/kernel_stats.txt" 2>&1
python - > "$OUT/timeline.txt" 2>&1 <<PY
import sqlite3
db = sqlite3.connect("/tmp/prof_$TAG/kt/kt_results.db")
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
rows = list(cur.execute("select * from kernels order by start desc limit ${TL_ROWS:-40}"))
ni = cols.index("name") if "name" in cols else None
si, ei = cols.index("start"), cols.index("end")
rows = rows[::-1]
t0 = rows[0][si]
for r in rows:
    print(f"{(r[si]-t0)/1e3:10.1f} us  dur {(r[ei]-r[si])/1e3:9.1f} us  {str(r[ni])[:70]}")
PY
grep '"metric"' "$OUT/bench_kernel_trace.log" > "$OUT/bench.json"
