#!/bin/bash
# timeline of the last single-query device calls: tools/dbg/lat_trace.sh [extra bench_latency args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/lt; mkdir -p /tmp/lt
timeout 600 rocprofv3 --kernel-trace -d /tmp/lt -o kt -- python tools/bench_latency.py --n 20000000 --sizes 1 --threads 1 --reps 5 "$@" > /tmp/lt/log 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("/tmp/lt/*results.db")[0])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
ni, si, ei = cols.index("name"), cols.index("start"), cols.index("end")
rows = [r for r in cur.execute("select * from kernels order by start") if "at::" not in str(r[ni]) and "rocprim" not in str(r[ni])]
# find the device-path calls: sequences starting with k_split_bf16; print the 3rd one from the first nq=1 group
starts = [i - 1 for i, r in enumerate(rows) if "k_coarse_gmin16" in str(r[ni])]
i0 = starts[4]; i1 = starts[5]
t0 = rows[i0][si]
busy = 0
for r in rows[i0:i1]:
    busy += r[ei] - r[si]
    print(f"{(r[si]-t0)/1e3:9.1f} us  dur {(r[ei]-r[si])/1e3:8.1f} us  {str(r[ni])[:60]}")
print("span %.1f us, kernels busy %.1f us, launches %d" % ((rows[i1-1][ei]-t0)/1e3, busy/1e3, i1-i0))
PY
