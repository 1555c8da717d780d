cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "smin_prefilter or yfcc_shape" 2>&1 | tail -4
timeout 600 python tests/fuzz_parity.py 30 12 2>&1 | tail -3
for o in "smin_pre=-1"; do
  timeout 600 python tests/bench_yfcc.py --w 2 64 --parity 0 --opt $o > gpurun_out/y_$o.json 2> gpurun_out/y_$o.log
  python - <<PY
import json
j = json.loads(open("gpurun_out/y_$o.json").read().strip().splitlines()[-1])
for w in ("w2", "w64"):
    o = j[w]; print("$o", w, o["queries_per_s"], o["ms_per_step"], o["stage_ms_per_step"], o["far_pairs_scanned_per_query"])
PY
done
