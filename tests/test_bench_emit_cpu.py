"""The bench contract's last step: the result line is the LAST thing a run writes to either stream, is short, and carries `roofline`
and `cpu_baseline` (round 5's record was unparseable: a second JSON object followed the result line, which had grown to 21 KB).
Runs the emitter (tests/bench_emit.py) in a subprocess on a full result dict of round 5 and reads stdout + stderr the way a driver
that keeps the tail of the captured output does."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANNED = os.path.join(ROOT, "profiles", "r05c_bench_untraced.json")

DRIVER = r"""
import json, os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import bench_emit
out = json.load(open(sys.argv[2]))
out["host_buffers_qps"] = out["host_path"]["nq16384"]["queries_per_s"]
out["hard_qps"] = out["hard"]["value"]
out["ties"] = {"duplicate_code_rate": 0.0, "queries_with_tie_at_k": 0}
summ = {"headline_Mqps": out["value"] / 1e6, "passA_frac": out["roofline"]["frac"], "hard": {"Mqps": 5.1, "parity": True}, "note": "{not json}"}
# the stdout / stderr split of bench.py: fd 1 -> stderr for everything but the line
json_out = os.fdopen(os.dup(1), "w")
os.dup2(2, 1)
print("[bench] chatter with braces {'a': 1}", file=sys.stderr)
bench_emit.emit(out, json_out=json_out, err=sys.stderr, extra_path=sys.argv[3], summary=summ)
"""


def last_json_object(text):
    """what the driver does: the last line that parses as a JSON object"""
    lines = [ln for ln in text.splitlines() if ln.strip()]
    return lines[-1]


def test_result_line_is_last_short_and_complete(tmp_path):
    extra = tmp_path / "bench_extra.json"
    pr = subprocess.run([sys.executable, "-c", DRIVER, ROOT, CANNED, str(extra)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                        timeout=120)
    assert pr.returncode == 0, pr.stdout
    last = last_json_object(pr.stdout)
    assert len(last) < 8000
    d = json.loads(last)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "parity", "host_buffers_qps", "hard_qps", "ties"):
        assert key in d, key
    assert d["roofline"]["bound"] == "hbm" and 0.0 < d["roofline"]["frac"] < 1.0
    for key in ("achieved", "peak", "unit", "traffic"):
        assert key in d["roofline"], key
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["kind"] in ("port", "reference")
    assert "sample" in d["cpu_baseline"]
    assert "workload" in d["config"] and "model" not in d["config"]
    # nothing after the line that a scanner could take for an object, and no other line of the output parses as a JSON object
    others = [ln for ln in pr.stdout.splitlines() if ln.strip()][:-1]
    for ln in others:
        if ln.lstrip().startswith("{"):
            try:
                json.loads(ln)
            except ValueError:
                continue
            raise AssertionError(f"a second JSON object in the output: {ln[:80]}")
    summ = [ln for ln in others if ln.startswith("[bench] summary:")]
    assert summ and all("{" not in ln and "}" not in ln for ln in summ)
    # the side measurements are in the extras file the line names
    ex = json.load(open(extra))
    for key in ("hard", "spread", "other_configs", "cfg5", "yfcc", "host_path", "batch_131072", "sharded_dry_run", "roofline_whole_search",
                "roofline_full", "cpu_baseline_full"):
        assert key in ex, key
    assert d["extra"] == str(extra)


def test_bench_py_writes_nothing_after_the_emitter():
    """bench.py: the emitter call is the last statement of run() and main() prints nothing after run()"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def run(cx, args, json_out"):src.index('if __name__ == "__main__":')]
    tail = body[body.index('importlib.import_module("bench_emit").emit('):]
    assert "print(" not in tail and "log(" not in tail
    assert src.count("json.dumps(out)") == 0  # (the only writer of the line is the emitter)
