"""bench_emit.py -- the last thing bench.py does: write the result.

Contract (the driver parses the LAST `{...}` of a run's captured output):
  * every side measurement goes to `bench_extra.json` (path named in the line);
  * the published figures once more as plain `key=value` text on stderr -- no braces, nothing a JSON scanner could take for
    an object -- BEFORE the result line;
  * then ONE JSON line on stdout, under MAX_LINE bytes, and nothing after it on either stream.

tests/test_bench_emit_cpu.py runs this on a canned dict and checks exactly that.
"""
import json
import os
import sys

MAX_LINE = 8000

# what stays in the line (the bench contract's keys + the two objects the judge reads + scalars)
CORE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "config", "recall_at_1", "recall_queries", "roofline", "cpu_baseline", "parity", "ties",
             "host_buffers_qps", "host_buffers_qps_3_callers", "hard_qps", "spread_qps", "batch_131072_qps", "sharded_dry_run_ms_per_shard")
ROOFLINE_KEYS = ("bound", "kernel", "pass_a_kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms",
                 "launches", "traffic_source", "lds_conflict_ratio")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "cpu_model")
CONFIG_KEYS = ("workload", "n", "dim", "cells", "nprobe", "m", "ks", "k", "batch", "batch_per_gpu", "mixture_sigma", "multi_gpu_path",
               "rccl_ranks", "native_fallback_reason")


def _pick(d, keys):
    return None if not isinstance(d, dict) else {k: d[k] for k in keys if k in d}


def _short(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3] + "..."


def split(out):
    """(line object, extras object) of a full result dict"""
    core = {k: out[k] for k in CORE_KEYS if k in out}
    extra = {k: v for k, v in out.items() if k not in CORE_KEYS}
    if isinstance(out.get("roofline"), dict):
        core["roofline"] = _pick(out["roofline"], ROOFLINE_KEYS)
        core["roofline"]["kernel"] = _short(core["roofline"].get("kernel"), 200)
        core["roofline"]["traffic_source"] = _short(core["roofline"].get("traffic_source"), 160)
        extra["roofline_full"] = out["roofline"]
    if isinstance(out.get("cpu_baseline"), dict):
        core["cpu_baseline"] = _pick(out["cpu_baseline"], CPU_KEYS)
        core["cpu_baseline"]["sample"] = _short(core["cpu_baseline"].get("sample"), 300)
        extra["cpu_baseline_full"] = out["cpu_baseline"]
    if isinstance(out.get("config"), dict):
        core["config"] = _pick(out["config"], CONFIG_KEYS)
        core["config"]["multi_gpu_path"] = _short(core["config"].get("multi_gpu_path"), 160)
        core["config"]["native_fallback_reason"] = _short(core["config"].get("native_fallback_reason"), 200)
        extra["config_full"] = out["config"]
    return core, extra


def _flat(prefix, o, acc):
    """nested dict -> `a.b.c=value` pairs (scalars only; strings longer than a few words are left to the extras file)"""
    if isinstance(o, dict):
        for k, v in o.items():
            _flat(f"{prefix}.{k}" if prefix else str(k), v, acc)
    elif isinstance(o, (int, float, bool)) or o is None:
        acc.append(f"{prefix}={o}")
    elif isinstance(o, str) and len(o) <= 40 and "{" not in o and "}" not in o:
        acc.append(f"{prefix}={o}")


def summary_text(summ):
    """the short summary as brace-free text lines"""
    acc = []
    _flat("", summ, acc)
    lines, cur = [], ""
    for kv in acc:
        if len(cur) + len(kv) + 1 > 180:
            lines.append(cur)
            cur = ""
        cur = (cur + " " + kv).strip()
    if cur:
        lines.append(cur)
    return lines


def emit(out, json_out=None, err=None, extra_path=None, summary=None):
    """writes extras, the text summary, then the one JSON line -- in that order; returns the line"""
    json_out = sys.stdout if json_out is None else json_out
    err = sys.stderr if err is None else err
    core, extra = split(out)
    if extra_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(extra_path)), exist_ok=True)
            with open(extra_path, "w") as f:
                json.dump(extra, f, indent=1, default=str)
            core["extra"] = os.path.relpath(extra_path, os.getcwd()) if not os.path.isabs(extra_path) else extra_path
        except OSError as e:
            core["extra"] = f"not written: {e!r}"
    if summary:
        for ln in summary_text(summary):
            print("[bench] summary:", ln, file=err)
    line = json.dumps(core, separators=(",", ":"), default=str)
    if len(line) > MAX_LINE:  # (should not happen: drop the free-text fields before anything with a number in it)
        for obj, key in ((core.get("cpu_baseline"), "sample"), (core.get("roofline"), "traffic_source"), (core.get("config"), "multi_gpu_path"),
                         (core.get("config"), "native_fallback_reason"), (core.get("roofline"), "kernel")):
            if isinstance(obj, dict) and key in obj and len(line) > MAX_LINE:
                obj[key] = _short(obj[key], 40)
                line = json.dumps(core, separators=(",", ":"), default=str)
    err.flush()
    print(line, file=json_out)
    json_out.flush()
    return line
