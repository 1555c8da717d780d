#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the committed counter summaries of a round (tools/profile_r05.sh parts 2 and 3):
    python tools/make_hbm_traffic.py r05
FETCH_SIZE / WRITE_SIZE come in KiB per dispatch; reads x 2: the gfx950 correction for streaming reads
(MI355X_MICROARCH.md, section HBM / rocprofv3).  bench.py, tests/bench_yfcc.py and tests/bench_extras.py read the result for
their `traffic` fields when the workload they run is the one recorded here."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(path):
    out = {}
    for ln in open(path):
        m = re.match(r"(FETCH_SIZE|WRITE_SIZE)\s+(.*?)\s+grid=\s*(\d+) n=\s*(\d+) mean=(\S+)\s+top n=\s*(\d+) mean=(\S+)", ln)
        if m:
            out.setdefault((m.group(1), m.group(2).strip()), []).append(
                {"grid": int(m.group(3)), "n": int(m.group(4)), "mean_kib": float(m.group(5)), "top_n": int(m.group(6)), "top_mean_kib": float(m.group(7))})
    return out


def pick(tab, counter, kernel, grid=None, top=False):
    best = None
    for (c, k), lst in tab.items():
        if c != counter or kernel not in k:
            continue
        for r in lst:
            if grid is not None and r["grid"] != grid:
                continue
            if best is None or r["mean_kib"] > best["mean_kib"]:
                best = r
    if best is None:
        return None
    return best["top_mean_kib" if top else "mean_kib"]


def main(tag):
    P = lambda name: os.path.join(ROOT, "profiles", "%s_%s_pmc_kernels.txt" % (tag, name))  # noqa: E731
    old = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    head = rows(P("headline"))
    hist = pick(head, "FETCH_SIZE", "k_scan_hist<16, 256, 256>")
    out = {
        "source": "profiles/%s_{headline,b131k,vlad,yfcc}_pmc_kernels.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, tools/profile_r05.sh parts 2-3 with this tag; "
                  "KiB x 1024, reads x 2: the gfx950 correction for streaming reads, MI355X_MICROARCH.md section HBM); hard / spread pass B: "
                  "profiles/r04_hard_pmc_kernels.txt, profiles/r04_spread_pmc_kernels.txt (K3m unchanged since)" % (tag,),
        "workload": old["workload"],
        "k_scan_hist_fetch_kib_per_step": hist,
        "k_scan_launches_per_step": 2,
        "hbm_bytes_per_step": int(hist * 1024 * 2 + 2 * 19.5 * 1024 * 2),
        "hbm_bytes_per_query": int((hist * 1024 * 2) / old["workload"]["batch"]),
        "k_coarse_front_sel_fetch_bytes_per_launch": int(pick(head, "FETCH_SIZE", "k_coarse_front_sel") * 2048),
        "k_coarse_gmin16_fetch_bytes_per_launch": int(pick(head, "FETCH_SIZE", "k_coarse_gmin16") * 2048),
        "k_coarse_gmin16_write_bytes_per_launch": int(pick(head, "WRITE_SIZE", "k_coarse_gmin16") * 1024),
    }
    for k in ("hard_pass_b_fetch_bytes_per_launch", "hard_pass_a_fetch_bytes_per_launch", "hard_note", "spread_pass_b_fetch_bytes_per_launch", "spread_note"):
        out[k] = old[k]
    b = rows(P("b131k"))
    s1 = pick(b, "FETCH_SIZE", "k_scan_mfma<4, 8, 1, 4>")
    s2 = pick(b, "FETCH_SIZE", "k_scan_mfma<4, 8, 2, 8>") + pick(b, "FETCH_SIZE", "k_scan_mfma<4, 8, 2, 4>")
    out["batch_131072"] = {
        "workload": dict(old["workload"], batch=131072),
        "sweep1_fetch_bytes_per_step": int(s1 * 2048), "sweep2_fetch_bytes_per_step": int(s2 * 2048),
        "sweeps_fetch_bytes_per_step": int((s1 + s2) * 2048),
        "sweeps_write_bytes_per_step": int((pick(b, "WRITE_SIZE", "k_scan_mfma<4, 8, 1, 4>") + pick(b, "WRITE_SIZE", "k_scan_mfma<4, 8, 2, 8>")) * 1024),
        "k_a1_verify_fetch_bytes_per_step": int(pick(b, "FETCH_SIZE", "k_a1_verify") * 2048),
        "k_a1_records_fetch_bytes_per_step": int(pick(b, "FETCH_SIZE", "k_a1_records") * 2048),
        "k_coarse_front_sel_fetch_bytes_per_step": int(pick(b, "FETCH_SIZE", "k_coarse_front_sel") * 2048),
        "k_merge_fetch_bytes_per_step": int(pick(b, "FETCH_SIZE", "k_merge") * 2048),
        "note": "bench.py --batch 131072 --nbatches 1 (K3ma pass A by the default gate); the cfg4 codes of all 8192 lists are 1.6 GB: each sweep reads them once "
                "plus the norms (4 B per code) and, sweep 2, its bitmap writes",
    }
    v = rows(P("vlad"))
    out["vlad"] = {"images": 20000, "descriptors": 9992493,
                   "k_vlad_fused_fetch_bytes_per_launch": int(pick(v, "FETCH_SIZE", "k_vlad_fused") * 2048),
                   "k_vlad_fused_write_bytes_per_launch": int(pick(v, "WRITE_SIZE", "k_vlad_fused") * 1024),
                   "note": "tests/bench_extras.py cfg5(): 20 k images of 200-800 SURF-64 descriptors, 128 centroids"}
    y = rows(P("yfcc"))
    out["yfcc"] = {"workload": [95213780, 1024, 64, 8192, 4096],
                   "k_scan_hist_fetch_bytes_per_launch": int(pick(y, "FETCH_SIZE", "k_scan_hist<64, 256, 1024>", grid=4194304) * 2048),
                   "k_scan_mfma_kc2_fetch_bytes_per_launch": int(pick(y, "FETCH_SIZE", "k_scan_mfma_kc2", top=True) * 2048),
                   "note": "tests/bench_yfcc.py defaults (n, D, m, cells, batch); k_scan_mfma_kc2: mean of the launches of the w = 64 between-clusters leg "
                           "(the `top` group of the summary: all 63 far pairs scanned)"}
    json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=2)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r05")
