#!/usr/bin/env python3
"""Round 6: add K3q's counters to profiles/hbm_traffic.json (the other entries -- K3m pass B, VLAD, yfcc -- are unchanged kernels and
keep their round-4 / round-5 figures).
    python tools/make_hbm_traffic_r06.py r06
reads profiles/<tag>_headline_pmc_kernels.txt and profiles/<tag>_b131k_pmc_kernels.txt (tools/profile_r06.sh part 2; rocprofv3 --pmc
FETCH_SIZE / WRITE_SIZE in separate passes, KiB per dispatch).  Reads x 2: the gfx950 correction for streaming reads
(MI355X_MICROARCH.md, section HBM / rocprofv3) is applied by the readers (bench.py), which multiply fetch_kib by 2048."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_hbm_traffic import pick, rows  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    tj = json.load(open(path))
    head = rows(os.path.join(ROOT, "profiles", f"{tag}_headline_pmc_kernels.txt"))
    b = rows(os.path.join(ROOT, "profiles", f"{tag}_b131k_pmc_kernels.txt"))
    src = f"profiles/{tag}_headline_pmc_kernels.txt"
    tj["pass_a"] = {
        "K3q": {"kernel": "k_scan_q<16, 8>", "fetch_kib_per_step": pick(head, "FETCH_SIZE", "k_scan_q<16, 8>"),
                "write_kib_per_step": pick(head, "WRITE_SIZE", "k_scan_q<16, 8>"), "source": src},
        "K3h": {"kernel": "k_scan_hist<16, 256, 256>", "fetch_kib_per_step": tj["k_scan_hist_fetch_kib_per_step"], "write_kib_per_step": 0.0,
                "source": "profiles/r05b_headline_pmc_kernels.txt (option passa_q = 0)"},
    }
    kq = tj["pass_a"]["K3q"]
    tj["hbm_bytes_per_step"] = int(kq["fetch_kib_per_step"] * 2048 + kq["write_kib_per_step"] * 1024)
    tj["hbm_bytes_per_query"] = int(tj["hbm_bytes_per_step"] / tj["workload"]["batch"])
    tj.setdefault("batch_131072", {})["k_scan_q"] = {"fetch_kib_per_step": pick(b, "FETCH_SIZE", "k_scan_q<16, 8>"),
                                                      "write_kib_per_step": pick(b, "WRITE_SIZE", "k_scan_q<16, 8>"),
                                                      "source": f"profiles/{tag}_b131k_pmc_kernels.txt"}
    tj["source"] = (f"round 6: pass_a.K3q and batch_131072.k_scan_q from profiles/{tag}_{{headline,b131k}}_pmc_kernels.txt (tools/profile_r06.sh part 2); "
                    "everything else as recorded in round 5: " + tj["source"].split("everything else as recorded in round 5: ")[-1])
    json.dump(tj, open(path, "w"), indent=2)
    print(json.dumps({"pass_a": tj["pass_a"], "b131k": tj["batch_131072"]["k_scan_q"]}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r06")
