"""profiles/hbm_traffic.json is what bench.py, tests/bench_yfcc.py and tests/bench_extras.py publish as `traffic`: the keys they read
must be there, the workloads they compare against must be the ones they run by default, and the counter summaries the file names
as its source must be committed next to it."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def test_hbm_traffic_json_has_what_the_benches_read():
    tj = json.load(open(os.path.join(PROF, "hbm_traffic.json")))
    wl = tj["workload"]
    assert (wl["n"], wl["dim"], wl["cells"], wl["nprobe"], wl["m"], wl["k"], wl["batch"]) == (100_000_000, 128, 8192, 32, 16, 100, 16384)
    for key in ("hbm_bytes_per_query", "k_scan_hist_fetch_kib_per_step", "hard_pass_b_fetch_bytes_per_launch", "spread_pass_b_fetch_bytes_per_launch"):
        assert tj[key] > 0, key
    # pass A by kernel family (round 6): K3h reads every query's nearest list about once (0.9 .. 1.5 x the algorithmic 12.2 k codes x
    # 16 bytes per query); K3q reads a list once per block of up to four queries (two queries per list at this batch: 0.4 .. 0.9 x)
    alg = 100_000_000 / 8192 * 16
    pa = tj["pass_a"]
    assert 0.9 < pa["K3h"]["fetch_kib_per_step"] * 2048 / wl["batch"] / alg < 1.5
    assert 0.4 < (pa["K3q"]["fetch_kib_per_step"] * 2048 + pa["K3q"]["write_kib_per_step"] * 1024) / wl["batch"] / alg < 0.9
    assert abs(tj["hbm_bytes_per_query"] * wl["batch"] - (pa["K3q"]["fetch_kib_per_step"] * 2048 + pa["K3q"]["write_kib_per_step"] * 1024)) < wl["batch"]
    assert tj["batch_131072"]["k_scan_q"]["fetch_kib_per_step"] > 0
    b = tj["batch_131072"]
    assert b["workload"]["batch"] == 131072 and b["sweeps_fetch_bytes_per_step"] > 2 * 1_600_000_000  # two sweeps over 1.6 GB of codes
    v = tj["vlad"]
    assert v["descriptors"] > 0 and v["k_vlad_fused_fetch_bytes_per_launch"] >= v["descriptors"] * 64 * 8
    assert v["k_vlad_fused_write_bytes_per_launch"] >= v["images"] * 128 * 64 * 8  # the vector at least once
    y = tj["yfcc"]
    assert y["workload"] == [95213780, 1024, 64, 8192, 4096]
    assert y["k_scan_hist_fetch_bytes_per_launch"] > 0 and y["k_scan_mfma_kc2_fetch_bytes_per_launch"] > 0


def test_profile_files_named_as_sources_exist():
    tj = json.load(open(os.path.join(PROF, "hbm_traffic.json")))
    m = re.search(r"profiles/(r\d+\w*)_\{([^}]*)\}_pmc_kernels\.txt", tj["source"])
    assert m, tj["source"]
    for part in m.group(2).split(","):
        assert os.path.exists(os.path.join(PROF, f"{m.group(1)}_{part}_pmc_kernels.txt")), part
    for f in re.findall(r"profiles/(r\d+\w*_[a-z_]+\.txt)", tj["source"]):
        assert os.path.exists(os.path.join(PROF, f)), f
    readme = open(os.path.join(PROF, "README.md")).read()
    for f in set(re.findall(r"`(r0[45]\w*_[A-Za-z0-9_]+\.(?:txt|json))`", readme)):
        assert os.path.exists(os.path.join(PROF, f)), f
