"""bench.py's N > 1 path end to end with the real kernels: two (and four) ranks (torch.distributed.run) share the box's one GPU, the
collectives run over gloo staged through the host (sharded.HostStagedDist -- RCCL refuses two ranks on one device), and
the answers every rank holds for its half of batch 0 must equal, bit for bit, what the unsharded index of a one-rank run
returns for the same queries: sharded index build (encode by chunk owner, records routed to the list owner), codebook
broadcast, query exchange, pass A / threshold all-reduce / pass B per shard, partial exchange, owner merge, tie replay."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--vectors", "3000000", "--cells", "512", "--w", "8", "--chunk", "500000", "--nbatches", "2", "--steps", "2", "--warmup", "1",
          "--settle", "2", "--no-cpu", "--gt", "0", "--hard-steps", "0", "--spread-steps", "0", "--other-configs", "0", "--exhaustive-steps", "0", "--extras", "0"]


def _result_line(out):
    """the bench contract: the LAST line of the output is the result object"""
    import json

    return json.loads([ln for ln in out.splitlines() if ln.strip()][-1])


def _run(cmd, env):
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("sigma,ranks", [("0.15", 2), ("1.0", 2), ("1.0", 4)])
def test_two_ranks_on_one_gpu_match_the_single_index(tmp_path, sigma, ranks):
    env = dict(os.environ)
    env.pop("MMIDX_LIB", None)
    one = str(tmp_path / "one")
    _run([sys.executable, "bench.py", "--gpus", "1", "--batch", "2048", "--sigma", sigma, "--dump", one] + COMMON, env)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    two = str(tmp_path / "two")
    env2 = dict(env, MMIDX_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                "--master-port", str(port), "bench.py", "--gpus", str(ranks), "--batch", str(2048 // ranks), "--sigma", sigma, "--dump", two] + COMMON, env2)
    assert _result_line(out)["n_gpus"] == ranks
    ref = np.load(one + ".rank0.npz")
    parts = [np.load(f"{two}.rank{r}.npz") for r in range(ranks)]
    for key in ("cnt", "iid", "dist"):
        got = np.concatenate([p[key] for p in parts])
        assert got.shape == ref[key].shape
        assert np.array_equal(got, ref[key]), key
    assert int(ref["cnt"].min()) == 100  # full answers: the comparison is not vacuous


@pytest.mark.gpu
@pytest.mark.parametrize("sigma,ranks", [("1.0", 2), ("0.15", 3)])
def test_native_sharded_bench_path_under_torchrun(tmp_path, sigma, ranks):
    """`bench.py --gpus N` as the driver launches it (torch.distributed.run, N ranks): rank 0 drives the library's own sharded handle
    over all N shards from its one process, the other ranks wait for its verdict (gloo) and leave.  On this one-GPU box the N shards
    are virtual (MMIDX_BENCH_VIRTUAL_SHARDS=1: every shard on device 0, in-process collectives); the answers must equal the plain
    single-GPU index's, bit for bit, and the JSON line must say that the native handle ran."""
    env = dict(os.environ)
    env.pop("MMIDX_LIB", None)
    one = str(tmp_path / "one")
    _run([sys.executable, "bench.py", "--gpus", "1", "--batch", str(ranks * 1024), "--sigma", sigma, "--dump", one] + COMMON, env)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    nat = str(tmp_path / "nat")
    env2 = dict(env, MMIDX_BENCH_VIRTUAL_SHARDS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                "--master-port", str(port), "bench.py", "--gpus", str(ranks), "--batch", "1024", "--sigma", sigma, "--dump", nat] + COMMON, env2)
    res = _result_line(out)
    assert res["n_gpus"] == ranks and "native sharded handle" in res["config"]["multi_gpu_path"] and res["config"]["native_fallback_reason"] is None
    ref = np.load(one + ".rank0.npz")
    got = np.load(nat + ".rank0.npz")
    for key in ("cnt", "iid", "dist"):
        assert got[key].shape == ref[key].shape
        assert np.array_equal(got[key], ref[key]), key
    assert int(ref["cnt"].min()) == 100
