import os
import sys


def _usable_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup quota (a GPU box shows 256 logical CPUs and grants
    16: a BLAS that starts 256 threads under that quota is throttled to a crawl -- the same suite took 5 or 14 minutes)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        qv, pv = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if qv != "max":
            n = max(1, min(n, int(float(qv) / float(pv) + 0.999)))
    except (OSError, ValueError):
        pass
    return n


for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):  # (before numpy is imported; inherited by subprocesses)
    os.environ.setdefault(_v, str(_usable_cpus()))

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure). Built on demand with gcc."""
    from oracle import oracle as o

    o.build()
    o.lib()
    return o


@pytest.fixture(scope="session", autouse=True)
def _blas_threads():
    """numpy may have been imported before the environment above was set: cap its thread pools at run time as well"""
    try:
        from threadpoolctl import threadpool_limits

        with threadpool_limits(limits=_usable_cpus()):
            yield
    except ImportError:
        yield
