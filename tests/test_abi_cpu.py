"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/mmidx.h declares, and fails loudly (never falls back) without a GPU."""
import ctypes as C
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mi():
    m = importlib.import_module("multimedia-indexing_amd")
    m.build()
    return m


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mmidx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmidx_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported(mi):
    L = mi.lib()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/mmidx.h but not exported"
    # and the python binding covers every one of them
    from importlib import import_module
    nat = import_module("multimedia-indexing_amd._native")
    assert sorted(nat.SIGNATURES) == names
    assert L.mmidx_abi_version() == 8


def test_argument_errors_without_touching_the_gpu(mi):
    L = mi.lib()
    h = C.c_void_p()
    # IVFPQ.java:181-183 / PQ.java:148-150
    st = L.mmidx_create(2, 10, 3, 256, 16, 0, None, None, 0, C.byref(h))
    assert st == 1 and b"subvectors is not valid" in L.mmidx_last_error()
    assert L.mmidx_create(7, 8, 2, 256, 16, 0, None, None, 0, C.byref(h)) == 6
    assert L.mmidx_create(2, 8, 2, 256, 16, 1, None, None, 0, C.byref(h)) == 6  # rotation w/o matrix
    assert L.mmidx_create(2, 8, 2, 256, 16, 0, None, None, 0, None) == 6
    assert L.mmidx_set_w(None, 3) == 6
    assert L.mmidx_destroy(None) == 0


@pytest.mark.skipif(importlib.import_module("multimedia-indexing_amd").lib().mmidx_device_count() > 0,
                    reason="a GPU is present")
def test_no_cpu_fallback(mi):
    """Without a HIP device the product path must fail, not silently compute on the host."""
    with pytest.raises(mi.MmidxError) as ei:
        mi.IVFPQ(8, 100, False, "", 2, 4, 0, 4, 512)
    assert ei.value.status == 8
    with pytest.raises(mi.MmidxError):
        mi.PQ(8, 100, False, "", 2, 4, 0, 512)


def test_product_does_not_reference_the_oracle():
    """The shipped path must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "multimedia-indexing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "mmidx_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
