"""The JNI shim and the Java declarations cannot be built here (no JDK).  What CAN be checked on any box:
  * mmidx_jni.c type-checks (gcc -fsyntax-only) against include/mmidx.h and a stand-in for jni.h that declares the JNI
    function-table entries it uses with the specification's signatures (tests/jni_stub/jni.h: never linked, never run);
  * every `native` method of MmidxNative.java has a JNI function of the matching mangled name in the shim and vice versa,
    with the same number of parameters;
  * every Gpu* class only calls MmidxNative methods that exist, with the declared number of arguments."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JNI = os.path.join(ROOT, "multimedia-indexing_amd", "jni")
JAVA = os.path.join(JNI, "java", "gr", "iti", "mklab", "visual")


def _natives():
    src = open(os.path.join(JAVA, "datastructures", "MmidxNative.java")).read()
    out = {}
    for m in re.finditer(r"public static native \w+(?:\[\])? (\w+)\(([^)]*)\)", src):
        args = [a for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = len(args)
    return out


def _shim_functions():
    src = open(os.path.join(JNI, "mmidx_jni.c")).read()
    out = {}
    for m in re.finditer(r"JNICALL JFN\((\w+)\)\(([^)]*)\)", src):
        args = [a for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = len(args) - 2  # JNIEnv *, jclass
    return out


def test_shim_type_checks():
    r = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "jni_stub"),
                        "-I", os.path.join(ROOT, "include"), os.path.join(JNI, "mmidx_jni.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_native_declarations_match_the_shim():
    nat, shim = _natives(), _shim_functions()
    assert len(nat) >= 25
    assert set(nat) == set(shim), (sorted(set(nat) ^ set(shim)))
    for name, n in nat.items():
        assert shim[name] == n, (name, n, shim[name])


def test_java_classes_call_existing_natives():
    nat = _natives()
    for d, _, files in os.walk(JAVA):
        for f in files:
            if not f.endswith(".java") or f == "MmidxNative.java":
                continue
            src = open(os.path.join(d, f)).read()
            for m in re.finditer(r"MmidxNative\.(\w+)\(", src):
                name = m.group(1)
                if name.isupper() or name.startswith("KIND"):
                    continue
                assert name in nat, (f, name)
                # argument count: scan to the matching parenthesis
                i, depth, args, cur = m.end(), 1, 0, ""
                while depth:
                    ch = src[i]
                    if ch in "([{":
                        depth += 1
                    elif ch in ")]}":
                        depth -= 1
                    if depth == 1 and ch == ",":
                        args += 1
                    elif depth >= 1:
                        cur += ch
                    i += 1
                n = args + (1 if cur.strip() else 0)
                assert n == nat[name], (f, name, n, nat[name])


def test_short_code_paths_exist():
    """numProductCentroids > 256 must travel as short[] end to end"""
    ivf = open(os.path.join(JAVA, "datastructures", "GpuIVFPQ.java")).read()
    pq = open(os.path.join(JAVA, "datastructures", "GpuPQ.java")).read()
    for src in (ivf, pq):
        assert "addVectorShort" in src and "addCodesShort" in src and "writeShort" in src and "readShort" in src
    for name in ("computeDistanceIVFADC", "getPQCodeByte", "getPQCodeShort", "getInvertedListId", "outputItemsPerList", "setW",
                 "loadCoarseQuantizer", "loadProductQuantizer", "indexPQCode"):
        assert re.search(r"public (?:synchronized )?[\w\[\]]+ " + name + r"\(", ivf), name


def test_java_crosscheck_tooling_round_trip(tmp_path):
    """tools/java_crosscheck: the exporter writes what CrossCheck.java reads, and the comparer accepts a dump that equals
    the expected answers (stand-in for the Java run, which needs a JDK + the reference jars) and rejects a corrupted one"""
    import shutil
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools", "java_crosscheck"))
    sys.path.insert(0, ROOT)
    import compare
    import export_fixtures
    from oracle import oracle as o

    fx, out = str(tmp_path / "fx"), str(tmp_path / "out")
    export_fixtures.main(fx)
    os.makedirs(out)
    with open(os.path.join(out, "permutation.csv"), "w") as f:
        for dim in (3, 8, 128):
            f.write(",".join([str(dim)] + [str(int(x)) for x in o.random_permutation(1, dim)]) + "\n")
    for name in os.listdir(fx):
        if not os.path.exists(os.path.join(fx, name, "expected.answers.csv")):  # (the RandomRotation case: tests/test_crosscheck_tools_cpu.py plays CrossCheck's part for it)
            shutil.rmtree(os.path.join(fx, name))
            continue
        shutil.copy(os.path.join(fx, name, "expected.answers.csv"), os.path.join(out, name + ".answers.csv"))
        meta = open(os.path.join(fx, name, "meta.csv")).read().strip().split(",")
        assert len(meta) == 9
        # doubles survive the hex transport
        first = open(os.path.join(fx, name, "base.csv")).readline().strip().split(",")
        assert len(first) == int(meta[0]) and all(len(x) == 16 for x in first)
    assert compare.main(fx, out) == 0
    p = os.path.join(out, "ivfpq_small.answers.csv")
    lines = open(p).read().split("\n")
    lines[3] = lines[3].replace(":", ":1", 1)
    open(p, "w").write("\n".join(lines))
    assert compare.main(fx, out) == 1
    src = open(os.path.join(ROOT, "tools", "java_crosscheck", "CrossCheck.java")).read()
    for call in ("loadCoarseQuantizer", "loadProductQuantizer", "indexVector", "computeNearestNeighbors", "setW"):
        assert call in src
