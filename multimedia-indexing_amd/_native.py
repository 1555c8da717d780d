"""ctypes binding of libmmidx_hip.so (the C ABI of include/mmidx.h).

The product path has no CPU fallback: if the HIP library is missing or no MI355X is visible,
every operation raises.  torch is NOT needed here; it is only used by callers that want their
inputs resident in HBM (bench, multi-GPU).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# MMIDX_LIB selects another build of the same library (kernel A/B experiments on the GPU box)
SO_PATH = os.environ.get("MMIDX_LIB") or os.path.join(CSRC, "libmmidx_hip.so")

OK = 0
STATUS_NAMES = {
    1: "INVALID_SUBVECTORS", 2: "WRONG_DIM", 3: "NOT_IN_MEMORY", 4: "BYTE_OVERFLOW", 5: "CAPACITY",
    6: "INVALID_ARG", 7: "NOT_READY", 8: "NO_DEVICE", 9: "HIP", 10: "UNSUPPORTED",
}
ERR_INVALID_SUBVECTORS, ERR_WRONG_DIM, ERR_NOT_IN_MEMORY, ERR_BYTE_OVERFLOW = 1, 2, 3, 4
ERR_CAPACITY, ERR_INVALID_ARG, ERR_NOT_READY, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED = 5, 6, 7, 8, 9, 10

KIND_PQ, KIND_IVFPQ = 1, 2
TR_NONE, TR_ROTATION, TR_PERMUTATION = 0, 1, 2


class MmidxError(Exception):
    """Mirrors the reference's `throw new Exception(msg)`; .status carries the C status code."""

    def __init__(self, status, msg):
        super().__init__(msg)
        self.status = status


class Stats(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("coarse_ms", C.c_double), ("scan_ms", C.c_double),
                ("merge_ms", C.c_double), ("scan_codes", C.c_int64), ("scan_launches", C.c_int32),
                ("tie_fallbacks", C.c_int32), ("passa_ms", C.c_double), ("passa_codes", C.c_int64),
                ("passa_launches", C.c_int32), ("passb_items_last", C.c_int32), ("verified_codes", C.c_int64),
                ("mfma_survivors", C.c_int64), ("mfma_redo_queries", C.c_int64),
                ("mfma_scan_ms", C.c_double), ("mfma_verify_ms", C.c_double), ("mfma_launches", C.c_int32), ("reserved0", C.c_int32),
                ("passa_mfma_launches", C.c_int64), ("passa_mfma_sweep1_ms", C.c_double), ("passa_mfma_select_ms", C.c_double),
                ("passa_mfma_sweep2_ms", C.c_double), ("passa_mfma_verify_ms", C.c_double)]


def build(force=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in ("mmidx_api.hip", "mmidx_learn.hip", "mmidx_probe.hip", "mmidx_kernels.h", "mmidx_scan_grp.h", "mmidx_frontend.h", "mmidx_sharded.h")]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "mmidx.h"))
    stale = not os.path.exists(SO_PATH) or any(os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s"] + (["-B"] if force else []))
    return SO_PATH


_lib = None

_vp, _i32p, _dp = C.c_void_p, C.c_void_p, C.c_void_p
SIGNATURES = {
    "mmidx_last_error": (C.c_char_p, []),
    "mmidx_abi_version": (C.c_int, []),
    "mmidx_device_count": (C.c_int, []),
    "mmidx_create": (C.c_int, [C.c_int] * 6 + [_vp, _vp, C.c_int, C.POINTER(C.c_void_p)]),
    "mmidx_destroy": (C.c_int, [_vp]),
    "mmidx_set_coarse": (C.c_int, [_vp, _dp]),
    "mmidx_set_pq": (C.c_int, [_vp, _dp]),
    "mmidx_set_w": (C.c_int, [_vp, C.c_int]),
    "mmidx_get_w": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "mmidx_size": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "mmidx_list_sizes": (C.c_int, [_vp, _i32p]),
    "mmidx_encode": (C.c_int, [_vp, C.c_int64, _dp, _i32p, _vp]),
    "mmidx_add_vectors": (C.c_int, [_vp, C.c_int64, _dp, _i32p, _i32p, _vp]),
    "mmidx_add_codes": (C.c_int, [_vp, C.c_int64, _i32p, _i32p, _vp]),
    "mmidx_add_vectors_device": (C.c_int, [_vp, C.c_int64, _dp, _i32p, C.c_int32, _vp]),
    "mmidx_add_codes_device": (C.c_int, [_vp, C.c_int64, _i32p, _i32p, _vp, _vp]),
    "mmidx_encode_device": (C.c_int, [_vp, C.c_int64, _dp, _i32p, _vp, _vp]),
    "mmidx_assign_device": (C.c_int, [_vp, C.c_int64, _dp, _i32p, _vp]),
    "mmidx_sync_index": (C.c_int, [_vp]),
    "mmidx_export": (C.c_int, [_vp, _vp, _i32p, _vp]),
    "mmidx_save": (C.c_int, [_vp, C.c_char_p]),
    "mmidx_load": (C.c_int, [_vp, C.c_char_p]),
    "mmidx_get_dims": (C.c_int, [_vp] + [C.POINTER(C.c_int)] * 5),
    "mmidx_get_codes": (C.c_int, [_vp, C.c_int64, _i32p, _i32p, _vp]),
    "mmidx_distance": (C.c_int, [_vp, C.c_int64, _dp, _i32p, _dp]),
    "mmidx_search": (C.c_int, [_vp, C.c_int, C.c_int64, _dp, _i32p, _dp, _i32p]),
    "mmidx_search_sdc": (C.c_int, [_vp, C.c_int, C.c_int64, _i32p, _i32p, _dp, _i32p]),
    "mmidx_search_device": (C.c_int, [_vp, C.c_int, C.c_int64, _dp, _i32p, _dp, _i32p, _vp]),
    "mmidx_coarse_device": (C.c_int, [_vp, C.c_int64, _dp, _i32p, _dp, _vp]),
    "mmidx_search_partial_device": (C.c_int, [_vp, C.c_int, C.c_int64, _dp, _i32p, _dp, _vp, _i32p, _vp]),
    "mmidx_shard_pass_a_device": (C.c_int, [_vp, C.c_int, C.c_int64, _dp, _i32p, _dp, _vp]),
    "mmidx_shard_pass_b_device": (C.c_int, [_vp, C.c_int, C.c_int64, _dp, _i32p, _dp, _dp, _dp, _vp, _i32p, _vp]),
    "mmidx_compact_partials_device": (C.c_int, [C.c_int, C.c_int, C.c_int64, _dp, _vp, _i32p, _vp, _dp, _vp, _vp]),
    "mmidx_merge_partials_device": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int, _dp, _vp, _i32p, _vp, _i32p, _dp, _i32p, _i32p, _vp]),
    "mmidx_shard_tie_phase_device": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int64, _dp, _i32p, _i32p, _dp, _i32p, _i32p, _i32p, _vp]),
    "mmidx_pca_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.POINTER(C.c_void_p)]),
    "mmidx_pca_destroy": (C.c_int, [_vp]),
    "mmidx_pca_project": (C.c_int, [_vp, C.c_int64, _dp, _dp]),
    "mmidx_pca_project_device": (C.c_int, [_vp, C.c_int64, _dp, _dp, _vp]),
    "mmidx_vlad_create": (C.c_int, [C.c_int, _i32p, C.c_int, _dp, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mmidx_vlad_destroy": (C.c_int, [_vp]),
    "mmidx_vlad_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "mmidx_vlad_vector_length": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "mmidx_vlad_aggregate": (C.c_int, [_vp, C.c_int64, _vp, _dp, _dp]),
    "mmidx_vlad_aggregate_device": (C.c_int, [_vp, C.c_int64, _vp, _dp, C.c_int, _dp, _vp]),
    "mmidx_linear_create": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]),
    "mmidx_linear_destroy": (C.c_int, [_vp]),
    "mmidx_linear_add": (C.c_int, [_vp, C.c_int64, _dp]),
    "mmidx_linear_size": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "mmidx_linear_get_vector": (C.c_int, [_vp, C.c_int64, _dp]),
    "mmidx_linear_search": (C.c_int, [_vp, C.c_int, C.c_int64, _dp, _i32p, _dp, _i32p]),
    "mmidx_vectorize": (C.c_int, [_vp, _vp, C.c_int64, _vp, _dp, _dp]),
    "mmidx_vectorize_device": (C.c_int, [_vp, _vp, C.c_int64, _vp, _dp, C.c_int, _dp, _vp]),
    "mmidx_kmeans_device": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, _dp, _dp, _dp, _i32p, _dp,
                                      _i32p, _i32p, _vp]),
    "mmidx_kmeans": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, _dp, _dp, _dp, _i32p, _dp, _i32p,
                               _i32p]),
    "mmidx_create_sharded": (C.c_int, [C.c_int] * 6 + [_vp, _vp, C.c_int, _vp, C.POINTER(C.c_void_p)]),
    "mmidx_shard_count": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "mmidx_shard_info": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "mmidx_search_sliced_device": (C.c_int, [_vp, C.c_int, C.c_int64, _vp, _vp, _vp, _vp]),
    "mmidx_add_vectors_sliced_device": (C.c_int, [_vp, _vp, _vp, C.c_int32]),
    "mmidx_pca_get_dims": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mmidx_vlad_descriptor_length": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "mmidx_linear_get_dim": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "mmidx_probe_lds_gather": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp]),
    "mmidx_probe_f64_mfma": (C.c_int, [C.c_int, _dp]),
    "mmidx_probe_split_gather": (C.c_int, [C.c_int, C.c_int, _dp]),
    "mmidx_set_profiling": (C.c_int, [_vp, C.c_int]),
    "mmidx_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "mmidx_get_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "mmidx_get_dispatch": (C.c_int, [_vp, C.c_char_p, C.c_int]),
}


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (same soname as /opt/rocm's).  One
    process can only drive the GPU through one HIP runtime, so when torch is installed -- the
    bench and the multi-GPU host use it for device buffers and RCCL -- bind to ITS copy, whatever
    the import order.  torch itself is not imported here.  Without torch (e.g. under the JNI shim)
    the library resolves to /opt/rocm/lib through its RUNPATH."""
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def preload_rccl():
    """A sharded handle (mmidx_create_sharded) loads RCCL with dlopen("librccl.so.1") on first use.  PyTorch-ROCm wheels bundle
    their own librccl.so next to their HIP runtime; when torch is installed bind to THAT pair (a library with the same SONAME that
    is already mapped is what dlopen returns), as _preload_hip_runtime does for libamdhip64.  Without torch (the JNI shim) the
    library resolves to /opt/rocm/lib through libmmidx_hip.so's RUNPATH."""
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    """Load libmmidx_hip.so; raises loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    _preload_hip_runtime()
    L = C.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(L, name)  # AttributeError if the symbol is not exported
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def check(status):
    if status != OK:
        msg = lib().mmidx_last_error().decode("utf-8", "replace")
        raise MmidxError(status, msg)
