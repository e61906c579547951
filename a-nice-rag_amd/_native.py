"""ctypes binding of libanrag.so (include/anrag.h) -- the only way the package computes.

Loading order matters on this image: PyTorch bundles its own HIP runtime
(`torch/lib/libamdhip64.so`, SONAME libamdhip64.so.7, the same SONAME as
/opt/rocm's).  `import torch` first makes libanrag.so bind to the runtime torch
already mapped, so device pointers and streams are interchangeable between
torch (device memory, streams, torch.distributed/RCCL) and the library.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ANRAG_LIB: another build of the same library (the `make dbg` diagnostic build); there is still no other backend
LIB_PATH = os.environ.get("ANRAG_LIB") or os.path.join(_HERE, "libanrag.so")

OK = 0
FUSED_K_MAX = 64
KERNEL_DENSE_SCAN, KERNEL_DENSE_BATCHED, KERNEL_BM25, KERNEL_SELECT, KERNEL_WRRF = range(5)


class AnragError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libanrag error {code}: {msg}")
        self.code = code


class Candidate(C.Structure):
    _fields_ = [("score", C.c_double), ("doc", C.c_int64)]


CANDIDATE_DTYPE = np.dtype([("score", "<f8"), ("doc", "<i8")])

LEG_DENSE, LEG_BM25 = 0, 1


class RankLeg(C.Structure):
    """`anrag_rank_leg` (include/anrag.h): one ranked list per query of an `anrag_rank_batch` call."""
    _fields_ = [("idx", C.c_void_p), ("kind", C.c_int32), ("queries", C.c_void_p), ("term_ids", C.c_void_p),
                ("term_offsets", C.c_void_p), ("allow_source", C.c_void_p), ("n_sources", C.c_int32),
                ("doc_of_row", C.c_void_p), ("weight", C.c_double)]


_lib: Optional[C.CDLL] = None

_p = C.c_void_p
_i32, _i64, _f64 = C.c_int32, C.c_int64, C.c_double

# name -> argtypes; every function returns int except anrag_last_error
_SIGNATURES = {
    "anrag_abi_version": [],
    "anrag_device_count": [C.POINTER(C.c_int)],
    "anrag_index_create": [C.c_int, C.POINTER(_p)],
    "anrag_index_destroy": [_p],
    "anrag_index_set_streams": [_p, _p, _p, _p],
    "anrag_index_sync": [_p],
    "anrag_index_wait_stream": [_p, _p],
    "anrag_index_signal_stream": [_p, _p],
    "anrag_dense_load": [_p, _p, _i64, _i32, _p, _p, _i64],
    "anrag_dense_search": [_p, _p, _i32, _i32, _p, _i32, _p, _p, _p],
    "anrag_dense_search_f64": [_p, _p, _i32, _p, _i32, _p, _p, _p],
    "anrag_dense_search_device": [_p, _p, _i32, _i32, _p, _p],
    "anrag_dense_search_batch_device": [_p, _p, _i32, _i32, _p, _p, _p],
    "anrag_set_batched_precision": [_p, _i32],
    "anrag_dense_scores": [_p, _p, _p],
    "anrag_bm25_load": [_p, _p, _i64, _p, _p, _p, _p, _i64, _f64, _f64, _f64, _p, _p, _i64],
    "anrag_bm25_search": [_p, _p, _i32, _i32, _p, _i32, _p, _p, _p],
    "anrag_bm25_search_device": [_p, _p, _i32, _i32, _p, _p],
    "anrag_bm25_search_group_device": [_p, _p, _p, _i32, _i32, _p, _p],
    "anrag_bm25_scores": [_p, _p, _i32, _p],
    "anrag_wrrf": [_p, _p, _p, _p, _i32, _f64, _i32, _p, _p, _p],
    "anrag_hybrid_search": [_p, _p, _p, _i32, _i32, _f64, _f64, _f64, _i32, _p, _i32, _p, _i32, _p, _p, _p],
    "anrag_hybrid_search_batch": [_p, _p, _p, _p, _i32, _i32, _f64, _f64, _f64, _i32, _p, _i32, _p, _i32, _p, _p, _p],
    "anrag_merge_candidates_device": [_p, _p, _i32, _i32, _i64, _p],
    "anrag_hybrid_candidates_device": [_p, _p, _p, _i32, _i32, _p, _p, _p],
    "anrag_hybrid_candidates_group_device": [_p, _p, _p, _p, _i32, _i32, _p, _p, _p],
    "anrag_merge_fuse_device": [_p, _p, _i32, _i32, _i64, _f64, _f64, _f64, _i32, _i32, _p, _p],
    "anrag_wrrf_device": [_p, _p, _i32, _p, _i32, _f64, _f64, _f64, _i32, _p, _p],
    "anrag_hybrid_search_device": [_p, _p, _p, _i32, _i32, _f64, _f64, _f64, _i32, _p, _p, _p, _p],
    "anrag_device_alloc": [_p, _i64, C.POINTER(_p)],
    "anrag_device_free": [_p, _p],
    "anrag_copy_to_device": [_p, _p, _p, _i64],
    "anrag_copy_to_host": [_p, _p, _p, _i64],
    "anrag_profile_enable": [_p, C.c_uint32],
    "anrag_profile_set_sampling": [_p, _i32],
    "anrag_profile_reset": [_p],
    "anrag_profile_read": [_p, C.c_int, C.POINTER(_f64), C.POINTER(_i64)],
    "anrag_profile_read_units": [_p, C.c_int, C.POINTER(_i64)],
    "anrag_index_info": [_p, C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)],
    "anrag_debug_alloc_calls": [C.POINTER(_i64)],
    "anrag_rank_batch": [_p, _i32, _i32, _i32, _f64, _i32, _i64, _p, _p, _p, _p, _p],
    "anrag_rank_caps": [C.POINTER(_i32), C.POINTER(_i32)],
}
EXPORTS = tuple(_SIGNATURES) + ("anrag_last_error",)


def load_library() -> C.CDLL:
    """dlopen libanrag.so once.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AnragError(-100, f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); this package has no CPU fallback")
    import torch  # noqa: F401  -- maps torch's libamdhip64.so.7 first (see module docstring)

    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, args in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if os.environ.get("ANRAG_LIB"):  # another build of the library (an older one, for A/B measurements)
                continue
            raise
        fn.argtypes = args
        fn.restype = C.c_int
    lib.anrag_last_error.argtypes = []
    lib.anrag_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def last_error() -> str:
    return load_library().anrag_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != OK:
        raise AnragError(rc, last_error())


def device_count() -> int:
    n = C.c_int(0)
    rc = load_library().anrag_device_count(C.byref(n))
    return n.value if rc == OK else 0


def ptr(a) -> Optional[int]:
    """Address of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data
