"""ctypes binding of include/mdd.h (libmdd_hip.so).  Fails loudly when the library is
missing: there is no CPU path in this package."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libmdd_hip.so")

MDD_UNIT = {"char": 0, "word": 1}

# name -> (restype, argtypes); the single source the loader and the symbol test share
_vp, _u64, _int, _u32 = C.c_void_p, C.c_uint64, C.c_int, C.c_uint32
SIGNATURES = {
    "mdd_version": (_int, []),
    "mdd_last_error": (C.c_char_p, []),
    "mdd_ctx_create": (_int, [_int, _vp, C.POINTER(_vp)]),
    "mdd_ctx_destroy": (_int, [_vp]),
    "mdd_ctx_sync": (_int, [_vp]),
    "mdd_ctx_set_segment_bytes": (_int, [_vp, _u32]),
    "mdd_ctx_set_filter_from": (_int, [_vp, _u32]),
    "mdd_make_perms": (_int, [_u32, _int, _vp, _vp]),
    "mdd_signatures": (_int, [_vp, _vp, _u64, _vp, _u64, _int, _int, _int, _vp, _vp, _vp]),
    "mdd_signatures_noalloc": (_int, [_vp, _vp, _u64, _vp, _u64, _int, _int, _int, _vp, _vp, _vp]),
    "mdd_signatures_from_host": (_int, [_vp, _vp, _vp, _u64, _vp, _vp, _u64, _int, _int, _int, _vp, _vp, _vp, _u64]),
    "mdd_stream_wait_ticks": (_int, [_vp, _vp]),
    "mdd_shingle_hashes": (_int, [_vp, _vp, _u64, _u64, _u64, _int, _int, _vp, _u64, _vp]),
    "mdd_band_keys": (_int, [_vp, _vp, _u64, _int, _int, _int, _u64, _int, _vp, _vp]),
    "mdd_packed_keys": (_int, [_vp, _vp, _vp, _u64, _int, _int, _vp]),
    "mdd_route_count": (_int, [_vp, _vp, _u64, _int, _vp]),
    "mdd_route_pack": (_int, [_vp, _vp, _int, _int, _int, _vp, _vp, _u64, _int, _vp, _vp, _vp]),
    "mdd_comm_unique_id": (_int, [_vp]),
    "mdd_comm_init": (_int, [_vp, _vp, _int, _int]),
    "mdd_comm_destroy": (_int, [_vp]),
    "mdd_exchange_counts": (_int, [_vp, _vp, _vp, _vp]),
    "mdd_exchange_bands": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _vp, _vp]),
    "mdd_allgather_counts": (_int, [_vp, _vp, _vp]),
    "mdd_allgather_edges": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "mdd_bucket_edges": (_int, [_vp, _vp, _vp, _u64, _vp, _u64, _u64, _u64, _int, _vp, _u64,
                                _vp, _vp, _u64, _vp, _vp, _vp]),
    "mdd_components_init": (_int, [_vp, _vp, _u64]),
    "mdd_components_union": (_int, [_vp, _vp, _u64, _vp, _vp, _u64]),
    "mdd_components_finish": (_int, [_vp, _vp, _u64]),
    "mdd_forest_edges": (_int, [_vp, _vp, _u64, _vp, _vp, _u64, _vp]),
    "mdd_signature_matches": (_int, [_vp, _vp, _u64, _int, _vp, _vp, _u64, _vp]),
    "mdd_removed_mask": (_int, [_vp, _vp, _u64, _u64, _vp, _vp]),
    "mdd_para_count": (_int, [_vp, _vp, _u64, _vp, _u64, _vp]),
    "mdd_para_units": (_int, [_vp, _vp, _u64, _vp, _u64, _vp, _u64, _vp, _vp]),
    "mdd_para_compact_count": (_int, [_vp, _vp, _u64, _vp, _u64, _vp, _u64, _vp, _vp, _vp, _vp]),
    "mdd_para_compact_write": (_int, [_vp, _vp, _u64, _vp, _u64, _vp, _u64, _vp, _vp, _vp, _vp, _u64, _vp]),
    "mdd_para_doc_sums": (_int, [_vp, _vp, _u64, _u64, _vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mdd_substr_mark": (_int, [_vp, _vp, _u64, _vp, _u64, _int, _int, _int, _vp, _u64, _vp]),
    "mdd_substr_mark_sorted": (_int, [_vp, _vp, _u64, _vp, _u64, _int, _int, _int, _vp, _vp, _u64, _vp, _vp, _u64, _vp, _vp]),
    "mdd_substr_candidates": (_int, [_vp, _vp, _u64, _vp, _u64, _int, _vp, _vp, _u64, _vp, _vp, _u64, _vp, _vp, _vp]),
    "mdd_substr_check": (_int, [_vp, _vp]),
    "mdd_substr_count": (_int, [_vp, _vp, _u64, _vp, _u64, _int, _vp, _vp, _vp]),
    "mdd_substr_write": (_int, [_vp, _vp, _u64, _vp, _u64, _int, _vp, _vp, _vp, _vp, _u64]),
    "mdd_normalize_en_count": (_int, [_vp, _vp, _u64, _vp, _u64, _vp, _vp]),
    "mdd_normalize_en_write": (_int, [_vp, _vp, _u64, _vp, _u64, _vp, _vp, _vp, _u64]),
    "mdd_gather_docs": (_int, [_vp, _vp, _u64, _vp, _u64, _vp, _u64, _vp, _vp, _u64]),
    "mdd_exact_group": (_int, [_vp, _vp, _u64, _vp, _vp, _u64, _u64, _int, _vp, _vp, _u64, _vp, _vp]),
    "mdd_exact_group_cps": (_int, [_vp, _vp, _u64, _vp, _vp, _u64, _u64, _int, _vp, _vp, _u64, _vp, _vp, _vp]),
    "mdd_jsonl_count_lines": (_int, [_vp, _vp, _u64, _vp, _vp]),
    "mdd_jsonl_line_begins": (_int, [_vp, _vp, _u64, _vp, _vp, _u64]),
    "mdd_jsonl_scan": (_int, [_vp, _vp, _u64, _vp, _u64, C.c_char_p, C.c_char_p, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mdd_jsonl_decode": (_int, [_vp, _vp, _u64, _u64, _vp, _vp, _vp, _vp, _u64]),
    "mdd_valu_calibrate": (_int, [_vp, _int, C.POINTER(C.c_double)]),
}


class MddError(RuntimeError):
    pass


_lib = None


def load():
    """Load libmdd_hip.so and bind every symbol of include/mdd.h."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise MddError(
                f"{SO_PATH} not found: build it with `make -C map_neo_amd/csrc` "
                "(or __graft_entry__.build()).  map_neo_amd has no CPU fallback.")
        # torch ships its own libamdhip64; load it FIRST so that libmdd_hip.so binds to the same
        # HIP runtime instance torch uses (two runtimes in one process do not share devices/streams)
        import torch  # noqa: F401
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)      # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mdd_last_error().decode("utf-8", "replace")
        raise MddError(f"{what or 'libmdd_hip'} failed (code {rc}): {msg}")
