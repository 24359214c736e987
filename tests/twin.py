"""TEST-ONLY: the hot path driven through the raw C ABI (include/mdd.h) with ctypes, on whichever of the two libraries
is handed in -- libmdd_hip.so (device pointers) or oracle/libmdd_cpu.so (host pointers).  The call sequence is the same;
only the buffers differ (`alloc` / `to_host`)."""
import ctypes as C
import os

import numpy as np

from map_neo_amd.lib import SIGNATURES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_SO = os.path.join(ROOT, "oracle", "libmdd_cpu.so")
TWIN_SYMBOLS = ("mdd_version", "mdd_last_error", "mdd_ctx_create", "mdd_ctx_destroy", "mdd_ctx_sync", "mdd_make_perms",
                "mdd_signatures", "mdd_band_keys", "mdd_bucket_edges", "mdd_components_init", "mdd_components_union",
                "mdd_components_finish", "mdd_signature_matches", "mdd_removed_mask")


def load_cpu():
    L = C.CDLL(CPU_SO)
    for name in TWIN_SYMBOLS:
        fn = getattr(L, name)
        fn.restype, fn.argtypes = SIGNATURES[name]
    return L


class HostBuffers:
    """numpy arrays; pointers are host pointers"""

    def put(self, arr):
        return np.ascontiguousarray(arr).copy()

    def zeros(self, n, dtype):
        return np.zeros(max(1, n), dtype=dtype)

    def ptr(self, a):
        return a.ctypes.data

    def get(self, a):
        return a


class DeviceBuffers:
    """torch tensors on the engine's device; pointers are device pointers"""

    def __init__(self, device):
        import torch
        self.torch, self.device = torch, device
        self.view = {np.dtype(np.uint8): torch.uint8, np.dtype(np.uint32): torch.int32, np.dtype(np.uint64): torch.int64}

    def put(self, arr):
        arr = np.ascontiguousarray(arr)
        signed = {np.dtype(np.uint32): np.int32, np.dtype(np.uint64): np.int64}.get(arr.dtype, arr.dtype)
        return self.torch.from_numpy(arr.view(signed).copy()).to(self.device)

    def zeros(self, n, dtype):
        return self.torch.zeros(max(1, n), dtype=self.view[np.dtype(dtype)], device=self.device)

    def ptr(self, t):
        return t.data_ptr()

    def get(self, t):
        a = t.cpu().numpy()
        return a.view({np.dtype(np.int32): np.uint32, np.dtype(np.int64): np.uint64}.get(a.dtype, a.dtype))


def run_path(L, ctx, B, text, offsets, K=128, width=13, unit=0, bands=9, rows=13, seed=1, key_bits=64, sync=None):
    """signatures -> band records -> bucket edges -> components -> removed mask through the C ABI.  Returns host arrays."""

    def ok(rc, what):
        assert rc == 0, (what, rc, L.mdd_last_error().decode())

    n = len(offsets) - 1
    tb = int(offsets[-1])
    a, b = np.zeros(K, np.uint64), np.zeros(K, np.uint64)
    ok(L.mdd_make_perms(seed, K, a.ctypes.data, b.ctypes.data), "mdd_make_perms")
    pad = np.zeros((tb + 3) // 4 * 4 + 128, np.uint8)
    pad[:tb] = text[:tb]
    t_d, o_d, a_d, b_d = B.put(pad), B.put(np.asarray(offsets, np.uint64)), B.put(a), B.put(b)
    sig = B.zeros(n * K, np.uint32)
    ok(L.mdd_signatures(ctx, B.ptr(t_d), tb, B.ptr(o_d), n, unit, width, K, B.ptr(a_d), B.ptr(b_d), B.ptr(sig)), "mdd_signatures")
    n_rec = n * bands
    key, meta = B.zeros(n_rec, np.uint64), B.zeros(n_rec, np.uint64)
    ok(L.mdd_band_keys(ctx, B.ptr(sig), n, K, bands, rows, 0, key_bits, B.ptr(key), B.ptr(meta)), "mdd_band_keys")
    tab_size = 2
    while tab_size < 2 * max(1, n_rec):
        tab_size *= 2
    table = B.zeros(tab_size, np.uint64)
    eu, ev = B.zeros(n_rec, np.uint32), B.zeros(n_rec, np.uint32)
    cnt, coll, rb = B.zeros(1, np.uint64), B.zeros(1, np.uint64), B.zeros(n_rec, np.uint64)
    ok(L.mdd_bucket_edges(ctx, B.ptr(key), B.ptr(meta), n_rec, B.ptr(sig), bands, K, rows, rows, B.ptr(table), tab_size,
                          B.ptr(eu), B.ptr(ev), n_rec, B.ptr(cnt), B.ptr(coll), B.ptr(rb)), "mdd_bucket_edges")
    ok(L.mdd_ctx_sync(ctx), "mdd_ctx_sync")
    n_edges = int(B.get(cnt)[0])
    labels = B.zeros(n, np.uint32)
    ok(L.mdd_components_init(ctx, B.ptr(labels), n), "mdd_components_init")
    ok(L.mdd_components_union(ctx, B.ptr(labels), n, B.ptr(eu), B.ptr(ev), n_edges), "mdd_components_union")
    ok(L.mdd_components_finish(ctx, B.ptr(labels), n), "mdd_components_finish")
    matches = B.zeros(n_edges, np.uint32)
    ok(L.mdd_signature_matches(ctx, B.ptr(sig), n, K, B.ptr(eu), B.ptr(ev), n_edges, B.ptr(matches)), "mdd_signature_matches")
    removed, n_rm = B.zeros(n, np.uint8), B.zeros(1, np.uint64)
    ok(L.mdd_removed_mask(ctx, B.ptr(labels), 0, n, B.ptr(removed), B.ptr(n_rm)), "mdd_removed_mask")
    ok(L.mdd_ctx_sync(ctx), "mdd_ctx_sync")
    edges = sorted(zip(B.get(eu)[:n_edges].tolist(), B.get(ev)[:n_edges].tolist(), B.get(matches)[:n_edges].tolist()))
    return {"sig": B.get(sig)[: n * K].reshape(n, K).copy(), "key": B.get(key)[:n_rec].copy(), "meta": B.get(meta)[:n_rec].copy(),
            "labels": B.get(labels)[:n].copy(), "removed": B.get(removed)[:n].copy(), "n_removed": int(B.get(n_rm)[0]),
            "rec_bucket": B.get(rb)[:n_rec].copy(), "edges": edges, "n_collisions": int(B.get(coll)[0])}
