"""Engine: thin host wrapper over the C ABI (include/mdd.h).  torch is used only to own
device memory and to supply the stream; every computation on the path is a HIP kernel in
libmdd_hip.so.  u32 data lives in torch.int32 tensors and u64 data in torch.int64 tensors
(same bits); `as_u32` / `as_u64` give the numpy views on the host.

Reference functions mirrored (paths relative to /root/reference, see include/mdd.h):
  signatures      <- generate_minhash_value      EN:56-64 / ZH-1:23-29
  band_keys       <- generate_bands              EN:66-71 / ZH-1:30-33
  bucket_edges    <- groupBy + generate_edges    EN:136,73-80 / ZH-2:19-40
  components      <- large/small-star loop       EN:82-115,138-144 / ZH-3:11-27
  removed_mask    <- remove_list                 EN:145-149 / ZH-4:17-22
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as _lib

# reference constants (EN:17-20, ZH-1:14-17)
WIDTH = 13
BANDS = 9
ROWS = 13
NUM_PERM = 128
SEED = 1


TEXT_PAD = 128      # MDD_TEXT_PAD (include/mdd.h)


def padded_size(text_bytes: int) -> int:
    """Capacity a text buffer needs for `text_bytes` of content."""
    return ((int(text_bytes) + 3) // 4) * 4 + TEXT_PAD


def as_u32(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy().view(np.uint32)


def as_u64(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy().view(np.uint64)


def pack_texts(texts: Iterable) -> Tuple[np.ndarray, np.ndarray]:
    """str/bytes documents -> (uint8 text + TEXT_PAD zero bytes, uint64 offsets[n+1]).
    A str that cannot be utf-8 encoded raises (the reference skips such docs: ZH-1:52-54;
    callers that want skip-and-keep semantics filter before packing, as the CLI drivers do)."""
    enc = [t.encode("utf8") if isinstance(t, str) else bytes(t) for t in texts]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        offs[1:] = np.cumsum([len(e) for e in enc], dtype=np.uint64)
    total = int(offs[-1])
    buf = np.zeros(padded_size(total), dtype=np.uint8)
    buf[:total] = np.frombuffer(b"".join(enc), dtype=np.uint8)
    return buf, offs


def to_padded_device(eng, text: np.ndarray):
    """The packed text of a shard (host uint8 array) as the device buffer mdd_signatures wants: MDD_TEXT_PAD zero bytes
    behind it, 4-byte rounded.  On a GPU engine the padding is made on the device and the bytes are copied once -- a padded
    HOST copy first (np.zeros + assignment) was ~40 ms per 136 MB shard on the thread that feeds the GPU.  Other engines
    (the CPU backend of the tests) get the padded host array through their own to_device."""
    n = int(text.size)
    dev = getattr(eng, "device", None)
    if not isinstance(dev, torch.device) or dev.type != "cuda":
        buf = np.zeros(padded_size(n), dtype=np.uint8)
        buf[:n] = text
        return eng.to_device(buf)
    out = torch.zeros(padded_size(n), dtype=torch.uint8, device=dev)
    if n:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", UserWarning)      # Arrow / frombuffer arrays are read-only: they are only read here
            src = torch.from_numpy(np.ascontiguousarray(text).view(np.uint8).reshape(-1))
        out[:n].copy_(src)
    return out


def make_perms(seed: int = SEED, num_perm: int = NUM_PERM) -> Tuple[np.ndarray, np.ndarray]:
    """Permutation table (datasketch _init_permutations, called per doc at EN:62 / ZH-1:28)
    computed once by libmdd_hip's own MT19937 (host code; needs no GPU)."""
    L = _lib.load()
    a = np.zeros(num_perm, np.uint64)
    b = np.zeros(num_perm, np.uint64)
    _lib.check(L.mdd_make_perms(seed, num_perm, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)),
               "mdd_make_perms")
    return a, b


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else C.c_void_p(0)


class Engine:
    """One context = one (GPU, stream).  Not thread-safe; one per rank."""

    def __init__(self, device: Optional[int] = None):
        self.L = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.MddError("no GPU visible to torch: map_neo_amd has no CPU fallback")
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.current_stream()
            h = C.c_void_p()
            _lib.check(self.L.mdd_ctx_create(self.device_index, C.c_void_p(self.stream.cuda_stream), C.byref(h)),
                       "mdd_ctx_create")
        self.ctx = h
        self._perm_cache = {}
        # bucket table slots per band record (rounded up to a power of two): load factor <= 1 / table_factor
        self.table_factor = float(os.environ.get("MDD_TABLE_FACTOR", "2.0"))
        # exact substring dedup: "table" (mdd_substr_mark), "sorted" (mdd_substr_mark_sorted) or "auto" (by size)
        self.substr_method = os.environ.get("MDD_SUBSTR_METHOD", "auto")
        # sorted method: prove most windows unique first (mdd_substr_candidates); "0" sorts every window (A/B, tests)
        self.substr_filter = os.environ.get("MDD_SUBSTR_FILTER", "1") != "0"
        self._exact_coll = 0

    def close(self):
        if getattr(self, "ctx", None):
            self.L.mdd_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_segment_bytes(self, seg_bytes: int = 0):
        """Documents longer than this are split across waves (0 = library default)."""
        _lib.check(self.L.mdd_ctx_set_segment_bytes(self.ctx, int(seg_bytes)), "mdd_ctx_set_segment_bytes")

    def set_filter_from(self, n_shingles: int = 0):
        """Permute-phase filter of the signature kernel: 0 = default, 1 = from the first shingle,
        0xFFFFFFFF = never.  Results are identical for every setting."""
        _lib.check(self.L.mdd_ctx_set_filter_from(self.ctx, int(n_shingles)), "mdd_ctx_set_filter_from")

    def sync(self):
        _lib.check(self.L.mdd_ctx_sync(self.ctx), "mdd_ctx_sync")

    def warm_up(self):
        """The first launch from libmdd_hip.so loads its code object (~0.1 s): do it now (one 64-element launch),
        e.g. while host threads are still parsing shards"""
        self.components_init(64)
        self.sync()

    # ------------------------------------------------------------------ helpers
    def empty(self, n, dtype):
        return torch.empty(int(n), dtype=dtype, device=self.device)

    def to_device(self, arr: np.ndarray) -> torch.Tensor:
        """numpy uint8/uint32/uint64 -> device tensor with the same bits."""
        arr = np.ascontiguousarray(arr)
        view = {np.dtype(np.uint8): np.uint8, np.dtype(np.uint32): np.int32, np.dtype(np.uint64): np.int64,
                np.dtype(np.int32): np.int32, np.dtype(np.int64): np.int64}[arr.dtype]
        return torch.from_numpy(arr.view(view)).to(self.device)

    def perms(self, seed: int, K: int):
        key = (seed, K)
        if key not in self._perm_cache:
            a, b = make_perms(seed, K)
            self._perm_cache[key] = (self.to_device(a), self.to_device(b))
        return self._perm_cache[key]

    # ------------------------------------------------------------------ stages
    def signatures(self, text: torch.Tensor, offsets: torch.Tensor, num_perm: int = NUM_PERM,
                   width: int = WIDTH, unit: str = "char", seed: int = SEED,
                   perms: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                   text_bytes: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """text: uint8 device tensor of capacity >= padded_size(text_bytes);
        offsets: int64 device tensor [n+1].  Returns int32 tensor [n, K] holding u32 bits."""
        assert text.dtype == torch.uint8 and offsets.dtype == torch.int64
        n = offsets.numel() - 1
        if text_bytes is None:
            text_bytes = int(offsets[-1].item()) if n >= 0 and offsets.numel() else 0
        cap = text.numel()
        if int(text_bytes) + TEXT_PAD > cap:
            raise _lib.MddError(f"text tensor must extend {TEXT_PAD} bytes past text_bytes (minhash.padded_size)")
        a, b = perms if perms is not None else self.perms(seed, num_perm)
        sig = out if out is not None else torch.empty((n, num_perm), dtype=torch.int32, device=self.device)
        _lib.check(self.L.mdd_signatures(self.ctx, _ptr(text), int(text_bytes), _ptr(offsets), n,
                                         _lib.MDD_UNIT[unit], width, num_perm, _ptr(a), _ptr(b), _ptr(sig)),
                   "mdd_signatures")
        return sig

    def signatures_from_host(self, text_host: torch.Tensor, offsets: torch.Tensor, num_perm: int = NUM_PERM,
                             width: int = WIDTH, unit: str = "char", seed: int = SEED, text_bytes: Optional[int] = None,
                             text_dev: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                             copy_stream: Optional["torch.cuda.Stream"] = None, chunk_bytes: int = 0):
        """Signatures of a corpus whose text lives in PINNED host memory (uint8 tensor): one kernel launch hashes the
        documents while the text is copied over PCIe in chunks on `copy_stream` (mdd_signatures_from_host).
        offsets: int64 device tensor [n+1].  Returns (sig int32 [n, K], text_dev uint8 [padded])."""
        assert text_host.dtype == torch.uint8 and not text_host.is_cuda and text_host.is_pinned()
        assert offsets.dtype == torch.int64 and offsets.is_cuda
        n = offsets.numel() - 1
        if text_bytes is None:
            text_bytes = int(offsets[-1].item()) if n > 0 else 0
        if text_host.numel() < int(text_bytes):
            raise _lib.MddError("text_host is shorter than text_bytes")
        if text_dev is None:
            text_dev = torch.empty(padded_size(text_bytes), dtype=torch.uint8, device=self.device)
        if text_dev.numel() < int(text_bytes) + TEXT_PAD:
            raise _lib.MddError(f"text_dev must extend {TEXT_PAD} bytes past text_bytes (minhash.padded_size)")
        if copy_stream is None:
            if getattr(self, "_copy_stream", None) is None:
                self._copy_stream = torch.cuda.Stream(device=self.device)
            copy_stream = self._copy_stream
        a, b = self.perms(seed, num_perm)
        sig = out if out is not None else torch.empty((n, num_perm), dtype=torch.int32, device=self.device)
        _lib.check(self.L.mdd_signatures_from_host(self.ctx, C.c_void_p(copy_stream.cuda_stream), C.c_void_p(text_host.data_ptr()),
                                                   int(text_bytes), _ptr(text_dev), _ptr(offsets), n, _lib.MDD_UNIT[unit], width,
                                                   num_perm, _ptr(a), _ptr(b), _ptr(sig), int(chunk_bytes)),
                   "mdd_signatures_from_host")
        return sig, text_dev

    def stream_wait_ms(self) -> float:
        """wave-milliseconds the last signatures_from_host kernel spent waiting for text (summed over its waves)"""
        t = C.c_uint64(0)
        _lib.check(self.L.mdd_stream_wait_ticks(self.ctx, C.byref(t)), "mdd_stream_wait_ticks")
        return t.value / 1e5

    def shingle_hashes(self, text: torch.Tensor, text_bytes: int, doc_begin: int, doc_end: int,
                       width: int = WIDTH, unit: str = "char") -> np.ndarray:
        cap = max(1, doc_end - doc_begin)
        out = self.empty(cap, torch.int32)
        n = torch.zeros(1, dtype=torch.int64, device=self.device)
        _lib.check(self.L.mdd_shingle_hashes(self.ctx, _ptr(text), int(text_bytes), doc_begin, doc_end,
                                             _lib.MDD_UNIT[unit], width, _ptr(out), cap, _ptr(n)),
                   "mdd_shingle_hashes")
        return as_u32(out)[: int(n.item())]

    def band_keys(self, sig: torch.Tensor, bands: int = BANDS, rows: int = ROWS, doc_base: int = 0,
                  key_bits: int = 64):
        n, K = sig.shape
        key = self.empty(n * bands, torch.int64)
        meta = self.empty(n * bands, torch.int64)
        _lib.check(self.L.mdd_band_keys(self.ctx, _ptr(sig), n, K, bands, rows, doc_base, key_bits,
                                        _ptr(key), _ptr(meta)), "mdd_band_keys")
        return key, meta

    def packed_keys(self, tuples: torch.Tensor, meta: torch.Tensor, rows: int, key_bits: int = 64):
        n_rec = meta.numel()
        key = self.empty(n_rec, torch.int64)
        _lib.check(self.L.mdd_packed_keys(self.ctx, _ptr(tuples), _ptr(meta), n_rec, rows, key_bits, _ptr(key)),
                   "mdd_packed_keys")
        return key

    def route(self, sig: torch.Tensor, key: torch.Tensor, meta: torch.Tensor, bands: int, rows: int, world: int):
        """Group band records by destination rank ((key>>32) % world) in PACKED layout.
        Returns (tuples int32[n_rec*rows], meta int64[n_rec], counts int64[world] on the DEVICE)."""
        n, K = sig.shape
        n_rec = key.numel()
        counts = torch.zeros(world, dtype=torch.int64, device=self.device)
        _lib.check(self.L.mdd_route_count(self.ctx, _ptr(key), n_rec, world, _ptr(counts)), "mdd_route_count")
        cursor = torch.cumsum(counts, 0) - counts          # exclusive prefix (plumbing, world elements)
        tuples = self.empty(n_rec * rows, torch.int32)
        meta_out = self.empty(n_rec, torch.int64)
        _lib.check(self.L.mdd_route_pack(self.ctx, _ptr(sig), K, bands, rows, _ptr(key), _ptr(meta), n_rec, world,
                                         _ptr(cursor), _ptr(tuples), _ptr(meta_out)), "mdd_route_pack")
        return tuples, meta_out, counts

    # ------------------------------------------------------------------ multi-GPU exchange (mdd_comm.hip)
    def comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * 128)()
        _lib.check(self.L.mdd_comm_unique_id(buf), "mdd_comm_unique_id")
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        """collective: every rank of the job calls it with rank 0's unique id"""
        assert len(unique_id) == 128
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _lib.check(self.L.mdd_comm_init(self.ctx, buf, int(rank), int(world)), "mdd_comm_init")
        self.comm_rank, self.comm_world = int(rank), int(world)

    def comm_destroy(self):
        _lib.check(self.L.mdd_comm_destroy(self.ctx), "mdd_comm_destroy")

    def exchange_bands(self, tuples_s: torch.Tensor, meta_s: torch.Tensor, send_counts: torch.Tensor, rows: int):
        """band records grouped by owner (route) -> the records this rank owns; send_counts: int64 DEVICE tensor [world].
        Returns (tuples int32 [n*rows], meta int64 [n], send counts, recv counts (python lists))."""
        W = self.comm_world
        sc_h = (C.c_uint64 * W)()
        rc_h = (C.c_uint64 * W)()
        _lib.check(self.L.mdd_exchange_counts(self.ctx, _ptr(send_counts), sc_h, rc_h), "mdd_exchange_counts")
        n_recv = sum(rc_h)
        tuples_r = self.empty(n_recv * rows, torch.int32)
        meta_r = self.empty(n_recv, torch.int64)
        _lib.check(self.L.mdd_exchange_bands(self.ctx, _ptr(tuples_s), _ptr(meta_s), sc_h, rc_h, rows, _ptr(tuples_r), _ptr(meta_r)),
                   "mdd_exchange_bands")
        return tuples_r, meta_r, list(sc_h), list(rc_h)

    def allgather_edges(self, eu: torch.Tensor, ev: torch.Tensor):
        """edges of every rank, rank by rank (int32 device tensors)"""
        W = self.comm_world
        n_local = torch.tensor([eu.numel()], dtype=torch.int64, device=self.device)
        cnt = (C.c_uint64 * W)()
        _lib.check(self.L.mdd_allgather_counts(self.ctx, _ptr(n_local), cnt), "mdd_allgather_counts")
        total = sum(cnt)
        all_u, all_v = self.empty(total, torch.int32), self.empty(total, torch.int32)
        _lib.check(self.L.mdd_allgather_edges(self.ctx, _ptr(eu), _ptr(ev), cnt, _ptr(all_u), _ptr(all_v)), "mdd_allgather_edges")
        return all_u, all_v

    def bucket_edges(self, key: torch.Tensor, meta: torch.Tensor, tuple_base: torch.Tensor, rec_per_doc: int,
                     doc_stride: int, band_stride: int, rows: int, want_buckets: bool = False):
        """Returns (edge_u, edge_v int32 device tensors trimmed to n_edges, number of surrogate-tag collisions
        met (statistics: buckets are exact), rec_bucket or None)."""
        n_rec = key.numel()
        tab_size = 2
        while tab_size < self.table_factor * max(1, n_rec):
            tab_size *= 2
        table = self.empty(tab_size, torch.int64)
        eu = self.empty(max(1, n_rec), torch.int32)
        ev = self.empty(max(1, n_rec), torch.int32)
        cnt = torch.zeros(2, dtype=torch.int64, device=self.device)
        rb = self.empty(n_rec, torch.int64) if want_buckets else None
        _lib.check(self.L.mdd_bucket_edges(self.ctx, _ptr(key), _ptr(meta), n_rec, _ptr(tuple_base), rec_per_doc,
                                           doc_stride, band_stride, rows, _ptr(table), tab_size,
                                           _ptr(eu), _ptr(ev), n_rec, C.c_void_p(cnt.data_ptr()),
                                           C.c_void_p(cnt.data_ptr() + 8), _ptr(rb)), "mdd_bucket_edges")
        n_edges, n_coll = (int(x) for x in cnt.cpu().tolist())
        return eu[:n_edges], ev[:n_edges], n_coll, rb

    def components_init(self, n_docs: int) -> torch.Tensor:
        labels = self.empty(n_docs, torch.int32)
        _lib.check(self.L.mdd_components_init(self.ctx, _ptr(labels), n_docs), "mdd_components_init")
        return labels

    def components_union(self, labels: torch.Tensor, eu: torch.Tensor, ev: torch.Tensor):
        assert eu.numel() == ev.numel()
        _lib.check(self.L.mdd_components_union(self.ctx, _ptr(labels), labels.numel(), _ptr(eu), _ptr(ev), eu.numel()),
                   "mdd_components_union")

    def components_finish(self, labels: torch.Tensor):
        _lib.check(self.L.mdd_components_finish(self.ctx, _ptr(labels), labels.numel()), "mdd_components_finish")

    def signature_matches(self, sig: torch.Tensor, eu: torch.Tensor, ev: torch.Tensor) -> torch.Tensor:
        """int32 [n_edges]: equal signature components of the two documents of every edge (K x MinHash.jaccard)."""
        n, K = sig.shape
        assert eu.numel() == ev.numel()
        out = self.empty(max(1, eu.numel()), torch.int32)
        _lib.check(self.L.mdd_signature_matches(self.ctx, _ptr(sig), n, K, _ptr(eu), _ptr(ev), eu.numel(), _ptr(out)),
                   "mdd_signature_matches")
        return out[: eu.numel()]

    def forest_edges(self, labels: torch.Tensor, max_edges: int):
        """(d, labels[d]) for every non-root d of flattened labels; max_edges bounds the output."""
        cap = max(1, int(max_edges))
        eu, ev = self.empty(cap, torch.int32), self.empty(cap, torch.int32)
        cnt = torch.zeros(1, dtype=torch.int64, device=self.device)
        _lib.check(self.L.mdd_forest_edges(self.ctx, _ptr(labels), labels.numel(), _ptr(eu), _ptr(ev), cap, _ptr(cnt)),
                   "mdd_forest_edges")
        n = min(cap, int(cnt.item()))
        return eu[:n], ev[:n]

    def removed_mask(self, labels: torch.Tensor, first: int, n: int):
        removed = self.empty(n, torch.uint8)
        cnt = torch.zeros(1, dtype=torch.int64, device=self.device)
        _lib.check(self.L.mdd_removed_mask(self.ctx, _ptr(labels), first, n, _ptr(removed), _ptr(cnt)),
                   "mdd_removed_mask")
        return removed, cnt

    @property
    def last_exact_collisions(self) -> int:
        """tag matches of the last exact_group call that were different bytes (read lazily: no synchronisation in the call)"""
        c = self._exact_coll
        return int(c.item()) if isinstance(c, torch.Tensor) else int(c)

    @last_exact_collisions.setter
    def last_exact_collisions(self, v):
        self._exact_coll = v

    def exact_group(self, text: torch.Tensor, ubeg: torch.Tensor, uend: torch.Tensor, key_bits: int = 64,
                    text_host: Optional[np.ndarray] = None, avg_unit_bytes: Optional[int] = None, want_cps: bool = False):
        """first_of[u] = smallest unit index with bytes identical to unit u (int32 tensor, u32 bits).  Exact: hash
        tag matches are confirmed on the bytes inside the kernels (`last_exact_collisions` counts the false ones).
        `text_host` is accepted for compatibility and unused.  avg_unit_bytes: picks the hash kernel (computed from the
        ranges, with a synchronisation, when not given).  want_cps: also return the units' code-point counts (int32)."""
        n = ubeg.numel()
        first_of = self.empty(n, torch.int32)
        cps = self.empty(n, torch.int32) if want_cps else None
        self.last_exact_collisions = 0
        if n == 0:
            return (first_of, cps) if want_cps else first_of
        if avg_unit_bytes is None:
            avg_unit_bytes = max(0, int((uend - ubeg).sum().item())) // n
        tab_size = 2
        while tab_size < 2 * n:
            tab_size *= 2
        key = self.empty(n, torch.int64)
        table = self.empty(tab_size, torch.int64)
        cnt = torch.zeros(1, dtype=torch.int64, device=self.device)
        tb = max(0, text.numel() - TEXT_PAD)
        if want_cps:
            _lib.check(self.L.mdd_exact_group_cps(self.ctx, _ptr(text), tb, _ptr(ubeg), _ptr(uend), n, avg_unit_bytes, key_bits, _ptr(key),
                                                  _ptr(table), tab_size, _ptr(first_of), _ptr(cnt), _ptr(cps)), "mdd_exact_group_cps")
        else:
            _lib.check(self.L.mdd_exact_group(self.ctx, _ptr(text), tb, _ptr(ubeg), _ptr(uend), n, avg_unit_bytes, key_bits, _ptr(key),
                                              _ptr(table), tab_size, _ptr(first_of), _ptr(cnt)), "mdd_exact_group")
        self.last_exact_collisions = cnt
        return (first_of, cps) if want_cps else first_of

    def paragraph_dedup(self, text: torch.Tensor, offsets: torch.Tensor, min_len: int = 20, key_bits: int = 64):
        """Paragraph dedup of a packed corpus (PAR:15-30), documents in (id, input) order.  Returns
        (out_text uint8 [padded], out_offsets int64 [n_kept+1], kept document indices int64)."""
        n = offsets.numel() - 1
        if n <= 0:
            return (torch.zeros(padded_size(0), dtype=torch.uint8, device=self.device),
                    torch.zeros(1, dtype=torch.int64, device=self.device), torch.zeros(0, dtype=torch.int64, device=self.device))
        text_bytes = max(0, text.numel() - TEXT_PAD)
        per_doc = torch.zeros(n, dtype=torch.int64, device=self.device)
        _lib.check(self.L.mdd_para_count(self.ctx, _ptr(text), text_bytes, _ptr(offsets), n, _ptr(per_doc)), "mdd_para_count")
        base = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(per_doc, 0, out=base[1:])
        n_units = int(base[-1].item())
        ubeg = self.empty(n_units, torch.int64)
        uend = self.empty(n_units, torch.int64)
        _lib.check(self.L.mdd_para_units(self.ctx, _ptr(text), text_bytes, _ptr(offsets), n, _ptr(base), n_units, _ptr(ubeg),
                                         _ptr(uend)), "mdd_para_units")
        # the grouping's hash pass counts the paragraphs' code points on the way, so the per-document sums come from the
        # units (~28 bytes per paragraph) instead of another pass over the text
        first_of, ucps = self.exact_group(text, ubeg, uend, key_bits, avg_unit_bytes=text_bytes // max(1, n_units), want_cps=True)
        kept_b = torch.zeros(n, dtype=torch.int64, device=self.device)
        kept_c = torch.zeros(n, dtype=torch.int64, device=self.device)
        has_unit = torch.zeros(n, dtype=torch.uint8, device=self.device)
        whole = torch.zeros(n, dtype=torch.uint8, device=self.device)
        _lib.check(self.L.mdd_para_doc_sums(self.ctx, _ptr(offsets), n, text_bytes, _ptr(base), n_units, _ptr(ubeg), _ptr(uend),
                                            _ptr(ucps), _ptr(first_of), _ptr(kept_b), _ptr(kept_c), _ptr(has_unit), _ptr(whole)),
                   "mdd_para_doc_sums")
        del ubeg, uend, ucps
        keep_doc = has_unit != 0
        if min_len > 0:
            keep_doc &= kept_c >= min_len
        out_offs_all = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(kept_b * keep_doc, 0, out=out_offs_all[1:])
        total = int(out_offs_all[-1].item())
        out_text = torch.zeros(padded_size(total), dtype=torch.uint8, device=self.device)
        keep_u8 = keep_doc.to(torch.uint8)
        _lib.check(self.L.mdd_para_compact_write(self.ctx, _ptr(text), text_bytes, _ptr(offsets), n, _ptr(base), n_units,
                                                 _ptr(first_of), _ptr(keep_u8), _ptr(out_offs_all), _ptr(out_text), total, _ptr(whole)),
                   "mdd_para_compact_write")
        docs = torch.nonzero(keep_doc).flatten()
        out_offs = torch.cat([out_offs_all[:-1][docs], out_offs_all[-1:]])
        return out_text, out_offs, docs

    SUBSTR_WS_WORDS = 11672587      # MDD_SUBSTR_SORTED_WS_WORDS

    def substr_dedup(self, text: torch.Tensor, offsets: torch.Tensor, window: int = 50, min_length: int = 20,
                     table_bytes: Optional[int] = None, method: str = "auto"):
        """Exact substring dedup of a packed corpus (SUB:34-95).  Returns (out_text uint8 [padded],
        out_offsets int64 [n_kept+1], kept document indices int64).  The window table (16 bytes per window)
        is limited to table_bytes (default: 70 % of the free HBM); beyond that the windows are processed in
        several passes partitioned by content key."""
        n = offsets.numel() - 1
        if n <= 0:
            return (torch.zeros(padded_size(0), dtype=torch.uint8, device=self.device),
                    torch.zeros(1, dtype=torch.int64, device=self.device), torch.zeros(0, dtype=torch.int64, device=self.device))
        text_bytes = max(0, text.numel() - TEXT_PAD)
        lens = offsets[1:] - offsets[:-1]
        n_win = int(torch.clamp(lens - window, min=0).sum().item())        # code points <= bytes: an upper bound
        dup = self.empty(max(1, text_bytes), torch.uint8)
        if table_bytes is None:
            # the workspace is released before the outputs are allocated (and the caching allocator hands its blocks on), so most
            # of the free memory can go to it: fewer passes, and every pass re-walks all windows
            free = torch.cuda.mem_get_info(self.device)[0] + torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)
            table_bytes = int(free * float(os.environ.get("MDD_SUBSTR_MEM_FRAC", "0.7")))   # (allocator-cached blocks are free too)
        # "sorted": partition the windows into leaves and group them in LDS (mdd_substr_mark_sorted) -- pays once the
        # table of the "table" method (one random atomic per window) is far beyond the caches; "table": mdd_substr_mark
        if method == "auto":
            method = self.substr_method
        if method == "auto":
            method = "sorted" if (n_win >= (1 << 24) and text_bytes < (1 << 36)) else "table"
        self.last_substr_candidates = None
        if method == "sorted":
            slack = 2048 * 28 * torch.cuda.get_device_properties(self.device).multi_processor_count + 14336
            per_win = 2 * 8 + 2                                                  # two entry arrays + the u16 side array
            ws = self.empty(self.SUBSTR_WS_WORDS, torch.int64)
            cand = None
            n_sort = n_win                              # windows the passes have to sort
            if window >= 50 and self.substr_filter:
                # candidate filter (mdd_substr_candidates): only windows whose selected 32-byte k-gram occurs twice in the
                # corpus can repeat; one anchor per ~10 windows goes through the same sort first.  A sixth of the windows
                # is ample room for them (beyond it the filter switches itself off: every position comes back flagged).
                cap_a = max(n_win // 6, (text_bytes + 71) // 8) + slack      # (a sixth: at C4's 27 GB per GPU a quarter was 40 GB more to map)
                cap_a = min(cap_a, max(slack + (text_bytes + 71) // 8, table_bytes // per_win))
                ent_a, ent_b = self.empty(cap_a, torch.int64), self.empty(cap_a, torch.int64)
                d1 = self.empty(cap_a, torch.int16)
                cand = self.empty(((text_bytes >> 6) + 2) * 8, torch.uint8)      # MDD_SUBSTR_CAND_BYTES: a bit per text byte
                n_cand = torch.zeros(1, dtype=torch.int64, device=self.device)
                _lib.check(self.L.mdd_substr_candidates(self.ctx, _ptr(text), text_bytes, _ptr(offsets), n, window, _ptr(ent_a), _ptr(ent_b),
                                                        cap_a, _ptr(d1), _ptr(ws), ws.numel(), _ptr(cand), _ptr(n_cand), _ptr(dup)),
                           "mdd_substr_candidates")
                n_sort = min(n_win, int(n_cand.item()))
                self.last_substr_candidates = (n_sort, n_win)
                del ent_a, ent_b, d1
            if cand is not None and n_sort == 0:        # every repeated anchor was settled in its leaf (window 50): dup is final
                n_parts, ent_a, ent_b, d1, cap = 0, None, None, None, 0
            else:
                n_parts = max(1, -(-int(1.06 * n_sort * per_win + slack * per_win) // max(1, table_bytes)),
                              -(-n_sort // (1 << 33)))     # and at most 2^33 windows per pass (2^22 leaves of ~2000: 4-5 LDS rounds per leaf)
                cap = int(1.06 * n_sort / n_parts) + slack + (n_sort // n_parts // 16 if n_parts > 1 else 0)   # key partitions are only nearly equal
                ent_a, ent_b = self.empty(cap, torch.int64), self.empty(cap, torch.int64)
                d1 = self.empty(cap, torch.int16)
            for part in range(n_parts):
                _lib.check(self.L.mdd_substr_mark_sorted(self.ctx, _ptr(text), text_bytes, _ptr(offsets), n, window, part, n_parts,
                                                         _ptr(ent_a), _ptr(ent_b), cap, _ptr(d1), _ptr(ws), ws.numel(), _ptr(dup), _ptr(cand)),
                           "mdd_substr_mark_sorted")
            if os.environ.get("MDD_LEAF_PROF"):      # profiling builds (-DMDD_LEAF_PROF): cycles per leaf-kernel phase
                torch.cuda.synchronize(self.device)
                base = (2048 + 1) + 2048 + (2 ** 22 + 1) + 2 ** 22 + 1024 + 2 ** 21
                self.last_leaf_prof = ws[base + 1: base + 6].tolist()
            del ent_a, ent_b, d1, ws, cand  # the workspace (most of the free HBM) goes back before the outputs are allocated
        else:
            n_parts = max(1, -(-(16 * n_win) // max(1, table_bytes)))
            slots = 2 * n_win if n_parts == 1 else int(2.2 * n_win / n_parts) + 4096      # hash partitions are only nearly equal
            table = self.empty(max(2, slots), torch.int64)
            for part in range(n_parts):
                _lib.check(self.L.mdd_substr_mark(self.ctx, _ptr(text), text_bytes, _ptr(offsets), n, window, part, n_parts,
                                                  _ptr(table), table.numel(), _ptr(dup)), "mdd_substr_mark")
            del table
        self.last_substr_parts = n_parts
        self.last_substr_method = method
        _lib.check(self.L.mdd_substr_check(self.ctx, None), "mdd_substr_mark")      # an undersized table is an error, not a miss
        kept_b = torch.zeros(n, dtype=torch.int64, device=self.device)
        kept_c = torch.zeros(n, dtype=torch.int64, device=self.device)
        _lib.check(self.L.mdd_substr_count(self.ctx, _ptr(text), text_bytes, _ptr(offsets), n, window, _ptr(dup), _ptr(kept_b),
                                           _ptr(kept_c)), "mdd_substr_count")
        keep_doc = kept_c >= min_length
        out_offs_all = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(kept_b * keep_doc, 0, out=out_offs_all[1:])
        total = int(out_offs_all[-1].item())
        out_text = torch.zeros(padded_size(total), dtype=torch.uint8, device=self.device)
        keep_u8 = keep_doc.to(torch.uint8)
        _lib.check(self.L.mdd_substr_write(self.ctx, _ptr(text), text_bytes, _ptr(offsets), n, window, _ptr(dup), _ptr(keep_u8),
                                           _ptr(out_offs_all), _ptr(out_text), total), "mdd_substr_write")
        docs = torch.nonzero(keep_doc).flatten()
        out_offs = torch.cat([out_offs_all[:-1][docs], out_offs_all[-1:]])
        return out_text, out_offs, docs

    def signatures_en(self, text: torch.Tensor, offsets: torch.Tensor, num_perm: int = NUM_PERM, width: int = WIDTH,
                      unit: str = "char", seed: int = SEED):
        """generate_minhash_value after normalize_text (EN:31-40 + EN:56-64) for RAW documents in HBM: normalised on
        the device (normalize_en), hashed, and the few documents the device cannot vouch for (flags == 1: combining
        marks, composable / decomposable characters, U+0130, U+03A3, unassigned code points, invalid utf-8) -- all
        documents when ftfy is installed, it rewrites ASCII text too -- go through the Python steps
        (hostio.normalize_text) and are hashed in a second call.  Returns (sig int32 [n, K], number of host documents)."""
        from . import hostio
        n = offsets.numel() - 1
        n_text, n_offs, flags = self.normalize_en(text, offsets)
        sig = self.signatures(n_text, n_offs, num_perm, width, unit, seed)
        host = torch.arange(n, device=self.device) if hostio._ftfy is not None else torch.nonzero(flags).flatten()
        if host.numel():
            o = offsets.cpu().numpy()
            docs = []
            for d in host.cpu().tolist():
                raw = text[int(o[d]):int(o[d + 1])].cpu().numpy().tobytes()
                try:
                    docs.append(hostio.normalize_text(raw.decode("utf8")))
                except UnicodeDecodeError:      # not a str the reference could have held: hashed as the empty document
                    docs.append("")
            buf, ho = pack_texts(docs)
            sig[host] = self.signatures(self.to_device(buf), self.to_device(ho), num_perm, width, unit, seed)
        return sig, int(host.numel())

    def normalize_en(self, text: torch.Tensor, offsets: torch.Tensor):
        """EN:31-40 on a packed batch, on the device (ASCII stretches by byte classes, everything else through the
        generated Unicode table, csrc/mdd_unicode_tab.h).  Returns (out_text uint8 [padded], out_offsets int64 [n+1],
        flags uint8 [n]); documents with flags[d] == 1 hold a code point the device cannot vouch for (see
        signatures_en): they have zero length in the output and must be normalised by the caller."""
        n = offsets.numel() - 1
        out_len = torch.zeros(max(n, 0), dtype=torch.int64, device=self.device)
        flags = torch.zeros(max(n, 0), dtype=torch.uint8, device=self.device)
        text_bytes = max(0, text.numel() - TEXT_PAD)
        _lib.check(self.L.mdd_normalize_en_count(self.ctx, _ptr(text), text_bytes, _ptr(offsets), n, _ptr(out_len), _ptr(flags)),
                   "mdd_normalize_en_count")
        out_offs = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(out_len, 0, out=out_offs[1:])
        total = int(out_offs[-1].item()) if n > 0 else 0
        out_text = torch.zeros(padded_size(total), dtype=torch.uint8, device=self.device)
        _lib.check(self.L.mdd_normalize_en_write(self.ctx, _ptr(text), text_bytes, _ptr(offsets), n, _ptr(flags), _ptr(out_offs),
                                                 _ptr(out_text), total), "mdd_normalize_en_write")
        return out_text, out_offs, flags

    def gather_docs(self, text: torch.Tensor, offsets: torch.Tensor, sel: torch.Tensor):
        """Packed corpus whose document k is document sel[k] of (text, offsets); sel int64 on the device, any order,
        repeats allowed.  Returns (out_text uint8 [padded], out_offsets int64 [len(sel)+1])."""
        n = offsets.numel() - 1
        sel = sel.to(device=self.device, dtype=torch.int64).contiguous()
        m = sel.numel()
        out_offs = torch.zeros(m + 1, dtype=torch.int64, device=self.device)
        if m:
            if n <= 0 or int(sel.min().item()) < 0 or int(sel.max().item()) >= n:
                raise IndexError("gather_docs: sel outside [0, n_docs)")
            torch.cumsum((offsets[1:] - offsets[:-1])[sel], 0, out=out_offs[1:])
        total = int(out_offs[-1].item()) if m else 0
        out_text = torch.zeros(padded_size(total), dtype=torch.uint8, device=self.device)
        text_bytes = max(0, text.numel() - TEXT_PAD)
        _lib.check(self.L.mdd_gather_docs(self.ctx, _ptr(text), text_bytes, _ptr(offsets), max(n, 0), _ptr(sel), m, _ptr(out_offs),
                                          _ptr(out_text), total), "mdd_gather_docs")
        return out_text, out_offs

    JSONL_TILE = 16384      # MDD_JSONL_TILE

    def jsonl_extract(self, raw: torch.Tensor, raw_bytes: int, content_field: str, id_field: Optional[str] = None):
        """Field extraction of an UNCOMPRESSED JSON-lines shard whose bytes are in HBM (mdd_jsonl_*; ZH-1:40-46 / EN:120-123
        for the lines the device can vouch for).  raw: uint8 device tensor of capacity >= padded_size(raw_bytes).
        Returns a dict:
          n_lines, status (uint8 host array per physical line: 0 OK, 1 no content, 2 fallback -- run json.loads on it --, 3 blank),
          line_begin (uint64 host array [n_lines+1]), lone_cr (lone "\\r" bytes: a text-mode reader splits there), n_cr (all "\\r"),
          text / offsets (device: packed decoded content of the OK lines, int64 [n_ok+1]), ok_lines (int64 host array),
          id_kind / id_beg / id_end (host arrays per line; None without id_field).
        None when the field names need JSON escaping or exceed 64 bytes (MDD_ERR_UNSUPPORTED: use the host reader)."""
        assert raw.dtype == torch.uint8 and raw.is_cuda and raw.numel() >= int(raw_bytes) + TEXT_PAD
        raw_bytes = int(raw_bytes)
        n_tiles = max(1, -(-raw_bytes // self.JSONL_TILE))
        tile_counts = torch.zeros(n_tiles, dtype=torch.int32, device=self.device)
        totals = torch.zeros(3, dtype=torch.int64, device=self.device)
        _lib.check(self.L.mdd_jsonl_count_lines(self.ctx, _ptr(raw), raw_bytes, _ptr(tile_counts), _ptr(totals)), "mdd_jsonl_count_lines")
        n_nl, lone_cr, n_cr = (int(x) for x in totals.cpu().tolist())
        last_open = raw_bytes > 0 and int(raw[raw_bytes - 1].item()) != 10
        n_lines = n_nl + (1 if last_open else 0)
        tile_base = torch.cumsum(tile_counts.to(torch.int64), 0) - tile_counts      # exclusive prefix (plumbing)
        line_begin = torch.zeros(n_lines + 1, dtype=torch.int64, device=self.device)
        _lib.check(self.L.mdd_jsonl_line_begins(self.ctx, _ptr(raw), raw_bytes, _ptr(tile_base), C.c_void_p(line_begin.data_ptr()), n_lines),
                   "mdd_jsonl_line_begins")
        status = torch.empty(n_lines, dtype=torch.uint8, device=self.device)
        cpos = torch.empty(n_lines, dtype=torch.int64, device=self.device)
        clen = torch.empty(n_lines, dtype=torch.int64, device=self.device)
        idk = idb = ide = None
        if id_field:
            idk = torch.empty(n_lines, dtype=torch.uint8, device=self.device)
            idb = torch.empty(n_lines, dtype=torch.int64, device=self.device)
            ide = torch.empty(n_lines, dtype=torch.int64, device=self.device)
        rc = self.L.mdd_jsonl_scan(self.ctx, _ptr(raw), raw_bytes, _ptr(line_begin), n_lines, content_field.encode("utf8"),
                                   id_field.encode("utf8") if id_field else None, _ptr(status), _ptr(cpos), _ptr(clen), _ptr(idk),
                                   _ptr(idb), _ptr(ide))
        if rc == -3:                                    # MDD_ERR_UNSUPPORTED: key needs escaping / too long
            return None
        _lib.check(rc, "mdd_jsonl_scan")
        out_offs = torch.zeros(n_lines + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(clen, 0, out=out_offs[1:])
        total = int(out_offs[-1].item()) if n_lines else 0
        text = torch.zeros(padded_size(total), dtype=torch.uint8, device=self.device)
        _lib.check(self.L.mdd_jsonl_decode(self.ctx, _ptr(raw), raw_bytes, n_lines, _ptr(status), _ptr(cpos), _ptr(out_offs), _ptr(text),
                                           total), "mdd_jsonl_decode")
        ok = torch.nonzero(status == 0).flatten()
        offsets = torch.cat([out_offs[:-1][ok], out_offs[-1:]])
        res = {"n_lines": n_lines, "status": status.cpu().numpy(), "line_begin": line_begin.cpu().numpy().view(np.uint64),
               "lone_cr": lone_cr, "n_cr": n_cr, "text": text, "offsets": offsets, "text_bytes": total, "ok_lines": ok.cpu().numpy(),
               "id_kind": None, "id_beg": None, "id_end": None}
        if id_field:
            res.update(id_kind=idk.cpu().numpy(), id_beg=idb.cpu().numpy().view(np.uint64), id_end=ide.cpu().numpy().view(np.uint64))
        return res

    def valu_calibrate(self, iters: int = 20000) -> float:
        out = C.c_double(0.0)
        _lib.check(self.L.mdd_valu_calibrate(self.ctx, iters, C.byref(out)), "mdd_valu_calibrate")
        return out.value


class HostPipeline:
    """Signatures for a corpus that lives in (pinned) HOST memory: documents are cut into
    batches, batch b+1 is copied over PCIe on a copy stream while batch b is hashed on the
    compute stream (two device staging buffers, events both ways), so the PCIe time hides
    behind the VALU-bound kernel.  One mdd context per stream (contexts are bound to a stream)."""

    def __init__(self, engine: "Engine", batch_bytes: int = 256 << 20):
        self.eng = engine
        self.batch_bytes = int(batch_bytes)
        self.copy_stream = torch.cuda.Stream(device=engine.device)

    def signatures(self, text_host: torch.Tensor, offsets_host: np.ndarray, num_perm: int = NUM_PERM,
                   width: int = WIDTH, unit: str = "char", seed: int = SEED) -> torch.Tensor:
        eng = self.eng
        offs = np.ascontiguousarray(offsets_host).astype(np.int64)
        n = offs.shape[0] - 1
        sig = torch.empty((n, num_perm), dtype=torch.int32, device=eng.device)
        if n == 0:
            return sig
        # batch boundaries: whole documents, about batch_bytes each
        cuts = [0]
        while cuts[-1] < n:
            target = offs[cuts[-1]] + self.batch_bytes
            nxt = int(np.searchsorted(offs, target, side="right")) - 1
            cuts.append(min(n, max(nxt, cuts[-1] + 1)))
        max_bytes = max(int(offs[b] - offs[a]) for a, b in zip(cuts[:-1], cuts[1:]))
        max_docs = max(b - a for a, b in zip(cuts[:-1], cuts[1:]))
        bufs = [torch.zeros(padded_size(max_bytes), dtype=torch.uint8, device=eng.device) for _ in range(2)]
        obufs = [torch.empty(max_docs + 1, dtype=torch.int64, device=eng.device) for _ in range(2)]
        ohost = [torch.empty(max_docs + 1, dtype=torch.int64).pin_memory() for _ in range(2)]
        copied = [torch.cuda.Event() for _ in range(2)]
        consumed = [torch.cuda.Event() for _ in range(2)]
        a_d, b_d = eng.perms(seed, num_perm)
        compute = torch.cuda.current_stream(eng.device)
        host_free = [None, None]
        for i, (d0, d1) in enumerate(zip(cuts[:-1], cuts[1:])):
            s = i & 1
            lo, hi = int(offs[d0]), int(offs[d1])
            nd = d1 - d0
            if host_free[s] is not None:
                host_free[s].synchronize()            # pinned offsets staging of batch i-2 has been read
            ohost[s][: nd + 1].copy_(torch.from_numpy(offs[d0:d1 + 1] - lo))
            with torch.cuda.stream(self.copy_stream):
                if i >= 2:
                    self.copy_stream.wait_event(consumed[s])      # kernel of batch i-2 is done with this buffer
                bufs[s][: hi - lo].copy_(text_host[lo:hi], non_blocking=True)
                obufs[s][: nd + 1].copy_(ohost[s][: nd + 1], non_blocking=True)
                copied[s].record(self.copy_stream)
                host_free[s] = copied[s]
            compute.wait_event(copied[s])
            eng.signatures(bufs[s], obufs[s][: nd + 1], num_perm, width, unit, seed, perms=(a_d, b_d),
                           text_bytes=hi - lo, out=sig[d0:d1])
            consumed[s].record(compute)
        return sig


