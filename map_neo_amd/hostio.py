"""Host-side ingest/egress around the GPU path: jsonl(.gz) shards in,
jsonl(.gz) / pickle artifacts out.  Pure I/O and bookkeeping -- no hashing happens here."""
from __future__ import annotations

import gzip
import io
import json
import os
import re
import string
import unicodedata
from typing import Iterator, List, Optional, Tuple


def list_files(directory: str) -> List[str]:
    """Regular, non-hidden files of a directory in sorted order (the reference iterates
    os.listdir / Spark partitions in arbitrary order; sorted order is one legal instance)."""
    out = []
    for name in sorted(os.listdir(directory)):
        path = os.path.join(directory, name)
        if os.path.isfile(path) and not name.startswith((".", "_")):
            out.append(path)
    return out


def id_order_key(v):
    """Total order over the values a doc_id column can hold (JSON scalars and None), in the spirit of Spark's
    ascending sort: nulls first, then booleans / numbers by value, then strings; anything else by its JSON text.
    Plain `sorted` raises TypeError on a missing (None) id or on mixed int / str ids."""
    if v is None:
        return (0, 0, "")
    if isinstance(v, (bool, int, float)):
        return (1, v, "")
    if isinstance(v, str):
        return (2, 0, v)
    import json
    return (3, 0, json.dumps(v, sort_keys=True, default=str))


def shard_output_names(paths, suffix: str = ".jsonl.gz"):
    """Output file name for every input shard: the input's base name with only its format suffix replaced
    (.jsonl / .json / .parquet / .pq / .txt, optionally followed by .gz), so that `c4-train.00000-of-01024.json.gz` and
    `c4-train.00001-of-01024.json.gz` -- or `data.0.jsonl` and `data.1.jsonl` -- keep distinct names.  Names that
    still coincide (`a.jsonl` and `a.jsonl.gz`, equal names in different directories) get their position in `paths`
    as a prefix.  The result is unique by construction."""
    import re
    stems = [re.sub(r"(\.(jsonl|json|parquet|pq|txt))?(\.gz)?$", "", os.path.basename(p), flags=re.I) or os.path.basename(p) for p in paths]
    seen = {}
    for s in stems:
        seen[s] = seen.get(s, 0) + 1
    names = [(s if seen[s] == 1 else "%05d_%s" % (i, s)) + suffix for i, s in enumerate(stems)]
    assert len(set(names)) == len(names)
    return names


def open_text(path: str, mode: str = "rt"):
    """Text-mode READS use universal newlines like the reference's own `open(path, 'r')` (ZH-1:40, ZH-5:9) and Spark's
    line reader (EN:120): "\n", "\r\n" and a lone "\r" all end a line.  Writes never translate."""
    nl = None if "r" in mode else "\n"
    if path.endswith(".gz"):
        return gzip.open(path, mode, encoding="utf8", newline=nl) if "t" in mode else gzip.open(path, mode)
    return open(path, mode, encoding="utf8", newline=nl) if "t" in mode else open(path, mode)


def iter_lines(path: str) -> Iterator[str]:
    with open_text(path, "rt") as f:
        for line in f:
            yield line


_PUNCT = str.maketrans("", "", string.punctuation)
_WS = re.compile(r"\s+")
try:  # optional third-party fix-up used by the reference (EN:33); absent in this image
    import ftfy as _ftfy
except Exception:  # pragma: no cover
    _ftfy = None


def normalize_text(text: str) -> str:
    """EN:31-40: ftfy.fix_text(normalization="NFC") -> lower -> strip ASCII punctuation ->
    strip -> collapse whitespace runs to one space.  When ftfy is not installed only its NFC
    normalisation step is applied (stated in DESIGN.md; identity on already-clean text)."""
    text = _ftfy.fix_text(text, normalization="NFC") if _ftfy is not None else unicodedata.normalize("NFC", text)
    text = text.lower().translate(_PUNCT)
    return _WS.sub(" ", text.strip())


def encodable(text: str) -> bool:
    """Lone surrogates cannot be utf-8 encoded: the reference's per-line try/except skips such
    documents (ZH-1:52-54), i.e. they are never hashed and therefore always kept."""
    try:
        text.encode("utf8")
        return True
    except UnicodeEncodeError:
        return False


def iter_records(path: str, fields=None):
    """Rows of one shard as dicts: JSON-lines (optionally .gz) or Parquet (.parquet / .pq, via
    pyarrow; `fields` restricts the columns read).  Blank JSON lines are skipped; a JSON line
    that does not parse is yielded as None so that the caller can count/skip it while keeping
    line numbers aligned with the file."""
    low = path.lower()
    if low.endswith((".parquet", ".pq")):
        import pyarrow.parquet as pq
        pf = pq.ParquetFile(path)
        cols = [c for c in fields if c in pf.schema_arrow.names] if fields else None
        for batch in pf.iter_batches(batch_size=65536, columns=cols):
            for row in batch.to_pylist():
                yield row
        return
    import json
    for line in iter_lines(path):
        if not line.strip():
            continue
        try:
            yield json.loads(line)
        except ValueError:
            yield None


# ------------------------------------------------------------------------------------------
# native JSON-lines ingest (libmdd_io.so, include/mdd_io.h) with per-line Python fallback
# ------------------------------------------------------------------------------------------
_IO_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmdd_io.so")
_io_lib = None


def _io():
    """libmdd_io.so or None (the pure-Python path is always available and equivalent)."""
    global _io_lib
    if _io_lib is None:
        import ctypes as C
        if not os.path.exists(_IO_SO):
            _io_lib = False
            return None
        L = C.CDLL(_IO_SO)
        L.mdd_io_read_jsonl.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.mdd_io_read_jsonl.restype = C.c_int
        L.mdd_io_free.argtypes = [C.c_void_p]
        L.mdd_io_free.restype = None
        L.mdd_io_last_error.restype = C.c_char_p
        for name in ("n_lines", "raw_bytes", "n_docs"):
            f = getattr(L, "mdd_io_" + name)
            f.argtypes, f.restype = [C.c_void_p], C.c_uint64
        for name in ("status", "line_no", "line_begin", "line_end", "id_kind", "id_begin", "id_end", "raw", "text", "offsets"):
            f = getattr(L, "mdd_io_" + name)
            f.argtypes, f.restype = [C.c_void_p], C.c_void_p
        L.mdd_io_write_jsonl.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                         C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
        L.mdd_io_write_jsonl.restype = C.c_int
        _io_lib = L
    return _io_lib or None


_SCALAR_IDS = (str, int, float, bool, type(None))


def write_jsonl_part(path: str, id_field: str, content_field: str, ids, raw: np.ndarray, offsets: np.ndarray,
                     gzip_level: int = 6, native: bool = True) -> int:
    """One output part of the exact-match stages: line k = json.dumps({id_field: ids[k], content_field:
    raw[offsets[k]:offsets[k+1]].decode("utf8")}, ensure_ascii=False) + "\n", gzip'ed (the columns PAR:27-28 /
    SUB:92 keep).  libmdd_io.so writes it without the GIL (callers run parts on threads); the Python loop below
    is the same thing and is what runs when the library is absent or native=False.  -> documents written"""
    import json
    import numpy as np
    n = len(ids)
    assert offsets.shape[0] == n + 1
    L = _io() if native else None
    if L is None:
        with (gzip.open(path, "wt", encoding="utf8", compresslevel=gzip_level, newline="\n") if gzip_level else
              open(path, "w", encoding="utf8", newline="\n")) as f:
            mv = memoryview(raw)
            for k, rid in enumerate(ids):
                row = {id_field: rid, content_field: bytes(mv[int(offsets[k]): int(offsets[k + 1])]).decode("utf8")}
                f.write(json.dumps(row, ensure_ascii=False) + "\n")
        return n
    import ctypes as C
    if all(type(x) in _SCALAR_IDS for x in ids):
        # one C-encoder call for all ids; a raw newline cannot occur inside an encoded scalar, so it separates them
        blob = json.dumps(list(ids), ensure_ascii=False, separators=("\n", ": "))[1:-1].encode("utf8") if n else b""
    else:
        blob = "\n".join(json.dumps(x, ensure_ascii=False) for x in ids).encode("utf8")
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    if n and int(offsets[-1]) > raw.size:
        raise ValueError("write_jsonl_part: offsets run past the text")
    written = C.c_uint64(0)
    rc = L.mdd_io_write_jsonl(path.encode(), json.dumps(id_field, ensure_ascii=False).encode("utf8"),
                              json.dumps(content_field, ensure_ascii=False).encode("utf8"), blob, len(blob),
                              raw.ctypes.data, offsets.ctypes.data, n, gzip_level, C.byref(written))
    if rc == -3:
        raise UnicodeDecodeError("utf-8", b"", 0, 1, L.mdd_io_last_error().decode("utf8", "replace"))
    if rc != 0:
        raise OSError(L.mdd_io_last_error().decode("utf8", "replace"))
    return n


def _py_line(line: str, content_field: str, id_field, normalize_en: bool, on_error: str):
    """The reference's own steps on one line -> (row_ok, id, content bytes or None)."""
    import json
    try:
        row = json.loads(line)
    except ValueError:
        if on_error == "raise":
            raise
        return False, None, None
    if not isinstance(row, dict):
        if on_error == "raise":
            raise TypeError("JSON line is not an object")
        return False, None, None
    rid = row.get(id_field) if id_field else None
    text = row.get(content_field)
    if isinstance(text, str) and encodable(text):
        return True, rid, (normalize_text(text) if normalize_en else text).encode("utf8")
    return True, rid, None


def read_jsonl_shard(path: str, content_field: str, id_field=None, normalize_en: bool = False,
                     on_error: str = "skip", use_native: bool = True):
    """One JSON-lines shard (optionally gzip) -> dict with
      ids      list, one entry per parsed row (None when the id field is absent / null)
      line_no  uint64 [n_rows], physical 0-based line number of each row (ZH-1:42 `enumerate(f)`)
      cand     int64 [n_docs], row index of every document that will be hashed
      text     uint8, packed utf-8 of those documents (normalised when normalize_en)
      offsets  uint64 [n_docs+1]
      n_bad    lines that do not parse (skipped when on_error == "skip", as ZH-1:52-54 does)
    The native scanner (libmdd_io.so) handles the lines it is sure about; every other line goes
    through `_py_line`, i.e. json.loads + normalize_text exactly as the pure-Python path does, so
    both paths return identical results (tests/test_hostio.py).  With ftfy installed the EN
    normalisation is always done in Python (ftfy rewrites ASCII text too)."""
    import ctypes as C
    import json

    import numpy as np
    L = _io() if use_native else None
    if L is None:
        ids, line_no, cand, enc, n_bad = [], [], [], [], 0
        for k, line in enumerate(iter_lines(path)):
            if not line.strip():
                continue
            ok, rid, data = _py_line(line, content_field, id_field, normalize_en, on_error)
            if not ok:
                n_bad += 1
                continue
            if data is not None:
                cand.append(len(ids))
                enc.append(data)
            ids.append(rid)
            line_no.append(k)
        offs = np.zeros(len(enc) + 1, dtype=np.uint64)
        if enc:
            offs[1:] = np.cumsum([len(e) for e in enc], dtype=np.uint64)
        return {"ids": ids, "line_no": np.asarray(line_no, dtype=np.uint64), "cand": np.asarray(cand, dtype=np.int64),
                "text": np.frombuffer(b"".join(enc), dtype=np.uint8), "offsets": offs, "n_bad": n_bad}

    native_norm = normalize_en and _ftfy is None
    h = C.c_void_p()
    rc = L.mdd_io_read_jsonl(path.encode(), content_field.encode(), id_field.encode() if id_field else None,
                             1 if native_norm else 0, C.byref(h))
    if rc != 0:
        raise OSError(L.mdd_io_last_error().decode("utf8", "replace"))
    try:
        n = int(L.mdd_io_n_lines(h))

        def arr(name, dtype, count):
            ptr = getattr(L, "mdd_io_" + name)(h)
            if not ptr or count == 0:
                return np.zeros(0, dtype=dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8 if dtype == np.uint8 else C.c_uint64)),
                                         shape=(count,)).copy()
        status = arr("status", np.uint8, n)
        line_no = arr("line_no", np.uint64, n)
        lb, le = arr("line_begin", np.uint64, n), arr("line_end", np.uint64, n)
        idk, idb, ide = arr("id_kind", np.uint8, n), arr("id_begin", np.uint64, n), arr("id_end", np.uint64, n)
        raw = arr("raw", np.uint8, int(L.mdd_io_raw_bytes(h))).tobytes()
        nd = int(L.mdd_io_n_docs(h))
        offs = arr("offsets", np.uint64, nd + 1)
        text = arr("text", np.uint8, int(offs[-1]) if nd else 0)
    finally:
        L.mdd_io_free(h)
    # The reference iterates its shards in TEXT mode (EN: Spark's line reader; ZH-1:40-42 `for idx, line in enumerate(f)`):
    # a lone "\r" ends a line there.  The native scanner numbers lines by "\n" alone -- identical for "\n" and "\r\n"
    # files ("\r" before "\n" is JSON whitespace); a shard with a lone "\r" takes the Python path, whose text-mode
    # iteration is the reference's.
    if raw.count(b"\r") != raw.count(b"\r\n"):
        return read_jsonl_shard(path, content_field, id_field, normalize_en, on_error, use_native=False)

    ids = [None] * n
    for i in np.flatnonzero(idk == 1).tolist():
        ids[i] = raw[idb[i]:ide[i]].decode("utf8")
    for i in np.flatnonzero(idk == 2).tolist():
        ids[i] = json.loads(raw[idb[i]:ide[i]])
    # documents the native pass produced, in line order; when normalisation must happen in Python
    # (ftfy installed) the native text is the raw field and is normalised here
    ok_rows = np.flatnonzero(status == 0)
    pieces = None
    if normalize_en and not native_norm:
        pieces = {int(r): normalize_text(text[int(offs[k]):int(offs[k + 1])].tobytes().decode("utf8")).encode("utf8")
                  for k, r in enumerate(ok_rows.tolist())}
    fb_rows = np.flatnonzero(status == 2)
    keep = np.ones(n, dtype=bool)
    fb_data, n_bad = {}, 0
    for i in fb_rows.tolist():
        try:
            line = raw[lb[i]:le[i]].decode("utf8")
            ok, rid, data = _py_line(line, content_field, id_field, normalize_en, on_error)
        except UnicodeDecodeError:
            if on_error == "raise":
                raise
            ok, rid, data = False, None, None
        if not ok:
            keep[i] = False
            n_bad += 1
            continue
        ids[i] = rid
        if data is not None:
            fb_data[i] = data
    if not fb_data and pieces is None and keep.all():
        cand = ok_rows.astype(np.int64)
        return {"ids": ids, "line_no": line_no, "cand": cand, "text": text, "offsets": offs, "n_bad": 0}
    # general merge: rebuild the packed text in row order
    doc_of_row = {int(r): k for k, r in enumerate(ok_rows.tolist())}
    new_index = np.cumsum(keep) - 1
    enc, cand = [], []
    for i in range(n):
        if not keep[i]:
            continue
        if i in doc_of_row:
            k = doc_of_row[i]
            enc.append(pieces[i] if pieces is not None else text[int(offs[k]):int(offs[k + 1])].tobytes())
            cand.append(int(new_index[i]))
        elif i in fb_data:
            enc.append(fb_data[i])
            cand.append(int(new_index[i]))
    offs2 = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        offs2[1:] = np.cumsum([len(e) for e in enc], dtype=np.uint64)
    return {"ids": [ids[i] for i in range(n) if keep[i]], "line_no": line_no[keep], "cand": np.asarray(cand, dtype=np.int64),
            "text": np.frombuffer(b"".join(enc), dtype=np.uint8), "offsets": offs2, "n_bad": n_bad}


def read_files_pinned(paths, pool=None):
    """The bytes of several plain (not gzip) files back to back in ONE pinned host tensor (torch's caching host allocator:
    the buffer is reused once freed), every file followed by a "\\n" (so a last line without one is closed; after a file that
    ends with "\\n" it is a blank line, which is no row).  Returns (uint8 tensor, starts int64 [n+1], is_gzip bool [n]): file
    k occupies [starts[k], starts[k+1]) including its added "\\n"; a gzip file's range is blanked (it goes to the host reader).
    `pool` (a ThreadPoolExecutor) reads the files side by side; the reads release the GIL."""
    import numpy as np
    import torch
    sizes = [os.path.getsize(p) for p in paths]
    starts = np.zeros(len(paths) + 1, dtype=np.int64)
    starts[1:] = np.cumsum([sz + 1 for sz in sizes])
    buf = torch.empty(max(int(starts[-1]), 1), dtype=torch.uint8, pin_memory=True)
    mv = memoryview(buf.numpy())
    is_gz = np.zeros(len(paths), dtype=bool)

    def one(k):
        lo, n = int(starts[k]), sizes[k]
        got = 0
        with open(paths[k], "rb") as f:
            while got < n:
                r = f.readinto(mv[lo + got: lo + n])
                if not r:
                    break
                got += r
        if got >= 2 and mv[lo] == 0x1f and mv[lo + 1] == 0x8b:
            is_gz[k] = True
            got = 0
        if got < n:                                   # gzip, or a file that shrank meanwhile: blank out the rest
            buf[lo + got: lo + n] = 32
        mv[lo + n] = 10
    if pool is not None and len(paths) > 1:
        list(pool.map(one, range(len(paths))))
    else:
        for k in range(len(paths)):
            one(k)
    return buf, starts, is_gz


def read_jsonl_batch_device(eng, paths, content_field: str, id_field=None, on_error: str = "skip", pre=None):
    """read_jsonl_shard for SEVERAL uncompressed shards at once with the field extraction ON THE DEVICE
    (Engine.jsonl_extract, csrc/mdd_jsonl.hip): the shards' bytes go files -> one pinned host buffer -> HBM, the GPU finds
    the lines, validates every line it vouches for and decodes the content field into packed utf-8 that never leaves HBM;
    lines it hands back (status fallback) run through `_py_line` (json.loads) here, exactly as in the host reader, and are
    merged in line order.  One call per batch instead of one per shard: the per-call latencies (a handful of host
    synchronisations, a scan kernel whose run time is the longest line's) are paid once.  Returns a dict
      file_of   int64 per ROW: index into `paths`;   line_no  uint64 per row: physical line number inside its file
      ids       list per row (None without id_field);  cand  int64: row index of every document that will be hashed
      text / offsets / text_bytes   DEVICE tensors: packed utf-8 (padded) and int64 offsets of those documents, in row order
      n_bad     int64 per file: lines that are not JSON objects
      line_starts  per file: int64 byte offset of every line start + the file size (None when the batch holds a "\\r")
      host_files  indices of files the device did not take (gzip): read them with read_jsonl_shard
    or None when nothing can be done on the device for this batch (field names that need JSON escaping, a lone "\\r"
    somewhere in the batch: text-mode line semantics) -- then every file goes to read_jsonl_shard.
    `pre` = a read_files_pinned result (prefetched on threads).  No EN normalisation: Engine.signatures_en does that."""
    import json

    import numpy as np
    import torch

    from .minhash import pack_texts, padded_size
    host, starts, is_gz = pre if pre is not None else read_files_pinned(paths)
    n = int(starts[-1])
    raw = torch.empty(padded_size(n), dtype=torch.uint8, device=eng.device)
    raw[:n].copy_(host[:n], non_blocking=True)
    raw[n:].zero_()
    x = eng.jsonl_extract(raw, n, content_field, id_field)
    if x is None or x["lone_cr"]:
        return None
    status, lb = x["status"], x["line_begin"].astype(np.int64)
    first_line = np.searchsorted(lb[:-1], starts[:-1], side="left")      # global index of every file's line 0
    rows = np.flatnonzero(status != 3)                                    # physical lines that are rows (blank lines are none)
    fb = np.flatnonzero(status == 2)
    hb = host.numpy()
    ids_of = {}
    if id_field:
        idk, idb, ide = x["id_kind"], x["id_beg"], x["id_end"]
        for i in np.flatnonzero(idk == 1).tolist():
            ids_of[i] = hb[int(idb[i]):int(ide[i])].tobytes().decode("utf8")
        for i in np.flatnonzero(idk == 2).tolist():
            ids_of[i] = json.loads(hb[int(idb[i]):int(ide[i])].tobytes())
    fb_docs, bad, blank = {}, [], []
    for i in fb.tolist():                                                 # lines handed back: the reference's own steps
        try:
            line = hb[int(lb[i]):int(lb[i + 1])].tobytes().decode("utf8")
            if not line.strip():                                          # blank to Python (U+0085, U+00A0 ...: str.strip knows more
                blank.append(i)                                           # whitespace than the device): no row, and no bad line
                continue
            ok, rid, data = _py_line(line, content_field, id_field, False, on_error)
        except UnicodeDecodeError:
            if on_error == "raise":
                raise
            ok, rid, data = False, None, None
        if not ok:
            bad.append(i)
            continue
        ids_of[i] = rid
        if data is not None:
            fb_docs[i] = data
    bad = np.asarray(bad, dtype=np.int64)
    if len(bad) or blank:
        rows = rows[~np.isin(rows, np.concatenate([bad, np.asarray(blank, dtype=np.int64)]))]
    file_of_line = lambda lines: np.searchsorted(first_line, lines, side="right") - 1      # noqa: E731
    n_bad = np.bincount(file_of_line(bad), minlength=len(paths)) if len(bad) else np.zeros(len(paths), dtype=np.int64)
    ids = [ids_of.get(i) for i in rows.tolist()] if id_field else [None] * len(rows)
    row_of_line = np.full(x["n_lines"] + 1, -1, dtype=np.int64)
    row_of_line[rows] = np.arange(len(rows))
    ok_lines = x["ok_lines"]
    text, offsets, total = x["text"], x["offsets"], x["text_bytes"]
    if fb_docs:
        # merge the handed-back documents in line order: [device docs | fallback docs] gathered by line
        fl = np.fromiter(sorted(fb_docs), dtype=np.int64)
        buf, fo = pack_texts([fb_docs[i] for i in fl.tolist()])
        ft, fo = eng.to_device(buf), eng.to_device(fo)
        cat_text = torch.cat([text[:total], ft])
        cat_offs = torch.cat([offsets, fo[1:] + total])
        all_lines = np.concatenate([ok_lines, fl])
        order = np.argsort(all_lines, kind="stable")
        text, offsets = eng.gather_docs(cat_text, cat_offs, eng.to_device(order.astype(np.int64)))
        total = int(offsets[-1].item())
        cand_lines = all_lines[order]
    else:
        cand_lines = ok_lines
    fo_rows = file_of_line(rows)
    # byte offset of every line start inside its file (+ the file size as the last entry), for byte-range copies of the
    # surviving lines (ZH-5); only when the batch holds no "\r" at all (else ZH-5's text-mode rewrite changes bytes)
    line_starts = None
    if x["n_cr"] == 0:
        line_starts = []
        ends = np.append(first_line[1:], x["n_lines"])
        for k in range(len(paths)):
            size_k = int(starts[k + 1] - starts[k] - 1)
            ls = lb[first_line[k]:ends[k]] - starts[k]
            line_starts.append(np.append(ls[ls < size_k], size_k))
    return {"file_of": fo_rows, "line_starts": line_starts, "line_no": (rows - first_line[fo_rows]).astype(np.uint64), "ids": ids,
            "cand": row_of_line[cand_lines], "text": text, "offsets": offsets, "text_bytes": total, "n_bad": n_bad,
            "host_files": np.flatnonzero(is_gz).tolist(), "device_lines": int(len(ok_lines)), "fallback_lines": int(len(fb))}


def read_jsonl_shard_device(eng, path: str, content_field: str, id_field=None, on_error: str = "skip"):
    """read_jsonl_shard with the field extraction on the device (a batch of one: read_jsonl_batch_device).  Same result
    keys as read_jsonl_shard, except that `text` (uint8, padded) and `offsets` (int64) are DEVICE tensors and `text_bytes`
    is added.  gzip shards, shards holding a lone "\\r" (text-mode line semantics) and field names that need JSON escaping
    are read by read_jsonl_shard and uploaded."""
    import numpy as np

    from .minhash import padded_size
    r = read_jsonl_batch_device(eng, [path], content_field, id_field, on_error)
    if r is None or r["host_files"]:
        h = read_jsonl_shard(path, content_field, id_field, False, on_error)
        t = np.zeros(padded_size(h["text"].size), dtype=np.uint8)
        t[: h["text"].size] = h["text"]
        return {"ids": h["ids"], "line_no": h["line_no"], "cand": h["cand"], "text": eng.to_device(t),
                "offsets": eng.to_device(np.ascontiguousarray(h["offsets"])), "text_bytes": int(h["text"].size), "n_bad": h["n_bad"],
                "device_lines": 0, "fallback_lines": 0}
    return {"ids": r["ids"], "line_no": r["line_no"], "cand": r["cand"], "text": r["text"], "offsets": r["offsets"],
            "text_bytes": r["text_bytes"], "n_bad": int(r["n_bad"][0]), "device_lines": r["device_lines"],
            "fallback_lines": r["fallback_lines"]}


# ------------------------------------------------------------------------------------------
# Parquet ingest without per-row Python objects: Arrow string buffers are already the packed
# utf-8 + offsets layout mdd_signatures consumes
# ------------------------------------------------------------------------------------------
def read_parquet_shard(path: str, content_field: str, id_field=None, normalize_en: bool = False):
    """One Parquet shard -> dict with
      ids        list, one entry per row (None when the id column is absent / null)
      cand       int64 [n_docs], row index of every document that will be hashed (content is a
                 non-null, valid utf-8 string; EN:124, ZH-1:52-54)
      text, offsets        packed utf-8 of the documents that are READY to hash
      ready      int64, position in `cand` of each of those documents
      raw_text, raw_offsets, raw   documents still to be normalised (EN:31-40) -- on the device with
                 Engine.signatures_en -- and their positions in `cand`
    Without normalize_en every document is ready.  With it everything is left to the device, unless
    ftfy is installed (it rewrites ASCII text too): then all documents are normalised here by
    `normalize_text`."""
    import numpy as np
    import pyarrow as pa
    import pyarrow.parquet as pq
    pf = pq.ParquetFile(path)
    names = pf.schema_arrow.names
    tbl = pf.read(columns=[c for c in dict.fromkeys((id_field, content_field)) if c and c in names])
    n = tbl.num_rows
    ids = tbl.column(id_field).to_pylist() if id_field and id_field in names else [None] * n
    e8, e64 = np.zeros(0, np.uint8), np.zeros(1, np.uint64)
    out = {"ids": ids, "cand": np.zeros(0, np.int64), "text": e8, "offsets": e64, "ready": np.zeros(0, np.int64),
           "raw_text": e8, "raw_offsets": e64, "raw": np.zeros(0, np.int64)}
    if content_field not in names:
        return out
    col = tbl.column(content_field)
    if not (pa.types.is_string(col.type) or pa.types.is_large_string(col.type)):
        return out
    arr = col.cast(pa.large_string()).combine_chunks()
    if isinstance(arr, pa.ChunkedArray):
        arr = arr.chunk(0) if arr.num_chunks else pa.array([], pa.large_string())
    valid = np.asarray(arr.is_valid())
    arr = arr.drop_null()
    try:
        arr.validate(full=True)                      # utf-8 validity of every value
        bad = None
    except pa.ArrowInvalid:
        bad = []
    bufs = arr.buffers()
    offs = np.frombuffer(bufs[1], dtype=np.int64)[arr.offset: arr.offset + len(arr) + 1].astype(np.uint64)
    data = np.frombuffer(bufs[2], dtype=np.uint8) if bufs[2] is not None else e8
    cand = np.flatnonzero(valid).astype(np.int64)
    if bad is not None:                              # rare: find the rows that are not utf-8 and leave them unhashed
        for k in range(len(arr)):
            try:
                data[int(offs[k]):int(offs[k + 1])].tobytes().decode("utf8")
            except UnicodeDecodeError:
                bad.append(k)
        keep = np.ones(len(arr), dtype=bool)
        keep[bad] = False
        cand = cand[keep]
        pieces = [data[int(offs[k]):int(offs[k + 1])].tobytes() for k in np.flatnonzero(keep).tolist()]
        data = np.frombuffer(b"".join(pieces), dtype=np.uint8)
        offs = np.zeros(len(pieces) + 1, dtype=np.uint64)
        if pieces:
            offs[1:] = np.cumsum([len(p) for p in pieces], dtype=np.uint64)
    base = int(offs[0]) if len(offs) else 0
    data = data[base:int(offs[-1])] if len(offs) else e8
    offs = offs - np.uint64(base)
    out["cand"] = cand
    nd = len(cand)
    if not normalize_en:
        out.update(text=data, offsets=offs, ready=np.arange(nd, dtype=np.int64))
        return out
    if _ftfy is not None:
        host = np.arange(nd, dtype=np.int64)
    else:
        # the device normalises non-ASCII documents too (generated Unicode table) and hands back the few it cannot
        # vouch for (Engine.signatures_en), so nothing is normalised here
        host = np.zeros(0, dtype=np.int64)
    pieces = [normalize_text(data[int(offs[k]):int(offs[k + 1])].tobytes().decode("utf8")).encode("utf8") for k in host.tolist()]
    t_off = np.zeros(len(pieces) + 1, dtype=np.uint64)
    if pieces:
        t_off[1:] = np.cumsum([len(p) for p in pieces], dtype=np.uint64)
    out.update(text=np.frombuffer(b"".join(pieces), dtype=np.uint8), offsets=t_off, ready=host)
    is_raw = np.ones(nd, dtype=bool)
    is_raw[host] = False
    raw = np.flatnonzero(is_raw).astype(np.int64)
    if len(host) == 0:
        out.update(raw_text=data, raw_offsets=offs, raw=raw)
    elif len(raw):
        lens = (offs[1:] - offs[:-1])[raw].astype(np.int64)
        r_off = np.zeros(len(raw) + 1, dtype=np.uint64)
        r_off[1:] = np.cumsum(lens).astype(np.uint64)
        # gather the byte ranges of the raw documents
        src = np.repeat(offs[:-1][raw].astype(np.int64) - r_off[:-1].astype(np.int64), lens) + np.arange(int(r_off[-1]), dtype=np.int64)
        out.update(raw_text=data[src], raw_offsets=r_off, raw=raw)
    return out
