"""Shared logic of the five ZH drop-in commands (Matrix/chinese_data_clean_pipeline/
deduplication/minhash_lsh/*.py).  Artifact formats are the reference's:
  <M>/<band>/<file>.gz   plain pickle (despite the suffix, ZH-1:58-61) of
                         [{'doc_id': "file@idx", 'hash': bytes(8*rows, big-endian)}, ...]
  <P>/dup_paris_<band>.txt   lines "a :: b"                       (ZH-2:35,46; sic "paris")
  <C>                    pickle (components: list[list[int]], n_components, {node: doc_id})  (ZH-3:55-59)
  <D>/<file>.pkl         pickle set[int] of line indices to drop  (ZH-4:28-31)
"""
from __future__ import annotations

import json
import mmap
import os
import pickle
import time
from typing import Dict, List, Tuple

import numpy as np

from .. import hostio

BANDS, ROWS, WIDTH, NUM_PERM = 9, 13, 13, 128      # ZH-1:14-17


def natural_key(doc_id: str) -> Tuple[str, int]:
    name, _, idx = doc_id.rpartition("@")
    return (name, int(idx))


def read_jsonl_texts(path: str, field: str):
    """(line_idx, text bytes) for every line that parses, has a string in `field` and can be utf-8
    encoded; other lines are skipped like ZH-1:43-54 (skipped lines are never removed).  Returns
    (line numbers uint64 [n], packed text uint8, offsets uint64 [n+1])."""
    r = hostio.read_jsonl_shard(path, field, None, normalize_en=False, on_error="skip")
    if r["n_bad"]:
        print(f"procces file {os.path.basename(path)}: {r['n_bad']} lines skipped (not JSON objects)")
    return r["line_no"][r["cand"]], r["text"], r["offsets"]


def iter_shards(files, field: str, workers: int):
    """read_jsonl_texts for every file, in file order, parsed on `workers` threads (the native reader holds no GIL) with
    at most 2 x workers parsed shards waiting for the consumer: parsing runs ahead of the device, host memory stays bounded."""
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, int(workers or 1))
    ahead = 2 * workers
    with ThreadPoolExecutor(workers) as pool:
        pending = [pool.submit(read_jsonl_texts, p, field) for p in files[:ahead]]
        for fi in range(len(files)):
            res = pending[fi].result()
            pending[fi] = None
            if fi + ahead < len(files):
                pending.append(pool.submit(read_jsonl_texts, files[fi + ahead], field))
            yield (fi,) + tuple(res)


def iter_shards_device(eng, files, field: str, workers: int, batch_bytes: int = 256 << 20, line_index=None):
    """iter_shards with the field extraction on the device (hostio.read_jsonl_batch_device): the files are taken in
    batches of about `batch_bytes`; threads only read a batch's files into ONE pinned host buffer, the GPU finds the
    lines, validates and decodes them; two batches are in flight on streams of their own.  Yields per batch
    (file indices int64 per document, line numbers uint64 per document, text DEVICE uint8 padded, offsets DEVICE int64
    [n+1], text_bytes).  gzip shards, batches with a lone "\\r" and unsuitable field names go through the host reader.
    line_index (a dict, optional) receives {file index: byte offsets of the file's line starts + its size} for the files
    whose lines can be copied as byte ranges (copy_without_lines)."""
    from concurrent.futures import ThreadPoolExecutor

    from ..minhash import to_padded_device
    workers = max(1, int(workers or 1))
    batches, cur, size = [], [], 0
    for k, p in enumerate(files):
        sz = os.path.getsize(p)
        if cur and size + sz > batch_bytes:
            batches.append(cur)
            cur, size = [], 0
        cur.append(k)
        size += sz
    if cur:
        batches.append(cur)

    def host_route(k):
        line_no, text, offs = read_jsonl_texts(files[k], field)
        return (np.full(len(line_no), k, dtype=np.int64), line_no, to_padded_device(eng, text), eng.to_device(np.ascontiguousarray(offs)),
                int(text.size))

    # Two producer threads, each with a stream and a context of its own, take the batches in turn: read (threads of
    # rpool) -> H2D -> line index / scan / decode -> the host's share of the bookkeeping.  So batch i+1 crosses PCIe while
    # batch i is being scanned, and the numpy work on one batch runs while the GPU has the other (ctypes calls, torch's
    # synchronisations and numpy release the interpreter lock).  This thread takes the results in batch order, orders its
    # own stream behind the producer's and hashes.
    import threading

    import torch

    from ..minhash import Engine
    tls = threading.local()
    timing = os.environ.get("MDD_TIMING")

    def produce(bi):
        b = batches[bi]
        if not hasattr(tls, "eng"):
            tls.stream = torch.cuda.Stream(device=eng.device)
            with torch.cuda.stream(tls.stream):
                tls.eng = Engine(eng.device_index)
        t0 = time.time()
        with torch.cuda.stream(tls.stream):
            pre = hostio.read_files_pinned([files[k] for k in b], rpool)
            t1 = time.time()
            r = hostio.read_jsonl_batch_device(tls.eng, [files[k] for k in b], field, None, on_error="skip", pre=pre)
            del pre
            ev = torch.cuda.Event()
            ev.record(tls.stream)
        if timing:
            print("[timing] batch %d (%d files): read %.1f ms, H2D + extract + bookkeeping %.1f ms, done at %.3f" %
                  (bi, len(b), (t1 - t0) * 1e3, (time.time() - t1) * 1e3, time.time()), flush=True)
        return r, ev

    ahead = 2
    with ThreadPoolExecutor(workers) as rpool, ThreadPoolExecutor(ahead) as bpool:
        pending = [bpool.submit(produce, bi) for bi in range(min(ahead, len(batches)))]
        for bi, b in enumerate(batches):
            r, ev = pending[bi].result()
            pending[bi] = None
            if bi + ahead < len(batches):
                pending.append(bpool.submit(produce, bi + ahead))
            if r is None:
                for k in b:
                    yield host_route(k)
                continue
            here = torch.cuda.current_stream(eng.device)
            here.wait_event(ev)
            for t in (r["text"], r["offsets"]):          # allocated on the producer's stream, used on this one from here on
                t.record_stream(here)
            for j in np.flatnonzero(r["n_bad"]).tolist():
                print(f"procces file {os.path.basename(files[b[j]])}: {int(r['n_bad'][j])} lines skipped (not JSON objects)")
            if line_index is not None and r["line_starts"] is not None:
                for j, k in enumerate(b):
                    if j not in r["host_files"]:
                        line_index[k] = r["line_starts"][j]
            cand = r["cand"]
            yield np.asarray(b, dtype=np.int64)[r["file_of"][cand]], r["line_no"][cand], r["text"], r["offsets"], r["text_bytes"]
            for j in r["host_files"]:
                yield host_route(b[j])


def band_key_bytes(sig_u32: np.ndarray, band: int, rows: int) -> List[bytes]:
    """_H (ZH-1:20-21): 8-byte big-endian serialisation of hashvalues[band*r:(band+1)*r]."""
    blk = np.ascontiguousarray(sig_u32[:, band * rows:(band + 1) * rows]).astype(">u8")
    raw = blk.tobytes()
    w = 8 * rows
    return [raw[i * w:(i + 1) * w] for i in range(sig_u32.shape[0])]


def key_bytes_to_tuples(keys: List[bytes], rows: int) -> np.ndarray:
    arr = np.frombuffer(b"".join(keys), dtype=">u8").reshape(len(keys), rows)
    if (arr >> np.uint64(32)).any():
        raise ValueError("band key holds a value >= 2^32: not a MinHash band artifact")
    return arr.astype(np.uint32)


def copy_without_lines(src: str, dst: str, drop, line_starts=None) -> None:
    """ZH-5:7-13: copy `src` to `dst` byte for byte without the lines whose index is in `drop`.
    Line i = data[start[i]:start[i+1]] (a last line without "\n" counts, as in `for line in f`); the kept text is
    written as the runs between dropped lines: O(dropped lines) Python steps instead of one per line, so threads
    copying different files do not queue on the interpreter lock.
    line_starts (optional; from the device-side line index, only given for plain files without any "\\r"): byte offset of
    every line start + the file size -- the source is then mapped and the kept runs leave in ONE gathered write per 1024
    runs (os.writev straight from the mapping: page cache to page cache, no Python-level copy, no call per run;
    tools/zh5_write_probe.py: twice the rate of a copy_file_range per run, whose unaligned ranges go through a pipe)."""
    if line_starts is not None and hasattr(os, "writev"):
        try:
            n_lines = len(line_starts) - 1
            ls = np.asarray(line_starts, dtype=np.int64)
            cuts = np.asarray(sorted(x for x in drop if 0 <= x < n_lines), dtype=np.int64)
            lo = np.concatenate(([0], ls[cuts + 1]))              # kept runs [lo, hi): between dropped lines
            hi = np.concatenate((ls[cuts], [ls[-1]]))
            keep = hi > lo
            fd = os.open(dst, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            try:
                if int(ls[-1]) > 0 and keep.any():
                    with open(src, "rb") as fi:
                        m = mmap.mmap(fi.fileno(), 0, prot=mmap.PROT_READ)
                    mv = memoryview(m)
                    try:
                        iov = [mv[a:b] for a, b in zip(lo[keep].tolist(), hi[keep].tolist())]
                        for i in range(0, len(iov), 1024):          # IOV_MAX
                            chunk = iov[i:i + 1024]
                            want, got = sum(len(x) for x in chunk), os.writev(fd, chunk)
                            if got < want:                          # short write: the rest of the chunk piece by piece
                                for x in chunk:
                                    if got >= len(x):
                                        got -= len(x)
                                        continue
                                    x = x[got:]
                                    got = 0
                                    while len(x):
                                        x = x[os.write(fd, x):]
                    finally:
                        iov = chunk = x = None
                        mv.release()
                        m.close()
            finally:
                os.close(fd)
            return
        except (OSError, ValueError):      # a source that cannot be mapped (a pipe, a special file): the byte route below
            pass
    with open(src, "rb") as f:
        data = f.read()
    # The reference reads plain text only (ZH-1:40, ZH-5:9); the ingest of these commands also accepts gzip'd shards and
    # numbers their DECOMPRESSED lines, so a gzip'd shard is inflated here, filtered, and written back as gzip -- cutting
    # byte ranges out of the compressed stream would corrupt the file.
    opener = open
    if data[:2] == b"\x1f\x8b":
        import gzip
        data = gzip.decompress(data)
        opener = gzip.open
    # ZH-5:9-13 reads and writes in TEXT mode: "\r\n" and a lone "\r" end a line on input and every line goes out with
    # "\n" (universal newlines).  Identity on "\n"-only files; otherwise the same translation is applied here, so line
    # indices and output bytes are the reference's.
    if b"\r" in data:
        data = data.replace(b"\r\n", b"\n").replace(b"\r", b"\n")
    with opener(dst, "wb") as fw:
        if not drop:
            fw.write(data)
            return
        nl = np.flatnonzero(np.frombuffer(data, dtype=np.uint8) == 10)
        start = np.concatenate(([0], nl + 1, [len(data)])) if (len(data) and data[-1] != 10) else np.concatenate(([0], nl + 1))
        n_lines = len(start) - 1
        mv = memoryview(data)
        pos = 0
        for d in sorted(x for x in drop if 0 <= x < n_lines):
            fw.write(mv[pos:int(start[d])])
            pos = int(start[d + 1])
        fw.write(mv[pos:])


def write_pickle(path: str, obj) -> None:
    with open(path, "wb") as f:
        pickle.dump(obj, f)


class _PlainUnpickler(pickle.Unpickler):
    """The artifacts of these commands (and of the reference scripts) hold nothing but list / dict / set / tuple / str /
    bytes / int: any other global a pickle asks for -- which is how a pickle runs code on load -- is refused."""

    _ALLOWED = {(m, n) for m in ("builtins", "__builtin__") for n in ("list", "dict", "set", "frozenset", "tuple", "str", "bytes", "bytearray", "int")}

    @staticmethod
    def _latin1(text, encoding="latin1"):       # how protocols <= 2 spell a bytes object: _codecs.encode(str, "latin1")
        if encoding != "latin1" or not isinstance(text, str):
            raise pickle.UnpicklingError("unexpected _codecs.encode call in a dedup artifact")
        return text.encode("latin1")

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        if (module, name) == ("_codecs", "encode"):
            return self._latin1
        raise pickle.UnpicklingError("refusing to load %s.%s from a dedup artifact (only plain containers are expected)" % (module, name))


def read_own_pickle(path: str):
    """Artifacts written by these commands or by the reference scripts (band lists ZH-1:58-61, the component triple
    ZH-3:55-59, line-index sets ZH-4:28-31): plain containers only, loaded with an Unpickler that resolves no other
    global, so a crafted file cannot execute anything."""
    with open(path, "rb") as f:
        return _PlainUnpickler(f).load()
