"""Exact-match dedup stages of the EN pipeline on top of the GPU exact-grouping primitive
(`mdd_exact_group`, include/mdd.h).  Paths relative to
/root/reference/Matrix/english_data_clean_pipeline/dedup:
  exact_document_dedup  <- Exact_Document_Deduplication/exact_document_deduplication.py:16-18
  paragraph_dedup       <- Paragraph_Deduplication/paragraph.py:15-30
  exact_substr_dedup    <- Exact_Substring_Deduplication/exact_substr_dedup.py:34-47,55-95
The GPU does the grouping (hash every unit, bucket, verify the bytes); cutting the text into
units and re-assembling the survivors is vectorised numpy on the host.  The reference groups
on SHA-256 digests; equal digest <=> equal bytes, so the groups are the same."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .minhash import Engine, as_u32, pack_texts


def _group(eng: Engine, buf: np.ndarray, ubeg: np.ndarray, uend: np.ndarray, key_bits: int = 64) -> np.ndarray:
    first = eng.exact_group(eng.to_device(buf), eng.to_device(ubeg.astype(np.uint64)), eng.to_device(uend.astype(np.uint64)),
                            key_bits, text_host=buf)
    return as_u32(first)


def exact_document_dedup(eng: Engine, contents: Sequence[Optional[str]], key_bits: int = 64) -> np.ndarray:
    """Keep mask: the first row of every group of identical contents survives (nulls are one group)."""
    idx = [i for i, c in enumerate(contents) if c is not None]
    keep = np.zeros(len(contents), dtype=bool)
    nulls = [i for i, c in enumerate(contents) if c is None]
    if nulls:
        keep[nulls[0]] = True
    if idx:
        buf, offs = pack_texts([contents[i] for i in idx])
        first = _group(eng, buf, offs[:-1], offs[1:], key_bits)
        keep[np.asarray(idx)[first == np.arange(len(idx), dtype=np.uint32)]] = True
    return keep


def _count_cp(buf: np.ndarray, beg: np.ndarray, end: np.ndarray) -> np.ndarray:
    """code points in each byte range (bytes that are not utf-8 continuation bytes)"""
    start = ((buf & 0xC0) != 0x80).astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(start)])
    return cs[end] - cs[beg]


def paragraph_dedup(eng: Engine, rows: Sequence[Tuple[object, Optional[str]]], min_len: int = 20,
                    key_bits: int = 64) -> List[Tuple[object, str]]:
    """(id, text) per surviving document, sorted by id.  Rows must carry unique, mutually comparable ids."""
    rows = sorted(((rid, c) for rid, c in rows if c is not None), key=lambda r: r[0])
    if not rows:
        return []
    buf, offs = pack_texts([c for _, c in rows])
    total = int(offs[-1])
    nl = np.flatnonzero(buf[:total] == 10).astype(np.uint64)                # 0x0A never occurs inside a multi-byte code point
    doc_of_nl = np.searchsorted(offs, nl, side="right") - 1
    # unit boundaries: every document start and every byte after a newline
    ubeg = np.concatenate([offs[:-1], nl + 1])
    uend = np.concatenate([offs[1:], nl])                                    # placeholder; fixed below after sorting
    order = np.argsort(ubeg, kind="stable")
    ubeg = ubeg[order]
    # a unit ends at the next newline or at its document's end, whichever comes first
    doc_of_unit = np.concatenate([np.arange(len(rows)), doc_of_nl])[order]
    nxt = np.searchsorted(nl, ubeg, side="left")
    next_nl = np.where(nxt < len(nl), nl[np.minimum(nxt, max(len(nl) - 1, 0))] if len(nl) else 0, np.uint64(total))
    uend = np.minimum(next_nl if len(nl) else np.full(len(ubeg), total, np.uint64), offs[doc_of_unit + 1])
    # units are in (document order = id order, pos) order, so "smallest unit index" == PAR's row_number() == 1
    first = _group(eng, buf, ubeg, uend, key_bits)
    keep = first == np.arange(len(ubeg), dtype=np.uint32)
    out = []
    kb, ke, kd = ubeg[keep], uend[keep], doc_of_unit[keep]
    starts = np.searchsorted(kd, np.arange(len(rows)), side="left")
    ends = np.searchsorted(kd, np.arange(len(rows)), side="right")
    raw = buf.tobytes()
    for d, (rid, _) in enumerate(rows):
        if starts[d] == ends[d]:
            continue                                                         # no surviving paragraph: the document disappears
        text = b"\n".join(raw[int(kb[k]):int(ke[k])] for k in range(starts[d], ends[d])).decode("utf8")
        if min_len <= 0 or len(text) >= min_len:
            out.append((rid, text))
    return out


def exact_substr_dedup(eng: Engine, rows: Sequence[Tuple[Optional[str], object]], duplicate_len: int = 50,
                       min_length: int = 20, key_bits: int = 64) -> List[Tuple[object, str]]:
    """(ori_doc_id, cleaned text) for documents that keep at least min_length characters."""
    texts = [t for t, _ in rows]
    buf, offs = pack_texts(texts)
    total = int(offs[-1])
    is_start = (buf[:total] & 0xC0) != 0x80
    cp_pos = np.flatnonzero(is_start).astype(np.uint64)                       # byte offset of every code point
    cp_pos_ext = np.concatenate([cp_pos, [np.uint64(total)]])
    doc_cp_beg = np.searchsorted(cp_pos, offs[:-1], side="left")             # first code point index of each document
    doc_cp_end = np.searchsorted(cp_pos, offs[1:], side="left")
    n_cp = len(cp_pos)
    doc_of_cp = np.repeat(np.arange(len(texts)), doc_cp_end - doc_cp_beg)
    g = np.arange(n_cp)
    # window at code point g is hashed iff g - doc_beg < len - duplicate_len  (SUB:37 range(len(text) - duplicate_len))
    valid = (g + duplicate_len) < doc_cp_end[doc_of_cp] if n_cp else np.zeros(0, bool)
    wg = g[valid]
    removed_cp = np.zeros(n_cp + 1, dtype=np.int64)
    if len(wg):
        ubeg = cp_pos[wg]
        uend = cp_pos_ext[wg + duplicate_len]
        first = _group(eng, buf, ubeg, uend, key_bits)
        dup = first != np.arange(len(wg), dtype=np.uint32)                   # every occurrence but the first (lowest (doc, offset))
        np.add.at(removed_cp, wg[dup], 1)
        np.add.at(removed_cp, wg[dup] + duplicate_len, -1)
    gone = np.cumsum(removed_cp[:-1]) > 0                                    # code points covered by a removed window
    out = []
    raw = buf
    for d, (_t, ori) in enumerate(rows):
        b, e = int(doc_cp_beg[d]), int(doc_cp_end[d])
        if gone[b:e].any():
            keep_cp = ~gone[b:e]
            byte_keep = np.repeat(keep_cp, np.diff(cp_pos_ext[b:e + 1]).astype(np.int64))
            text = raw[int(offs[d]):int(offs[d + 1])][byte_keep].tobytes().decode("utf8")
            n_chars = int(keep_cp.sum())
        else:
            text, n_chars = texts[d], e - b
        if n_chars >= min_length:
            out.append((ori, text))
    return out
