"""Exact-match dedup stages of the EN pipeline on top of the GPU exact-grouping primitive
(`mdd_exact_group`, include/mdd.h).  Paths relative to
/root/reference/Matrix/english_data_clean_pipeline/dedup:
  exact_document_dedup  <- Exact_Document_Deduplication/exact_document_deduplication.py:16-18
  paragraph_dedup       <- Paragraph_Deduplication/paragraph.py:15-30
  exact_substr_dedup    <- Exact_Substring_Deduplication/exact_substr_dedup.py:34-47,55-95
The `*_packed` functions work on a packed corpus resident in HBM (uint8 text with MDD_TEXT_PAD
spare bytes + int64 offsets[n+1], documents in the order that defines "first") and keep every
intermediate on the device: cutting the text into units, the grouping (HIP), marking and
compacting the survivors (torch index plumbing on the same stream).  The list-level wrappers
pack, call them and decode.  The reference groups on SHA-256 digests; equal digest <=> equal
bytes, so the groups are the same."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .minhash import Engine, pack_texts, padded_size


_CHUNK = 1 << 30      # torch's scan-based ops (nonzero, boolean indexing, cumsum) are kept below 2^31 elements per call


def _nonzero(mask: torch.Tensor) -> torch.Tensor:
    if mask.numel() <= _CHUNK:
        return torch.nonzero(mask).flatten()
    return torch.cat([torch.nonzero(mask[c: c + _CHUNK]).flatten() + c for c in range(0, mask.numel(), _CHUNK)])


def _select(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    if mask.numel() <= _CHUNK:
        return x[mask]
    return torch.cat([x[c: c + _CHUNK][mask[c: c + _CHUNK]] for c in range(0, mask.numel(), _CHUNK)])


def _cumsum(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """inclusive prefix sum of x into out (same length; out's dtype accumulates)"""
    carry = None
    for c in range(0, x.numel(), _CHUNK):
        o = out[c: c + _CHUNK]
        torch.cumsum(x[c: c + _CHUNK], 0, dtype=out.dtype, out=o)
        if carry is not None:
            o += carry
        carry = o[-1].clone()
    return out


def _covered(n: int, beg: torch.Tensor, end: torch.Tensor) -> torch.Tensor:
    """bool [n]: position lies in at least one of the ranges [beg[k], end[k])"""
    diff = torch.zeros(n + 1, dtype=torch.int32, device=beg.device)
    one = torch.ones(beg.numel(), dtype=torch.int32, device=beg.device)
    diff.index_add_(0, beg, one)
    diff.index_add_(0, end, -one)
    return _cumsum(diff[:-1], torch.empty(n, dtype=torch.int32, device=beg.device)) > 0


def _arange_like(t: torch.Tensor, n: int) -> torch.Tensor:
    return torch.arange(n, dtype=torch.int64, device=t.device)


def _is_first(eng: Engine, text, ubeg, uend, key_bits: int) -> torch.Tensor:
    """bool [n_units]: unit is the first (lowest index) of its group of identical units"""
    if ubeg.numel() >= 1 << 31:
        raise ValueError(f"{ubeg.numel()} units in one call: split the corpus (limit 2^31 - 1)")
    first = eng.exact_group(text, ubeg.contiguous(), uend.contiguous(), key_bits)
    return first == torch.arange(ubeg.numel(), dtype=torch.int32, device=first.device)   # u32 bits; n_units < 2^31 compares equal


def _compact(text: torch.Tensor, total: int, offs: torch.Tensor, keep_byte: torch.Tensor, keep_doc: torch.Tensor):
    """Packed corpus of the kept bytes of the kept documents -> (text', offsets', kept document indices)"""
    csum = torch.zeros(total + 1, dtype=torch.int64, device=text.device)
    _cumsum(keep_byte, csum[1:])
    doc_len = csum[offs[1:]] - csum[offs[:-1]]
    del csum
    docs = _nonzero(keep_doc)
    keep_byte = keep_byte & _covered(total, offs[:-1][docs], offs[1:][docs])
    body = _select(text[:total], keep_byte)
    out = torch.zeros(padded_size(body.numel()), dtype=torch.uint8, device=text.device)
    out[: body.numel()] = body
    new_offs = torch.zeros(docs.numel() + 1, dtype=torch.int64, device=text.device)
    torch.cumsum(doc_len[docs], 0, out=new_offs[1:])
    return out, new_offs, docs


# ------------------------------------------------------------------------------------------
# exact document dedup
# ------------------------------------------------------------------------------------------
def exact_document_dedup_packed(eng: Engine, text: torch.Tensor, offs: torch.Tensor, key_bits: int = 64) -> torch.Tensor:
    """bool [n_docs]: document is the first with its content (EXD:17 dropDuplicates)"""
    return _is_first(eng, text, offs[:-1], offs[1:], key_bits)


def exact_document_dedup(eng: Engine, contents: Sequence[Optional[str]], key_bits: int = 64) -> np.ndarray:
    """Keep mask: the first row of every group of identical contents survives (nulls are one group)."""
    idx = [i for i, c in enumerate(contents) if c is not None]
    keep = np.zeros(len(contents), dtype=bool)
    nulls = [i for i, c in enumerate(contents) if c is None]
    if nulls:
        keep[nulls[0]] = True
    if idx:
        buf, offs = pack_texts([contents[i] for i in idx])
        k = exact_document_dedup_packed(eng, eng.to_device(buf), eng.to_device(offs), key_bits)
        keep[np.asarray(idx)[k.cpu().numpy()]] = True
    return keep


# ------------------------------------------------------------------------------------------
# paragraph dedup
# ------------------------------------------------------------------------------------------
def paragraph_dedup_packed(eng: Engine, text: torch.Tensor, offs: torch.Tensor, min_len: int = 20, key_bits: int = 64,
                           fused: bool = True):
    """Documents in (id, then input) order.  Returns (text', offsets', kept document indices):
    every paragraph ("\\n"-separated, PAR:16-17) that already occurred earlier -- in an earlier
    document or earlier in the same one -- is dropped (PAR:18-23), the survivors of a document are
    re-joined with "\\n" (PAR:26-27), documents left without a paragraph or with fewer than
    min_len code points are dropped (PAR:29).
    fused (default): `mdd_para_*` kernels cut and re-join the paragraphs around `mdd_exact_group`;
    fused=False: the same steps as torch index plumbing (independent implementation for cross-checks)."""
    if fused:
        return eng.paragraph_dedup(text, offs, min_len, key_bits)
    dev = text.device
    nd = offs.numel() - 1
    total = int(offs[-1].item()) if nd > 0 else 0
    if nd <= 0:
        return torch.zeros(padded_size(0), dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int64, device=dev), \
            torch.zeros(0, dtype=torch.int64, device=dev)
    body = text[:total]
    nl = _nonzero(body == 10)                                  # 0x0A never occurs inside a multi-byte code point
    doc_of_nl = torch.searchsorted(offs, nl, right=True) - 1
    nl_before = torch.searchsorted(nl, offs[:-1])                             # newlines before each document
    n_units = nd + nl.numel()
    # unit index: first paragraph of document d = nl_before[d] + d; the paragraph after newline j = j + doc_of_nl[j] + 1
    d_idx = _arange_like(offs, nd)
    first_unit = nl_before + d_idx
    after_nl = _arange_like(offs, nl.numel()) + doc_of_nl + 1
    ubeg = torch.empty(n_units, dtype=torch.int64, device=dev)
    uend = torch.empty(n_units, dtype=torch.int64, device=dev)
    doc_of_unit = torch.empty(n_units, dtype=torch.int64, device=dev)
    ubeg[first_unit] = offs[:-1]
    ubeg[after_nl] = nl + 1
    uend[after_nl - 1] = nl
    last_unit = torch.cat([first_unit[1:], torch.tensor([n_units], dtype=torch.int64, device=dev)]) - 1
    uend[last_unit] = offs[1:]
    doc_of_unit[first_unit] = d_idx
    doc_of_unit[after_nl] = doc_of_nl
    keep_unit = _is_first(eng, text, ubeg, uend, key_bits)
    # first kept paragraph of every document: it is written without the newline in front of it
    kept_units = _nonzero(keep_unit)
    kd = doc_of_unit[kept_units]
    n_kept = torch.zeros(nd, dtype=torch.int64, device=dev).index_add_(0, kd, torch.ones_like(kd))
    is_first_kept = torch.ones_like(kept_units, dtype=torch.bool)
    if kept_units.numel() > 1:
        is_first_kept[1:] = kd[1:] != kd[:-1]
    start = ubeg[kept_units] - (~is_first_kept).to(torch.int64)               # take the "\n" that precedes the paragraph along
    keep_byte = _covered(total, start, uend[kept_units])
    keep_doc = n_kept > 0
    if min_len > 0:
        starts = torch.zeros(total + 1, dtype=torch.int64, device=dev)
        _cumsum(keep_byte & ((body & 0xC0) != 0x80), starts[1:])
        keep_doc &= (starts[offs[1:]] - starts[offs[:-1]]) >= min_len
    return _compact(text, total, offs, keep_byte, keep_doc)


def _decode(out_text, out_offs) -> List[str]:
    t, o = out_text.cpu().numpy().tobytes(), out_offs.cpu().numpy()
    return [t[int(o[k]):int(o[k + 1])].decode("utf8") for k in range(len(o) - 1)]


def paragraph_dedup(eng: Engine, rows: Sequence[Tuple[object, Optional[str]]], min_len: int = 20,
                    key_bits: int = 64, fused: bool = True) -> List[Tuple[object, str]]:
    """(id, text) per surviving document, sorted by id.  Ids must be unique and mutually comparable
    (the reference would merge the rows of a repeated id, PAR:26)."""
    from .hostio import id_order_key
    rows = sorted(((rid, c) for rid, c in rows if c is not None), key=lambda r: id_order_key(r[0]))
    if not rows:
        return []
    buf, offs = pack_texts([c for _, c in rows])
    t, o, docs = paragraph_dedup_packed(eng, eng.to_device(buf), eng.to_device(offs), min_len, key_bits, fused)
    return [(rows[d][0], s) for d, s in zip(docs.cpu().tolist(), _decode(t, o))]


# ------------------------------------------------------------------------------------------
# exact substring dedup
# ------------------------------------------------------------------------------------------
def exact_substr_dedup_packed(eng: Engine, text: torch.Tensor, offs: torch.Tensor, duplicate_len: int = 50,
                              min_length: int = 20, key_bits: int = 64, fused: bool = True):
    """Documents in input order (SUB:75 zipWithIndex).  Returns (text', offsets', kept document
    indices): every window of duplicate_len code points starting at offsets
    range(len - duplicate_len) (SUB:38) that already occurred at a smaller (document, offset) has
    its code points deleted (SUB:44-47,58-67); documents left with fewer than min_length code
    points are dropped (SUB:90).
    fused (default): the dedicated kernels (`mdd_substr_mark/count/write`: windows are never materialised,
    exact table, no torch index plumbing).  fused=False: the generic route through `mdd_exact_group` with
    explicit unit ranges (kept as an independent implementation for cross-checks; honours key_bits)."""
    if fused and key_bits == 64 and 1 <= duplicate_len <= 128:
        return eng.substr_dedup(text, offs, duplicate_len, min_length)
    dev = text.device
    nd = offs.numel() - 1
    total = int(offs[-1].item()) if nd > 0 else 0
    if nd <= 0:
        return torch.zeros(padded_size(0), dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int64, device=dev), \
            torch.zeros(0, dtype=torch.int64, device=dev)
    body = text[:total]
    is_start = (body & 0xC0) != 0x80
    cp_pos = _nonzero(is_start)                                 # byte offset of every code point
    n_cp = cp_pos.numel()
    cp_end_of_doc = torch.searchsorted(cp_pos, offs[1:])                       # code points before each document's end
    doc_of_cp = torch.searchsorted(offs, cp_pos, right=True) - 1
    g = _arange_like(offs, n_cp)
    wg = _nonzero((g + duplicate_len) < cp_end_of_doc[doc_of_cp])
    del doc_of_cp, g
    gone = torch.zeros(n_cp, dtype=torch.bool, device=dev)
    if wg.numel():
        ubeg = cp_pos[wg]
        uend = cp_pos[wg + duplicate_len]                                      # in range: wg + duplicate_len < the document's last code point + 1
        dup = ~_is_first(eng, text, ubeg, uend, key_bits)
        del ubeg, uend
        dw = _select(wg, dup)
        gone = _covered(n_cp, dw, dw + duplicate_len)                          # code points covered by a removed window
    if n_cp:
        cp_of_byte = _cumsum(is_start, torch.empty(total, dtype=torch.int64, device=dev)) - 1
        keep_byte = ~gone[cp_of_byte]
        del cp_of_byte
    else:
        keep_byte = torch.zeros(total, dtype=torch.bool, device=dev)
    kept_cp = torch.zeros(n_cp + 1, dtype=torch.int64, device=dev)
    _cumsum(~gone, kept_cp[1:])
    cp_beg_of_doc = torch.searchsorted(cp_pos, offs[:-1])
    keep_doc = (kept_cp[cp_end_of_doc] - kept_cp[cp_beg_of_doc]) >= min_length
    return _compact(text, total, offs, keep_byte, keep_doc)


def exact_substr_dedup(eng: Engine, rows: Sequence[Tuple[Optional[str], object]], duplicate_len: int = 50,
                       min_length: int = 20, key_bits: int = 64, fused: bool = True) -> List[Tuple[object, str]]:
    """(ori_doc_id, cleaned text) for documents that keep at least min_length characters."""
    if not rows:
        return []
    buf, offs = pack_texts([t for t, _ in rows])
    t, o, docs = exact_substr_dedup_packed(eng, eng.to_device(buf), eng.to_device(offs), duplicate_len, min_length, key_bits, fused)
    return [(rows[d][1], s) for d, s in zip(docs.cpu().tolist(), _decode(t, o))]
