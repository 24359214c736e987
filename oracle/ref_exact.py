"""CPU ORACLE (test infrastructure, NOT product code) for the exact-match stages that bracket
MinHash in the reference's EN pipeline.  Paths relative to
/root/reference/Matrix/english_data_clean_pipeline/dedup:
  EXD = Exact_Document_Deduplication/exact_document_deduplication.py
  PAR = Paragraph_Deduplication/paragraph.py
  SUB = Exact_Substring_Deduplication/exact_substr_dedup.py
The reference runs them on Spark (absent here); its grouping keys are SHA-256 digests, restated
literally with hashlib.  Where Spark leaves the survivor of a group unspecified (EXD
dropDuplicates, SUB groupByKey order) the canonical rule "first in input order" is used -- one
legal outcome of the reference.  PARITY PIN STATUS: the reference ships no tests or fixtures
for these stages either => parity unpinned, restatement-derived."""
from __future__ import annotations

import hashlib
from typing import List, Optional, Sequence, Tuple


def exact_document_dedup(contents: Sequence[Optional[str]]) -> List[bool]:
    """EXD:16-18 df.dropDuplicates([raw_content]): one row per distinct content (nulls form one
    group).  Returns the keep mask with the first row of every group kept."""
    seen, keep = set(), []
    for c in contents:
        k = None if c is None else hashlib.sha256(c.encode("utf8")).digest()
        keep.append(k not in seen)
        seen.add(k)
    return keep


def paragraph_dedup(rows: Sequence[Tuple[object, Optional[str]]], min_len: int = 20) -> List[Tuple[object, str]]:
    """PAR:15-30: split on "\\n", sha2(paragraph, 256), keep per hash the row with the smallest
    (id, pos), re-join each document's surviving paragraphs in pos order with "\\n", drop
    documents shorter than min_len characters.  Documents with null content or without a
    surviving paragraph disappear.  Returns (id, text) sorted by id."""
    best = {}
    for rid, content in rows:
        if content is None:
            continue
        for pos, par in enumerate(content.split("\n")):          # F.split keeps empty strings
            h = hashlib.sha256(par.encode("utf8")).hexdigest()    # F.sha2(paragraph, 256)
            if h not in best or (rid, pos) < best[h][:2]:
                best[h] = (rid, pos, par)
    docs = {}
    for rid, pos, par in best.values():
        docs.setdefault(rid, []).append((pos, par))
    out = []
    for rid in sorted(docs):
        text = "\n".join(p for _, p in sorted(docs[rid]))
        if min_len <= 0 or len(text) >= min_len:
            out.append((rid, text))
    return out


def exact_substr_dedup(rows: Sequence[Tuple[Optional[str], object]], duplicate_len: int = 50,
                       min_length: int = 20) -> List[Tuple[object, str]]:
    """SUB:34-47,55-70,73-95: every window text[offset:offset+duplicate_len] for offset in
    range(len(text) - duplicate_len) (sic: the last window is not hashed), SHA-256 as a big
    integer, groupByKey, all occurrences but the first are removed character by character;
    documents shorter than min_length afterwards are dropped.  rows = (text, ori_doc_id) in
    input order (zipWithIndex, SUB:77); "first" = smallest (new_doc_id, offset)."""
    groups = {}
    for new_id, (text, _ori) in enumerate(rows):
        for offset in range(len(text) - duplicate_len):
            h = int(hashlib.sha256(text[offset:offset + duplicate_len].encode("utf-8")).hexdigest(), 16)
            groups.setdefault(h, []).append((new_id, offset))
    remove = {}
    for occ in groups.values():
        for new_id, offset in sorted(occ)[1:]:
            remove.setdefault(new_id, []).append(offset)
    out = []
    for new_id, (text, ori) in enumerate(rows):
        offs = remove.get(new_id)
        if offs is not None:
            idx = set()
            for o in offs:
                idx.update(range(o, o + duplicate_len))
            text = "".join(ch for i, ch in enumerate(text) if i not in idx)
        if len(text) >= min_length:
            out.append((ori, text))
    return out
