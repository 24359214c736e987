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


def open_text(path: str, mode: str = "rt"):
    if path.endswith(".gz"):
        return gzip.open(path, mode, encoding="utf8", newline="\n") if "t" in mode else gzip.open(path, mode)
    return open(path, mode, encoding="utf8", newline="\n") if "t" in mode else open(path, mode)


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
