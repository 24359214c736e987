"""The reference's own function names, batched (SURVEY.md 8(b) "Python function level").  A maintainer who keeps
Matrix/english_data_clean_pipeline/dedup/Minhash_Lsh_Deduplication/minhashLsh.py (EN) or
Matrix/chinese_data_clean_pipeline/deduplication/minhash_lsh/generate_minhash.py (ZH-1) can replace the per-element
map(...) calls by one call per partition; every element of the result equals what the reference function returns
for that element (same types: uint64 hashvalues, 8*r-byte big-endian band keys).

    generate_minhash_value   EN:56-64   [(text, doc_id)]        -> [(uint64[num_perm], doc_id)]
    generate_bands           EN:66-71   [(hashvalues, doc_id)]  -> [(band, bytes, doc_id)] (flatMap order)
    generate_hash_values     ZH-1:23-33 [text]                  -> [[bytes] * bands]
    normalize_text           EN:31-40   [(text, doc_id)]        -> [(text, doc_id)]
    generate_edges           EN:73-80   [doc_id]                -> [(larger, smaller)] consecutive pairs
    remove_list              EN:138-145 edges                   -> doc_ids that are not the minimum of their component
The GPU does the hashing (generate_minhash_value / generate_hash_values) and the connected components
(remove_list); the byte serialisation and the tiny generate_edges are host code, as they only re-shape results."""
from __future__ import annotations

from typing import Hashable, List, Optional, Sequence, Tuple

import numpy as np

from . import hostio
from .minhash import BANDS, NUM_PERM, ROWS, WIDTH, Engine, as_u32, pack_texts

_engine: Optional[Engine] = None


def engine() -> Engine:
    """one lazily created context on the current device (the reference's functions are stateless)"""
    global _engine
    if _engine is None:
        _engine = Engine()
    return _engine


def _H(hs: np.ndarray) -> bytes:
    """EN:28-29 / ZH-1:20-21: bytes(hs.byteswap().data) of a uint64 array = 8-byte big-endian values"""
    return np.ascontiguousarray(hs, dtype=np.uint64).astype(">u8").tobytes()


def normalize_text(pairs: Sequence[Tuple[str, Hashable]]) -> List[Tuple[str, Hashable]]:
    return [(hostio.normalize_text(t), d) for t, d in pairs]


def _signatures(texts: Sequence[str], num_perm: int, width: int) -> np.ndarray:
    if not texts:
        return np.zeros((0, num_perm), dtype=np.uint64)
    eng = engine()
    buf, offs = pack_texts(texts)                     # raises UnicodeEncodeError on lone surrogates, as x.encode('utf8') does
    sig = eng.signatures(eng.to_device(buf), eng.to_device(offs), num_perm, width, "char", 1)
    return as_u32(sig).reshape(len(texts), num_perm).astype(np.uint64)     # MinHash.hashvalues is uint64


def generate_minhash_value(pairs: Sequence[Tuple[str, Hashable]], num_perm: int = NUM_PERM, width: int = WIDTH):
    hv = _signatures([t for t, _ in pairs], num_perm, width)
    return [(hv[k], d) for k, (_, d) in enumerate(pairs)]


def generate_bands(pairs, bands: int = BANDS, r: int = ROWS):
    ret = []
    for hashvalues, doc_id in pairs:
        for idx in range(bands):
            ret.append((idx, _H(hashvalues[idx * r: (idx + 1) * r]), doc_id))
    return ret


def generate_hash_values(texts: Sequence[str], bands: int = BANDS, r: int = ROWS, num_perm: int = NUM_PERM, width: int = WIDTH):
    hv = _signatures(list(texts), num_perm, width)
    return [[_H(row[idx * r: (idx + 1) * r]) for idx in range(bands)] for row in hv]


def generate_edges(doc_id_list: Sequence) -> List[tuple]:
    edges = []
    for i in range(len(doc_id_list) - 1):
        a, b = doc_id_list[i], doc_id_list[i + 1]
        edges.append((a, b) if a > b else (b, a))
    return edges


def remove_list(edges: Sequence[tuple]) -> set:
    """What the large-star / small-star loop leaves in its first column (EN:138-145): every doc_id that is not the
    smallest of its connected component.  Union-find on the GPU over dense indices of the ids."""
    ids = sorted({x for e in edges for x in e})
    if not ids:
        return set()
    index = {d: k for k, d in enumerate(ids)}         # sorted: smallest index == smallest doc_id
    eng = engine()
    e = np.asarray([(index[u], index[v]) for u, v in edges], dtype=np.uint32).reshape(-1, 2)
    labels = eng.components_init(len(ids))
    eng.components_union(labels, eng.to_device(np.ascontiguousarray(e[:, 0])), eng.to_device(np.ascontiguousarray(e[:, 1])))
    eng.components_finish(labels)
    lab = as_u32(labels)
    return {ids[k] for k in np.flatnonzero(lab != np.arange(len(ids), dtype=np.uint32)).tolist()}
