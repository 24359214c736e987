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
import os
import pickle
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


def write_pickle(path: str, obj) -> None:
    with open(path, "wb") as f:
        pickle.dump(obj, f)


def read_own_pickle(path: str):
    """Artifacts written by these commands (or by the reference scripts run by the same user).
    Pickle executes code on load: only point this at files you produced."""
    with open(path, "rb") as f:
        return pickle.load(f)
